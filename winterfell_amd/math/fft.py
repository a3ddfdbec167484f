"""math::fft on the GPU (math/src/fft/mod.rs).  Same function names, argument meaning and error behaviour.

Polynomials / evaluation vectors are 1-D arrays of base-field words in the reference's internal form; an
extension-field vector of n elements is n*ext_degree words (AoS).  Inputs may be numpy uint64 arrays (copied to
HBM and back, the Rust shim's behaviour) or torch int64 CUDA tensors (left in HBM).
"""
import ctypes

import numpy as np

from .. import _lib
from .._lib import WF_FIELD_F64, default_context, ptr
from . import fields


def _log2(n, what):
    if n == 0 or n & (n - 1):
        # math/src/fft/mod.rs:90,173,269,356: "... must be a power of 2"
        raise AssertionError("number of %s must be a power of 2" % what)
    return n.bit_length() - 1


def _is_np(x):
    return isinstance(x, np.ndarray)


def _offset_arg(domain_offset, field):
    if domain_offset == 0:
        raise AssertionError("domain offset cannot be zero")   # mod.rs:185,368
    words = field.element_words(int(domain_offset))
    return words, words.ctypes.data_as(ctypes.c_void_p)


def get_twiddles(domain_size, ctx=None, field=fields.f64):
    """fft::get_twiddles (mod.rs:455-468): domain_size/2 twiddles, bit-reverse permuted (device tensor)."""
    ctx = ctx or default_context()
    log_n = _log2(domain_size, "domain size")
    if log_n > field.TWO_ADICITY:
        raise AssertionError("multiplicative subgroup of size %d does not exist in the specified base field" % domain_size)
    out = ctx.empty_u64((domain_size // 2) * field.W)
    ctx.call("wf_fft_get_twiddles", field.ID, log_n, 0, ptr(out))
    return out


def get_inv_twiddles(domain_size, ctx=None, field=fields.f64):
    """fft::get_inv_twiddles (mod.rs:491-505)."""
    ctx = ctx or default_context()
    log_n = _log2(domain_size, "domain size")
    out = ctx.empty_u64((domain_size // 2) * field.W)
    ctx.call("wf_fft_get_twiddles", field.ID, log_n, 1, ptr(out))
    return out


def _inplace(name, p, ext_degree, ctx, batch=1, field=fields.f64):
    ctx = ctx or default_context()
    host = _is_np(p)
    d = ctx.to_device(p) if host else p
    n = d.numel() // (ext_degree * batch * field.W)
    log_n = _log2(n, "coefficients" if "evaluate" in name else "values")
    if log_n > field.TWO_ADICITY:
        raise AssertionError("multiplicative subgroup of size %d does not exist in the specified base field" % n)
    ctx.call(name, field.ID, ext_degree, ptr(d), log_n, batch)
    if host:
        p[...] = ctx.to_host(d).reshape(p.shape)
        return p
    return d


def evaluate_poly(p, twiddles=None, ext_degree=1, ctx=None, batch=1, field=fields.f64):
    """fft::evaluate_poly (mod.rs:85-112): in place; `twiddles` is accepted for signature parity and only its
    length is checked (the kernels use cached tables)."""
    if twiddles is not None:
        n = (p.size if _is_np(p) else p.numel()) // (ext_degree * batch * field.W)
        if len(twiddles) * 2 != n * field.W:
            raise AssertionError("invalid number of twiddles: expected %d but received %d" % (n // 2, len(twiddles) // field.W))
    return _inplace("wf_fft_evaluate_poly", p, ext_degree, ctx, batch, field)


def interpolate_poly(evaluations, inv_twiddles=None, ext_degree=1, ctx=None, batch=1, field=fields.f64):
    """fft::interpolate_poly (mod.rs:264-295): in place."""
    if inv_twiddles is not None:
        n = (evaluations.size if _is_np(evaluations) else evaluations.numel()) // (ext_degree * batch * field.W)
        if len(inv_twiddles) * 2 != n * field.W:
            raise AssertionError("invalid number of twiddles: expected %d but received %d" % (n // 2, len(inv_twiddles) // field.W))
    return _inplace("wf_fft_interpolate_poly", evaluations, ext_degree, ctx, batch, field)


def evaluate_poly_with_offset(p, twiddles, domain_offset, blowup_factor, ext_degree=1, ctx=None, field=fields.f64):
    """fft::evaluate_poly_with_offset (mod.rs:168-211): returns n*blowup_factor evaluations over the coset.
    `domain_offset` is a base-field element in internal form (e.g. fields.new(7))."""
    ctx = ctx or default_context()
    host = _is_np(p)
    d = ctx.to_device(p) if host else p
    n = d.numel() // (ext_degree * field.W)
    log_n = _log2(n, "coefficients")
    log_b = _log2(blowup_factor, "blowup factor")
    if log_n + log_b > field.TWO_ADICITY:
        raise AssertionError("multiplicative subgroup of size %d does not exist in the specified base field" % (n * blowup_factor))
    out = ctx.empty_u64(n * blowup_factor * ext_degree * field.W)
    _keep, off = _offset_arg(domain_offset, field)
    ctx.call("wf_fft_evaluate_poly_with_offset", field.ID, ext_degree, ptr(d), log_n, off, log_b, ptr(out))
    return ctx.to_host(out) if host else out


def interpolate_poly_with_offset(evaluations, inv_twiddles, domain_offset, ext_degree=1, ctx=None, field=fields.f64):
    """fft::interpolate_poly_with_offset (mod.rs:351-386): in place."""
    ctx = ctx or default_context()
    host = _is_np(evaluations)
    d = ctx.to_device(evaluations) if host else evaluations
    n = d.numel() // (ext_degree * field.W)
    log_n = _log2(n, "values")
    _keep, off = _offset_arg(domain_offset, field)
    ctx.call("wf_fft_interpolate_poly_with_offset", field.ID, ext_degree, ptr(d), log_n, off)
    if host:
        evaluations[...] = ctx.to_host(d).reshape(evaluations.shape)
        return evaluations
    return d


def serial_fft(values, twiddles, ext_degree=1, ctx=None, field=fields.f64):
    """fft::serial_fft (mod.rs:405-429): `fft_in_place` followed by `permute`, i.e. natural order in, natural order out —
    the same transform as evaluate_poly, with its own argument checks (the twiddles are mandatory here)."""
    n = (values.size if _is_np(values) else values.numel()) // (ext_degree * field.W)
    if n == 0 or n & (n - 1):
        raise AssertionError("number of values must be a power of 2, but was %d" % n)
    nt = (twiddles.size if _is_np(twiddles) else twiddles.numel()) // field.W
    if nt * 2 != n:
        raise AssertionError("invalid number of twiddles: expected %d but received %d" % (n // 2, nt))
    return _inplace("wf_fft_evaluate_poly", values, ext_degree, ctx, 1, field)


def permute_index(size, index):
    """fft::permute_index (mod.rs:570-578): bit reversal of `index` in a domain of `size` (a power of two)."""
    assert size & (size - 1) == 0 and index < size
    bits = size.bit_length() - 1
    return int(format(index, "0%db" % bits)[::-1], 2) if bits else 0


def infer_degree(evaluations, domain_offset, ext_degree=1, ctx=None, field=fields.f64):
    """fft::infer_degree (mod.rs:543-562): interpolate over the coset and return the index of the highest non-zero
    coefficient (polynom::degree_of); the reduction runs on the device."""
    ctx = ctx or default_context()
    n = (evaluations.size if _is_np(evaluations) else evaluations.numel()) // (ext_degree * field.W)
    if n == 0 or n & (n - 1):
        raise AssertionError("number of evaluations must be a power of 2")
    if n.bit_length() - 1 > field.TWO_ADICITY:
        raise AssertionError("multiplicative subgroup of size %d does not exist in the specified base field" % n)
    if domain_offset == 0:
        raise AssertionError("domain offset cannot be zero")
    d = ctx.to_device(evaluations) if _is_np(evaluations) else evaluations.clone()
    coeffs = interpolate_poly_with_offset(d, None, domain_offset, ext_degree=ext_degree, ctx=ctx, field=field)
    nz = (coeffs.reshape(n, -1) != 0).any(dim=1)
    top = int(nz.flip(0).to(dtype=coeffs.dtype).argmax())
    return n - 1 - top if bool(nz[n - 1 - top]) else 0
