"""TracePolyTable + DeepCompositionPoly (prover/src/trace/poly_table.rs, prover/src/composer/mod.rs) on the GPU.

Elements of the extension field E are arrays of `ext_degree * field.W` uint64 words in internal form (what the
reference's `E::elements_as_bytes` would show); the random coefficients and the point z come from the caller's channel.
"""
import ctypes

import numpy as np

from .._lib import ptr
from ..math import fft
from .matrix import ColMatrix


def _words(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def evaluate_columns_at(m: ColMatrix, points, ext_degree):
    """ColMatrix::evaluate_columns_at for each point (polynom::eval per column): points (k, ext_degree*W) ->
    (k, num_cols, ext_degree*W) words."""
    f = m.field
    pts, p_pts = _words(points)
    k = pts.size // (ext_degree * f.W)
    out = np.empty((k, m.num_cols(), ext_degree * f.W), dtype=np.uint64)
    m.ctx.call("wf_polys_evaluate_at", f.ID, m.ext_degree, ext_degree, ptr(m.data), m.num_cols(), m.col_stride(),
               m.num_rows().bit_length() - 1, p_pts, k, out.ctypes.data_as(ctypes.c_void_p))
    return out


class TracePolyTable:
    """prover/src/trace/poly_table.rs:24-100: main-segment polys (base field) + optional aux-segment polys (over E)."""

    def __init__(self, main_trace_polys: ColMatrix):
        assert main_trace_polys.ext_degree == 1
        self.main_trace_polys = main_trace_polys
        self.aux_trace_polys = None

    def add_aux_segment(self, aux_trace_polys: ColMatrix):
        assert self.main_trace_polys.num_rows() == aux_trace_polys.num_rows(), \
            "polynomials in auxiliary segment must be of the same size as in the main segment"     # poly_table.rs:43-47
        self.aux_trace_polys = aux_trace_polys

    def poly_size(self):
        return self.main_trace_polys.num_rows()

    def num_cols(self):
        return self.main_trace_polys.num_cols() + (self.aux_trace_polys.num_cols() if self.aux_trace_polys else 0)

    def mul_base(self, z, g_int, ext_degree):
        """z * E::from(g) for a base-field g given as a canonical integer: coordinate-wise."""
        f = self.main_trace_polys.field
        zi = f.unpack(np.ascontiguousarray(z, dtype=np.uint64).reshape(-1))
        return f.pack([f.new(f.as_int(v) * g_int % f.M) for v in zi])

    def get_ood_frame(self, z, ext_degree):
        """poly_table.rs:68-76: (current_row, next_row) = all columns at z and z*g, main columns first."""
        f = self.main_trace_polys.field
        n = self.poly_size()
        g = f.get_root_of_unity(n.bit_length() - 1)
        pts = np.stack([np.ascontiguousarray(z, dtype=np.uint64).reshape(-1), self.mul_base(z, g, ext_degree)])
        rows = evaluate_columns_at(self.main_trace_polys, pts, ext_degree)
        if self.aux_trace_polys is not None:
            assert self.aux_trace_polys.ext_degree == ext_degree
            rows = np.concatenate([rows, evaluate_columns_at(self.aux_trace_polys, pts, ext_degree)], axis=1)
        return rows[0], rows[1]


def composition_poly_ood_frame(poly, z, ext_degree):
    """CompositionPoly::get_ood_frame (composition_poly.rs:101-108): columns at z and z*g, g of the column length."""
    t = TracePolyTable.__new__(TracePolyTable)
    t.main_trace_polys, t.aux_trace_polys = poly.data, None
    f = poly.data.field
    n = poly.data.num_rows()
    g = f.get_root_of_unity(n.bit_length() - 1)
    pts = np.stack([np.ascontiguousarray(z, dtype=np.uint64).reshape(-1), t.mul_base(z, g, ext_degree)])
    rows = evaluate_columns_at(poly.data, pts, ext_degree)
    return rows[0], rows[1]


class DeepCompositionPoly:
    """prover/src/composer/mod.rs:24-182.  `cc_trace` / `cc_constraints`: DeepCompositionCoefficients {trace,
    constraints} as (k, ext_degree*W) word arrays."""

    def __init__(self, z, cc_trace, cc_constraints, ext_degree):
        self.z = np.ascontiguousarray(z, dtype=np.uint64).reshape(-1)
        self.cc_trace = np.ascontiguousarray(cc_trace, dtype=np.uint64)
        self.cc_constraints = np.ascontiguousarray(cc_constraints, dtype=np.uint64)
        self.ext_degree = ext_degree
        self.coefficients = None
        self.field = self.ctx = None

    def poly_size(self):
        return 0 if self.coefficients is None else self.coefficients.numel() // (self.ext_degree * self.field.W)

    def add_trace_polys(self, trace_polys: TracePolyTable, quotient_polys, ood_trace_states=None, ood_quotient_states=None):
        """composer/mod.rs:67-169.  The out-of-domain frames only feed the remainder that syn_div discards (see
        include/winterfell_hip.h, wf_deep_compose), so they are accepted for signature parity and not read."""
        assert self.coefficients is None                                        # composer/mod.rs:74
        main, aux, q = trace_polys.main_trace_polys, trace_polys.aux_trace_polys, quotient_polys.data
        f, ctx, D = main.field, main.ctx, self.ext_degree
        n = trace_polys.poly_size()
        assert q.num_rows() == n and q.ext_degree == D and (aux is None or aux.ext_degree == D)
        c_aux = aux.num_cols() if aux is not None else 0
        assert self.cc_trace.size == (main.num_cols() + c_aux) * D * f.W and self.cc_constraints.size == q.num_cols() * D * f.W
        out = ctx.empty_u64(n * D * f.W)
        _z, pz = _words(self.z)
        _t, pt = _words(self.cc_trace)
        _c, pc = _words(self.cc_constraints)
        ctx.call("wf_deep_compose", f.ID, D, ptr(main.data), main.num_cols(), main.col_stride(),
                 ptr(aux.data) if c_aux else None, c_aux, aux.col_stride() if c_aux else 0,
                 ptr(q.data), q.num_cols(), q.col_stride(), n.bit_length() - 1, pz, pt, pc, ptr(out))
        self.coefficients, self.field, self.ctx = out, f, ctx

    def degree(self):
        """polynom::degree_of: index of the highest non-zero coefficient (reduced on the device; one scalar comes back)."""
        n = self.poly_size()
        nz = (self.coefficients.reshape(n, -1) != 0).any(dim=1)
        first_from_top = int(nz.flip(0).to(nz.device, dtype=self.coefficients.dtype).argmax())
        return n - 1 - first_from_top if bool(nz[n - 1 - first_from_top]) else 0

    def evaluate(self, domain):
        """composer/mod.rs:174-181: evaluations over the LDE domain (device vector)."""
        return fft.evaluate_poly_with_offset(self.coefficients, None, domain.offset, domain.blowup, ext_degree=self.ext_degree,
                                             ctx=self.ctx, field=self.field)


# ---- the same steps with the point and the coefficients in DEVICE memory (prove() against a device coin, prover/prove.py) ------------
def ood_frame_dev(m: ColMatrix, d_z, ext_degree):
    """get_ood_frame with z where the device coin drew it: (2, num_cols, ext_degree*W) words on the device — row 0 at z, row 1 at
    z*g, g = the generator of the column length's domain (poly_table.rs:68-76, composition_poly.rs:101-108).  No wait."""
    f = m.field
    out = m.ctx.empty_u64(2, m.num_cols(), ext_degree * f.W)
    m.ctx.call("wf_polys_evaluate_at_dev", f.ID, m.ext_degree, ext_degree, ptr(m.data), m.num_cols(), m.col_stride(),
               m.num_rows().bit_length() - 1, ptr(d_z), 1, ptr(out))
    return out


def deep_compose_dev(trace_polys: TracePolyTable, quotient_polys, d_z, d_cc, ext_degree):
    """DeepCompositionPoly::add_trace_polys (composer/mod.rs:67-169) with z and the coefficients (trace columns first, then the
    composition columns: DeepCompositionCoefficients in draw order) on the device -> a DeepCompositionPoly whose coefficients are set."""
    main, q = trace_polys.main_trace_polys, quotient_polys.data
    assert trace_polys.aux_trace_polys is None
    f, ctx, D = main.field, main.ctx, ext_degree
    n = trace_polys.poly_size()
    assert q.num_rows() == n and q.ext_degree == D
    assert d_cc.numel() == (main.num_cols() + q.num_cols()) * D * f.W
    out = ctx.empty_u64(n * D * f.W)
    ctx.call("wf_deep_compose_dev", f.ID, D, ptr(main.data), main.num_cols(), main.col_stride(), None, 0, 0, ptr(q.data), q.num_cols(),
             q.col_stride(), n.bit_length() - 1, ptr(d_z), ptr(d_cc), ptr(out))
    deep = DeepCompositionPoly.__new__(DeepCompositionPoly)
    deep.z = deep.cc_trace = deep.cc_constraints = None          # filled in by the caller once it has read the transcript back
    deep.ext_degree, deep.coefficients, deep.field, deep.ctx = D, out, f, ctx
    return deep
