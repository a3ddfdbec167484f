"""FriProver commit phase on the GPU (fri/src/prover/mod.rs:100-239).

The Fiat–Shamir channel stays on the host and is supplied by the caller (the reference's `ProverChannel` trait,
fri/src/prover/channel.rs:24-50): any object with `commit_fri_layer(root)` and `draw_fri_alpha() -> E` works.
"""
import ctypes

import numpy as np

from .._lib import WF_FIELD_F64, default_context, ptr
from ..crypto.merkle import MerkleTree
from ..math import fft, fields


class FriOptions:
    """fri::FriOptions (fri/src/options.rs:13-93)."""

    def __init__(self, blowup_factor, folding_factor, remainder_max_degree, field=fields.f64):
        self.field = field
        assert blowup_factor & (blowup_factor - 1) == 0, "blowup factor must be a power of two"          # options.rs:33-36
        assert folding_factor in (2, 4, 8, 16), "folding factor %d is not supported" % folding_factor   # options.rs:37-44
        self.blowup_factor = blowup_factor
        self.folding_factor = folding_factor
        self.remainder_max_degree = remainder_max_degree

    def domain_offset(self):
        return self.field.new(self.field.GENERATOR)          # options.rs:52-54: B::GENERATOR

    def num_fri_layers(self, domain_size):
        """options.rs:85-93"""
        result, max_rem = 0, (self.remainder_max_degree + 1) * self.blowup_factor
        while domain_size > max_rem:
            domain_size //= self.folding_factor
            result += 1
        return result


class FriLayer:
    """fri/src/prover/mod.rs:111-115: the layer's commitment and its transposed evaluations (device resident)."""

    def __init__(self, commitment: MerkleTree, evaluations):
        self.commitment = commitment
        self.evaluations = evaluations


class FriProver:
    def __init__(self, options: FriOptions, hasher, ext_degree=1, ctx=None):
        self.options, self.hasher, self.D = options, hasher, ext_degree
        self.ctx = ctx or default_context()
        self.layers = []
        self.remainder_poly = None

    def folding_factor(self):
        return self.options.folding_factor

    def num_layers(self):
        return len(self.layers)

    def reset(self):
        self.layers, self.remainder_poly = [], None

    def build_layers(self, channel, evaluations):
        """mod.rs:179-199.  evaluations: len*D words (numpy or device tensor) over the LDE coset."""
        assert not self.layers, "a prior proof generation request has not been completed yet"
        ctx, D, N, f = self.ctx, self.D, self.options.folding_factor, self.options.field
        ev = ctx.to_device(evaluations) if isinstance(evaluations, np.ndarray) else evaluations
        ev = ev.reshape(-1)
        length = ev.numel() // (D * f.W)
        assert length & (length - 1) == 0
        off = f.element_words(int(self.options.domain_offset()))
        off_p = off.ctypes.data_as(ctypes.c_void_p)
        for _ in range(self.options.num_fri_layers(length)):
            log_len = length.bit_length() - 1
            rows = length // N
            transposed = ctx.empty_u64(rows, N * D * f.W)
            leaves = ctx.empty_u8(rows, 32)
            nodes = ctx.empty_u8(rows, 32)
            root = np.empty(32, dtype=np.uint8)
            # build_layer (mod.rs:202-222): commit ...
            ctx.call("wf_fri_layer_commit", self.hasher.HASH_ID, f.ID, D, ptr(ev), log_len, N, ptr(transposed), ptr(leaves),
                     ptr(nodes), root.ctypes.data_as(ctypes.c_void_p))
            channel.commit_fri_layer(root)
            # ... draw alpha, fold
            alpha = np.ascontiguousarray(channel.draw_fri_alpha(), dtype=np.uint64)
            assert alpha.size == D * f.W
            folded = ctx.empty_u64(rows * D * f.W)
            ctx.call("wf_fri_apply_drp", f.ID, D, ptr(transposed), log_len, N, off_p, alpha.ctypes.data_as(ctypes.c_void_p),
                     ptr(folded))
            self.layers.append(FriLayer(MerkleTree(self.hasher, leaves, nodes, ctx), transposed))
            ev, length = folded, rows
        self._set_remainder(channel, ev, length)

    def _set_remainder(self, channel, ev, length):
        """mod.rs:230-239: interpolate over the coset, keep len/blowup coefficients in reverse order, commit to them."""
        D, f = self.D, self.options.field
        if length > 1:
            coeffs = fft.interpolate_poly_with_offset(ev.clone(), None, self.options.domain_offset(), ext_degree=D, ctx=self.ctx, field=f)
            host = self.ctx.to_host(coeffs).reshape(length, D * f.W)
        else:
            host = self.ctx.to_host(ev).reshape(1, D * f.W)
        size = length // self.options.blowup_factor
        rem = np.ascontiguousarray(host[:size][::-1])
        commitment = self.hasher.hash_elements(rem.reshape(-1), self.ctx, field=f)
        channel.commit_fri_layer(commitment)
        self.remainder_poly = rem
