"""FriProver commit phase on the GPU (fri/src/prover/mod.rs:100-239).

The Fiat–Shamir channel stays on the host and is supplied by the caller (the reference's `ProverChannel` trait,
fri/src/prover/channel.rs:24-50): any object with `commit_fri_layer(root)` and `draw_fri_alpha() -> E` works.
"""
import ctypes

import numpy as np

from .._lib import WF_FIELD_F64, default_context, ptr, torch_u64
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
        coin = channel.fri_device_coin() if hasattr(channel, "fri_device_coin") else None
        if coin is not None and self.options.num_fri_layers(length) > 0:
            return self._build_layers_fused(channel, coin, ev, length, off_p)
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

    def _build_layers_fused(self, channel, coin, ev, length, off_p, defer=False):
        """the same loop AND the remainder step as one library call against a device-resident coin (wf_fri_build_layers): commit,
        reseed, draw, fold for every layer, then interpolate / reverse / hash / reseed for the remainder, are queued back to back;
        roots, alphas, the remainder and the coin come back in one read at the end"""
        ctx, D, N, f = self.ctx, self.D, self.options.folding_factor, self.options.field
        nl = self.options.num_fri_layers(length)
        log_len = length.bit_length() - 1
        ew = D * f.W * 8                                         # bytes per E element
        # every layer's shape is checked BEFORE the coin moves to the device and anything is queued: a layer of one row has no
        # Merkle tree (MerkleTree::new: TooFewLeaves, crypto/src/merkle/mod.rs:117), and finding that out mid-chain would leave
        # the device coin reseeded for the earlier layers and the channel's host coin stale
        rows_k = length
        for k in range(nl):
            rows_k //= N
            if rows_k < 2:
                raise ValueError("FRI layer %d would have %d row(s): a Merkle tree needs at least two leaves (folding factor %d, "
                                 "domain size %d)" % (k, rows_k, N, length))
        # one allocation for every layer's four arrays, one for what comes back (roots | alphas | remainder | the coin)
        sizes, rows = [], length
        for _ in range(nl):
            rows //= N
            sizes.append((rows * N * ew, rows * 32, rows * 32, rows * ew))
        pool = ctx.empty_u8(sum(sum(t) for t in sizes))
        tr, lv, nd, fo, at = [], [], [], [], 0
        for k, (a, b, c, d) in enumerate(sizes):
            r = b // 32
            tr.append(pool[at:at + a].view(torch_u64()).view(r, N * D * f.W))
            lv.append(pool[at + a:at + a + b].view(r, 32))
            nd.append(pool[at + a + b:at + a + b + c].view(r, 32))
            fo.append(pool[at + a + b + c:at + a + b + c + d].view(torch_u64()))
            at += a + b + c + d
        rem_size = rows // self.options.blowup_factor
        assert rem_size >= 1
        o_alpha, o_rem, o_coin = (nl + 1) * 32, (nl + 1) * 32 + nl * ew, (nl + 1) * 32 + nl * ew + rem_size * ew
        back = ctx.empty_u8(o_coin + 64)
        roots, alphas, rem, state = back[:o_alpha], back[o_alpha:o_rem], back[o_rem:o_coin], back[o_coin:]
        coin.move_to(state)
        arr = lambda ts: (ctypes.c_void_p * nl)(*[t.data_ptr() for t in ts])
        def queue():
            ctx.call("wf_fri_build_layers", self.hasher.HASH_ID, f.ID, D, ptr(ev), log_len, N, nl, off_p, ptr(coin.state), arr(tr), arr(lv), arr(nd),
                     arr(fo), ptr(roots), ptr(alphas), self.options.blowup_factor, ptr(rem))

        def finish():
            host = ctx.to_host(back)                             # the one wait of the commit phase
            coin.set_host_image(host[o_coin:])
            h_roots = host[:o_alpha].reshape(nl + 1, 32)
            channel.absorb_fri_layers(coin, h_roots[:nl], host[o_alpha:o_rem].view(np.uint64).reshape(nl, D * f.W), remainder_commitment=h_roots[nl])
            for k in range(nl):
                self.layers.append(FriLayer(MerkleTree(self.hasher, lv[k], nd[k], ctx), tr[k]))
            self.remainder_poly = np.array(host[o_rem:o_coin].view(np.uint64).reshape(rem_size, D * f.W), copy=True)

        try:
            queue()
            if defer:
                # prove() against a device coin (prover/prove.py): the caller keeps queueing — grinding, the query draw — on the coin
                # that now lives in `back`, and calls finish() when its own wait comes
                return finish
            finish()
        except Exception:
            # the device coin may have absorbed some of the layers: the channel's coin must not be used with a transcript that
            # no longer matches it
            if hasattr(channel, "invalidate_coin"):
                channel.invalidate_coin()
            raise

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


    # ---- query phase (fri/src/prover/mod.rs:253-290, 296-317) ------------------------------------------------------------------
    def build_proof(self, positions):
        """FriProver::build_proof: for every layer the queried rows (N evaluations each, gathered from the device-resident
        transposed layer) and the batch opening against the layer commitment; then the remainder.  Resets the prover."""
        assert self.remainder_poly is not None, "FRI layers have not been built yet"
        f, D, N = self.options.field, self.D, self.options.folding_factor
        layers = []
        if self.layers:
            positions = list(positions)
            domain_size = self.layers[0].evaluations.shape[0] * N
            for layer in self.layers:
                positions = fold_positions(positions, domain_size, N)
                _, proof = layer.commitment.open_many(positions)
                pos = np.ascontiguousarray(positions, dtype=np.uint64)
                rows = np.empty((len(pos), N * D * f.W), dtype=np.uint64)
                self.ctx.call("wf_rows_fetch", ptr(layer.evaluations), N * D, N * D, 8 * f.W, pos.ctypes.data_as(ctypes.c_void_p), len(pos),
                              rows.ctypes.data_as(ctypes.c_void_p))
                layers.append(FriProofLayer(rows, proof))
                domain_size //= N
        remainder = self.remainder_poly
        self.reset()
        return FriProof(layers, remainder, 1)


def fold_positions(positions, source_domain_size, folding_factor):
    """fri::folding::fold_positions (fri/src/folding/mod.rs:159-176): position mod the folded domain size, duplicates dropped
    (first occurrence kept)."""
    target = source_domain_size // folding_factor
    out, seen = [], set()
    for p in positions:
        q = p % target
        if q not in seen:
            seen.add(q)
            out.append(q)
    return out


class FriProofLayer:
    """fri/src/proof.rs:240-270: the queried values of one layer (one row of N evaluations per folded position, in the order
    of the positions) and the batch opening proof.  Kept as arrays; byte serialisation is out of scope."""

    def __init__(self, query_values, proof):
        assert len(query_values) > 0, "query values cannot be empty"
        self.values, self.proof = query_values, proof


class FriProof:
    """fri/src/proof.rs:27-74."""

    def __init__(self, layers, remainder, num_partitions=1):
        assert len(remainder) > 0, "number of remainder elements must be greater than zero"
        assert len(remainder) & (len(remainder) - 1) == 0, "size of the remainder must be a power of two, but was %d" % len(remainder)
        assert num_partitions > 0 and num_partitions & (num_partitions - 1) == 0, "number of partitions must be a power of two, but was %d" % num_partitions
        self.layers, self.remainder, self._log_partitions = layers, remainder, num_partitions.bit_length() - 1

    def num_layers(self):
        return len(self.layers)

    def num_remainder_elements(self):
        return len(self.remainder)

    def num_partitions(self):
        return 1 << self._log_partitions


def apply_drp(values, domain_offset, alpha, folding_factor, ext_degree=1, ctx=None, field=fields.f64):
    """fri::folding::apply_drp (fri/src/folding/mod.rs:86-118): `values` = the transposed layer (rows of N evaluations, flat),
    returns the len/N folded evaluations.  alpha: ext_degree*W words; domain_offset: python int in internal form."""
    ctx = ctx or default_context()
    host = isinstance(values, np.ndarray)
    d = ctx.to_device(values) if host else values
    ew = ext_degree * field.W
    length = d.numel() // ew
    assert length & (length - 1) == 0 and folding_factor in (2, 4, 8, 16)
    out = ctx.empty_u64((length // folding_factor) * ew)
    off = field.element_words(int(domain_offset))
    a = np.ascontiguousarray(alpha, dtype=np.uint64)
    ctx.call("wf_fri_apply_drp", field.ID, ext_degree, ptr(d), length.bit_length() - 1, folding_factor, off.ctypes.data_as(ctypes.c_void_p),
             a.ctypes.data_as(ctypes.c_void_p), ptr(out))
    return ctx.to_host(out) if host else out
