"""DefaultTraceLde (prover/src/trace/trace_lde/default/mod.rs:31-231) on the GPU."""
import ctypes

import numpy as np

from .._lib import WF_FIELD_F64, load_library, ptr
from ..crypto.merkle import MerkleTree
from ..math import fields
from .matrix import ColMatrix, PartitionOptions, RowMatrix


class StarkDomain:
    """The part of prover::StarkDomain the LDE needs (prover/src/domain.rs:15-76): trace length, blowup, offset."""

    def __init__(self, trace_length, blowup, offset=None, field=fields.f64):
        self.trace_length = trace_length
        self.blowup = blowup
        self.offset = field.new(field.GENERATOR) if offset is None else offset   # domain offset = B::GENERATOR

    def lde_domain_size(self):
        return self.trace_length * self.blowup

    def trace_to_lde_blowup(self):
        return self.blowup


def build_trace_commitment(hasher, trace: ColMatrix, domain: StarkDomain, partition_options=None, skip_interpolate=False, fetch_root=True):
    """build_trace_commitment (trace_lde/default/mod.rs:245-282): returns (trace_lde: RowMatrix, tree: MerkleTree,
    trace_polys: ColMatrix).  One fused library call: interpolate -> coset LDE -> row hashes -> Merkle tree.
    fetch_root=False: the call does not wait for the root (a caller that reseeds a DEVICE coin with tree.nodes_device[1])."""
    ctx = trace.ctx
    po = partition_options or PartitionOptions()
    n, b, D, f = trace.num_rows(), domain.blowup, trace.ext_degree, trace.field
    assert n == domain.trace_length
    log_n, log_b = n.bit_length() - 1, b.bit_length() - 1
    polys = trace.data.clone()
    rw = load_library().wf_row_width(trace.num_cols(), D)
    N = n * b
    lde = ctx.empty_u64(N, rw * f.W)
    leaves = ctx.empty_u8(N, 32)
    nodes = ctx.empty_u8(N, 32)
    root = np.empty(32, dtype=np.uint8)
    off = f.element_words(int(domain.offset))
    ctx.call("wf_build_trace_commitment", hasher.HASH_ID, f.ID, D, ptr(polys), trace.num_cols(), trace.col_stride(), log_n, log_b,
             off.ctypes.data_as(ctypes.c_void_p), po.num_partitions, po.hash_rate or 256, int(skip_interpolate),
             ptr(lde), ptr(leaves), ptr(nodes), root.ctypes.data_as(ctypes.c_void_p) if fetch_root else None)
    trace_lde = RowMatrix(lde, rw, trace.num_base_cols(), D, ctx, f)
    tree = MerkleTree(hasher, leaves, nodes, ctx)
    assert trace_lde.num_rows() == domain.lde_domain_size()
    return trace_lde, tree, ColMatrix(polys, D, ctx, f)


class DefaultTraceLde:
    """TraceLde implementation (trait: prover/src/trace/trace_lde/mod.rs:26-76)."""

    def __init__(self, hasher, main_trace: ColMatrix, domain: StarkDomain, partition_options=None, fetch_root=True):
        self.hasher = hasher
        self.partition_options = partition_options or PartitionOptions()
        self._blowup = domain.blowup
        self.main_segment_lde, self.main_segment_oracles, self.main_segment_polys = build_trace_commitment(
            hasher, main_trace, domain, self.partition_options, fetch_root=fetch_root)
        self.aux_segment_lde = None
        self.aux_segment_oracles = None

    @classmethod
    def new(cls, hasher, main_trace, domain, partition_options=None, fetch_root=True):
        """DefaultTraceLde::new (default/mod.rs:63-86) -> (trace_lde, trace_polys)."""
        t = cls(hasher, main_trace, domain, partition_options, fetch_root=fetch_root)
        return t, t.main_segment_polys

    def get_main_trace_commitment(self):
        return self.main_segment_oracles.root()

    def set_aux_trace(self, aux_trace: ColMatrix, domain: StarkDomain):
        """default/mod.rs:140-166 — panics (AssertionError) if already set or if the row counts differ."""
        assert self.aux_segment_lde is None, "the auxiliary trace has already been added"
        lde, tree, polys = build_trace_commitment(self.hasher, aux_trace, domain, self.partition_options)
        assert lde.num_rows() == self.main_segment_lde.num_rows(), \
            "the number of rows in the auxiliary segment must be the same as in the main segment"
        self.aux_segment_lde, self.aux_segment_oracles = lde, tree
        return polys, tree.root()

    def read_main_trace_frame_into(self, lde_step):
        """default/mod.rs:169-180: rows lde_step and (lde_step + blowup) % N."""
        nxt = (lde_step + self.blowup()) % self.trace_len()
        r = self.main_segment_lde.rows([lde_step, nxt])
        return r[0], r[1]

    def read_aux_trace_frame_into(self, lde_step):
        assert self.aux_segment_lde is not None, "expected aux segment to be present"
        nxt = (lde_step + self.blowup()) % self.trace_len()
        r = self.aux_segment_lde.rows([lde_step, nxt])
        return r[0], r[1]

    def query(self, positions):
        """default/mod.rs:199-215: rows at `positions` + batch opening, per segment."""
        out = [(self.main_segment_lde.rows(positions), self.main_segment_oracles.prove_batch(list(positions)))]
        if self.aux_segment_oracles is not None:
            out.append((self.aux_segment_lde.rows(positions), self.aux_segment_oracles.prove_batch(list(positions))))
        return out

    def trace_len(self):
        return self.main_segment_lde.num_rows()

    def blowup(self):
        return self._blowup
