"""MerkleTree (crypto/src/merkle/mod.rs:91-458) with nodes built on the GPU.

`nodes` uses the reference's heap layout (root at 1, children of i at 2i / 2i+1, nodes[0] = default digest,
nodes[n/2..n) = parents of leaf pairs; mod.rs:344-368) so openings read it exactly as the reference does.
Openings (prove / prove_batch) are index walks over that array and stay on the host, as in the reference;
verification recomputes merges with the GPU hasher.
"""
import ctypes

import numpy as np

from .._lib import WfError, default_context, ptr


class MerkleTreeError(Exception):
    """crypto/src/errors.rs MerkleTreeError variants, by name."""


class BatchMerkleProof:
    """crypto/src/merkle/proofs.rs BatchMerkleProof { nodes: Vec<Vec<Digest>>, depth }."""

    def __init__(self, nodes, depth):
        self.nodes = nodes
        self.depth = depth

    def get_root(self, hasher, indexes, leaves, ctx=None):
        """proofs.rs:110-250: recompute the root from the opened leaves and the proof's sibling nodes.  The walk is
        level by level over the set of known nodes; each level's merges run as one GPU batch."""
        if len(indexes) == 0:
            raise MerkleTreeError("TooFewLeafIndexes")
        n = 1 << self.depth
        pos = {}
        for k, idx in enumerate(indexes):
            if idx >= n:
                raise MerkleTreeError("LeafIndexOutOfBounds(%d, %d)" % (n, idx))
            pos[idx] = k
        if len(pos) != len(indexes):
            raise MerkleTreeError("DuplicateLeafIndex")
        pairs = sorted({i - (i & 1) for i in indexes})
        if len(pairs) != len(self.nodes) or len(leaves) < len(indexes):
            raise MerkleTreeError("InvalidProof")
        used = [0] * len(pairs)                       # next unread proof node per pair (proof_pointers)

        def take(i):
            if used[i] >= len(self.nodes[i]):
                raise MerkleTreeError("InvalidProof")
            used[i] += 1
            return np.asarray(self.nodes[i][used[i] - 1])

        batch = []
        for i, p in enumerate(pairs):
            l = np.asarray(leaves[pos[p]]) if p in pos else take(i)
            r = np.asarray(leaves[pos[p + 1]]) if p + 1 in pos else take(i)
            batch.append(np.stack([l.reshape(-1).view(np.uint8), r.reshape(-1).view(np.uint8)]))
        vals = hasher.merge(np.stack(batch), ctx).reshape(len(pairs), 32)
        cur = [(p + n) >> 1 for p in pairs]
        for _ in range(1, self.depth):
            known = {node: vals[k] for k, node in enumerate(cur)}
            nxt, batch = [], []
            k = 0
            while k < len(cur):
                node, sib = cur[k], cur[k] ^ 1
                if k + 1 < len(cur) and cur[k + 1] == sib:
                    other = known[sib]
                    k += 1
                else:
                    # proof nodes are filed under the node's position in this level's list (as prove_batch wrote them)
                    other = take(k).reshape(-1).view(np.uint8)
                a, b = (known[node], other) if node & 1 == 0 else (other, known[node])
                batch.append(np.stack([a, b]))
                nxt.append(node >> 1)
                k += 1
            vals = hasher.merge(np.stack(batch), ctx).reshape(len(nxt), 32)
            cur = nxt
        if len(cur) != 1 or cur[0] != 1:
            raise MerkleTreeError("InvalidProof")
        return vals[0]


class MerkleTree:
    def __init__(self, hasher, leaves_dev, nodes_dev, ctx):
        self.hasher = hasher
        self.ctx = ctx
        self._leaves_dev = leaves_dev
        self._nodes_dev = nodes_dev
        self._nodes = None
        self._leaves = None

    # ---- construction (MerkleTree::new, mod.rs:116-135) ------------------------------------------------
    @classmethod
    def new(cls, hasher, leaves, ctx=None):
        ctx = ctx or default_context()
        if isinstance(leaves, np.ndarray):
            lv = np.ascontiguousarray(leaves).view(np.uint8).reshape(-1, 32)
            d_leaves = ctx.to_device(lv)
        else:
            d_leaves = leaves
        n = d_leaves.numel() // 32
        d_nodes = ctx.empty_u8(max(n, 1), 32)
        try:
            ctx.call("wf_merkle_build", hasher.HASH_ID, ptr(d_leaves), n, ptr(d_nodes))
        except WfError as e:
            if e.status == 3:
                raise MerkleTreeError("TooFewLeaves(2, %d)" % n) from None
            if e.status == 2:
                raise MerkleTreeError("NumberOfLeavesNotPowerOfTwo(%d)" % n) from None
            raise
        return cls(hasher, d_leaves, d_nodes, ctx)

    # ---- accessors ----------------------------------------------------------------------------------------
    @property
    def nodes(self):
        """the whole node array on the host (copied once, on first use; openings below do not need it)"""
        if self._nodes is None:
            self._nodes = self.ctx.to_host(self._nodes_dev).reshape(-1, 32)
        return self._nodes

    @property
    def leaves(self):
        if self._leaves is None:
            self._leaves = self.ctx.to_host(self._leaves_dev).reshape(-1, 32)
        return self._leaves

    @property
    def nodes_device(self):
        return self._nodes_dev

    def num_leaves(self):
        return len(self._leaves) if self._leaves_dev is None else self._leaves_dev.numel() // 32

    def _fetch(self, which, idxs):
        """digests at the given positions of the leaf ('L') or node ('N') array: from the host copy when one exists,
        otherwise one gather on the device (wf_rows_fetch, 32-byte rows) — a query touches ~depth * num_queries of the
        2n digests, so the tree itself stays in HBM."""
        host = self._leaves if which == "L" else self._nodes
        if host is not None:
            return host[np.asarray(idxs, dtype=np.int64)]
        dev = self._leaves_dev if which == "L" else self._nodes_dev
        pos = np.ascontiguousarray(idxs, dtype=np.uint64)
        out = np.empty((len(pos), 32), dtype=np.uint8)
        if len(pos):
            self.ctx.call("wf_rows_fetch", ptr(dev), 4, 4, 8, pos.ctypes.data_as(ctypes.c_void_p), len(pos),
                          out.ctypes.data_as(ctypes.c_void_p))
        return out

    def root(self):
        return self._fetch("N", [1])[0]

    def depth(self):
        return self.num_leaves().bit_length() - 1

    # ---- openings (mod.rs:193-272) ------------------------------------------------------------------------
    def prove(self, index):
        n = self.num_leaves()
        if index >= n:
            raise MerkleTreeError("LeafIndexOutOfBounds(%d, %d)" % (n, index))
        lv = self._fetch("L", [index, index ^ 1])
        path = []
        i = (index + n) >> 1
        while i > 1:
            path.append(i ^ 1)
            i >>= 1
        return lv[0], [lv[1]] + list(self._fetch("N", path))

    def prove_batch(self, indexes):
        if len(indexes) == 0:
            raise MerkleTreeError("TooFewLeafIndexes")
        n = self.num_leaves()
        index_map = {}
        for pos, idx in enumerate(indexes):
            if idx >= n:
                raise MerkleTreeError("LeafIndexOutOfBounds(%d, %d)" % (n, idx))
            index_map[idx] = pos
        if len(index_map) != len(indexes):
            raise MerkleTreeError("DuplicateLeafIndex")
        # pass 1: the index walk of mod.rs:217-272, recording WHICH digests go where; pass 2 fetches them in two gathers
        pairs = sorted({i - (i & 1) for i in indexes})
        want_l, want_n = [], []                      # positions in the leaf / node arrays
        leaf_slots = [None] * len(index_map)         # -> index into want_l
        nodes = []                                   # per pair: list of ('L' | 'N', index into want_*)
        nxt = []
        for p in pairs:
            missing = []
            for i in (p, p + 1):
                want_l.append(i)
                if i in index_map:
                    leaf_slots[index_map[i]] = len(want_l) - 1
                else:
                    missing.append(("L", len(want_l) - 1))
            nodes.append(missing)
            nxt.append((p + n) >> 1)
        for _ in range(1, self.depth()):
            cur, nxt = nxt, []
            i = 0
            while i < len(cur):
                sib = cur[i] ^ 1
                if i + 1 < len(cur) and cur[i + 1] == sib:
                    i += 1
                else:
                    want_n.append(sib)
                    nodes[i].append(("N", len(want_n) - 1))
                nxt.append(sib >> 1)
                i += 1
        got = {"L": self._fetch("L", want_l), "N": self._fetch("N", want_n)}
        leaves = [got["L"][k] for k in leaf_slots]
        return leaves, BatchMerkleProof([[got[w][k] for w, k in lst] for lst in nodes], self.depth())

    # ---- verification (mod.rs:283-307) ----------------------------------------------------------------------
    @staticmethod
    def verify_batch(hasher, root, indexes, leaves, proof, ctx=None):
        """MerkleTree::verify_batch (mod.rs:296-307): Ok(()) <=> returns None; raises InvalidProof otherwise."""
        if not np.array_equal(proof.get_root(hasher, indexes, leaves, ctx), np.asarray(root).reshape(-1).view(np.uint8)):
            raise MerkleTreeError("InvalidProof")

    @staticmethod
    def verify(hasher, root, index, leaf, proof, ctx=None):
        pair = [leaf, proof[0]] if index & 1 == 0 else [proof[0], leaf]
        v = hasher.merge(np.stack(pair), ctx)
        index = (index + (1 << len(proof))) >> 1
        for p in proof[1:]:
            pair = [v, p] if index & 1 == 0 else [p, v]
            v = hasher.merge(np.stack(pair), ctx)
            index >>= 1
        if not np.array_equal(v, np.asarray(root)):
            raise MerkleTreeError("InvalidProof")
