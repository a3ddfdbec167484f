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

    @classmethod
    def from_single_proofs(cls, proofs, indexes):
        """proofs.rs:38-108: aggregate single openings `(leaf, [sibling leaf, sibling nodes...])` (MerkleTree.prove's
        output) into a batch proof.  Panics (AssertionError) on no proofs / length mismatch / unequal proof lengths."""
        assert len(proofs) > 0, "at least one proof must be provided"
        assert len(proofs) == len(indexes), "number of proofs must equal number of indexes"
        depth = len(proofs[0][1])
        by_index = {}
        for idx, pr in zip(indexes, proofs):
            assert len(pr[1]) == depth, "not all proofs have the same length"
            by_index[idx] = pr
        order = sorted(by_index)
        nodes, level = [], []                      # level: [(node index at this depth, proof)] for the surviving chains
        i = 0
        while i < len(order):
            pr = by_index[order[i]]
            if i + 1 < len(order) and order[i] & 1 == 0 and order[i + 1] == order[i] + 1:
                nodes.append([])                   # the sibling is itself an opened leaf
                i += 1
            else:
                nodes.append([np.asarray(pr[1][0])])
            level.append((order[i] >> 1, pr))
            i += 1
        for d in range(1, depth):
            nxt, i = [], 0
            while i < len(level):
                idx, pr = level[i]
                if i + 1 < len(level) and idx & 1 == 0 and level[i + 1][0] == idx + 1:
                    i += 1
                else:
                    nodes[i].append(np.asarray(pr[1][d]))
                nxt.append((idx >> 1, pr))
                i += 1
            # chains that merged keep the slot of the first of the pair, exactly like the BTreeMap walk of the reference
            dedup = {}
            for idx, pr in nxt:
                dedup[idx] = pr
            level = sorted(dedup.items())
        return cls(nodes, depth)

    def _resolve(self, hasher, indexes, leaves, ctx=None):
        """Walk the proof up to the root (proofs.rs:110-236), one GPU merge batch per level.  Returns (root, known) with
        `known` = {heap index: digest} of every node the walk touched (opened leaves, proof nodes, computed parents)."""
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
        known = {}

        def take(i):
            if used[i] >= len(self.nodes[i]):
                raise MerkleTreeError("InvalidProof")
            used[i] += 1
            return np.asarray(self.nodes[i][used[i] - 1]).reshape(-1).view(np.uint8)

        batch = []
        for i, p in enumerate(pairs):
            l = np.asarray(leaves[pos[p]]).reshape(-1).view(np.uint8) if p in pos else take(i)
            r = np.asarray(leaves[pos[p + 1]]).reshape(-1).view(np.uint8) if p + 1 in pos else take(i)
            known[n + p], known[n + p + 1] = l, r
            batch.append(np.stack([l, r]))
        vals = hasher.merge(np.stack(batch), ctx).reshape(len(pairs), 32)
        cur = [(p + n) >> 1 for p in pairs]
        for k, node in enumerate(cur):
            known[node] = vals[k]
        for _ in range(1, self.depth):
            nxt, batch = [], []
            k = 0
            while k < len(cur):
                node, sib = cur[k], cur[k] ^ 1
                if k + 1 < len(cur) and cur[k + 1] == sib:
                    k += 1
                else:
                    # proof nodes are filed under the node's position in this level's list (as prove_batch wrote them)
                    known[sib] = take(k)
                a, b = (known[node], known[sib]) if node & 1 == 0 else (known[sib], known[node])
                batch.append(np.stack([a, b]))
                nxt.append(node >> 1)
                k += 1
            vals = hasher.merge(np.stack(batch), ctx).reshape(len(nxt), 32)
            cur = nxt
            for k, node in enumerate(cur):
                known[node] = vals[k]
        if len(cur) != 1 or cur[0] != 1:
            raise MerkleTreeError("InvalidProof")
        return vals[0], known

    def get_root(self, hasher, indexes, leaves, ctx=None):
        """proofs.rs:110-236: recompute the root from the opened leaves and the proof's sibling nodes."""
        return self._resolve(hasher, indexes, leaves, ctx)[0]

    def into_openings(self, hasher, leaves, indexes, ctx=None):
        """proofs.rs:244-380: the individual openings `(leaf, [sibling leaf, sibling nodes...])` this batch proof aggregates,
        in the order of `indexes`."""
        if len(indexes) == 0:
            raise MerkleTreeError("TooFewLeafIndexes")
        if len(indexes) != len(leaves):
            raise MerkleTreeError("InvalidProof")
        _, known = self._resolve(hasher, indexes, leaves, ctx)
        n = 1 << self.depth
        out = []
        for idx in indexes:
            node, path = idx + n, []
            for _ in range(self.depth):
                if (node ^ 1) not in known:
                    raise MerkleTreeError("InvalidProof")
                path.append(known[node ^ 1])
                node >>= 1
            out.append((known[idx + n], path))
        return out

    def __eq__(self, other):
        return (isinstance(other, BatchMerkleProof) and self.depth == other.depth and len(self.nodes) == len(other.nodes)
                and all(len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b)) for a, b in zip(self.nodes, other.nodes)))


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

    @classmethod
    def from_raw_parts(cls, hasher, nodes, leaves, ctx=None):
        """MerkleTree::from_raw_parts (mod.rs:148-160): adopt already computed nodes (heap layout) and leaves — what the
        Rust shim does with the arrays the library built.  Numpy arrays stay on the host, device tensors on the device."""
        nd = np.ascontiguousarray(nodes).view(np.uint8).reshape(-1, 32) if isinstance(nodes, np.ndarray) else nodes
        lv = np.ascontiguousarray(leaves).view(np.uint8).reshape(-1, 32) if isinstance(leaves, np.ndarray) else leaves
        n_nodes = len(nd) if isinstance(nd, np.ndarray) else nd.numel() // 32
        n_leaves = len(lv) if isinstance(lv, np.ndarray) else lv.numel() // 32
        if n_leaves < 2:
            raise MerkleTreeError("TooFewLeaves(2, %d)" % n_leaves)
        if n_leaves & (n_leaves - 1):
            raise MerkleTreeError("NumberOfLeavesNotPowerOfTwo(%d)" % n_leaves)
        assert n_nodes == n_leaves, "number of nodes must equal number of leaves"      # mod.rs:155
        t = cls(hasher, None if isinstance(lv, np.ndarray) else lv, None if isinstance(nd, np.ndarray) else nd, ctx)
        if isinstance(lv, np.ndarray):
            t._leaves = lv
        if isinstance(nd, np.ndarray):
            t._nodes = nd
        return t

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

    # ---- VectorCommitment for MerkleTree (mod.rs:401-458): the names the prover calls the tree by ----------------------------
    @classmethod
    def with_options(cls, hasher, items, options=None, ctx=None):
        return cls.new(hasher, items, ctx)

    def commitment(self):
        return self.root()

    def domain_len(self):
        return 1 << self.depth()

    @staticmethod
    def get_proof_domain_len(proof):
        return 1 << len(proof)

    @staticmethod
    def get_multiproof_domain_len(proof):
        return 1 << proof.depth

    def open(self, index):
        return self.prove(index)

    def open_many(self, indexes):
        return self.prove_batch(indexes)

    @staticmethod
    def verify_many(hasher, commitment, indexes, items, proof, ctx=None):
        return MerkleTree.verify_batch(hasher, commitment, indexes, items, proof, ctx)

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
