"""Multi-GPU sharding of the commitment path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).

The reference's own multi-device hook is `PartitionOptions` (air/src/options.rs:391-451): with P partitions a row's
leaf is  H::merge_many([H::hash_elements(row[k*ps .. (k+1)*ps]) for k < P])  (prover/src/matrix/row_matrix.rs:204-223).
Sharding columns so that GPU g owns partition g makes the interpolation, the coset LDE and the partition digests
purely local; only digests cross xGMI:

  1. every rank: iNTT + LDE + partition digests of its own columns                          (no communication)
  2. all-to-all of partition digests: rank r receives rows [r*N/G, (r+1)*N/G) of every partition  (32 B * N * (G-1)/G per rank)
  3. every rank: leaf = merge_many(partition digests) for its row range, then its subtree of depth log2(N/G)
  4. all-gather of the G sub-roots (G * 32 B); every rank finishes the top log2(G) levels itself

The result (root, every node) is bit-identical to the single-device commitment built with the same
PartitionOptions(G, hash_rate) — not to an unpartitioned one (SURVEY 8e caveat).

The pure index logic and the collectives are separated from the compute backend so that they run under gloo on CPU
(tests/test_parallel_cpu.py drives them with the CPU oracle as the compute backend); `HipBackend` runs the same steps
on the GPU kernels.
"""
import numpy as np


def column_partitions(num_cols, num_partitions, hash_rate=1, ext_degree=1):
    """Column ranges of the reference's partitions (PartitionOptions::partition_size / num_partitions,
    air/src/options.rs:428-444).  Returns [(c0, c1), ...]; its length is the actual number of partitions."""
    if num_partitions == 1:
        return [(0, num_cols)]
    ps = max(-(-num_cols // num_partitions), hash_rate // ext_degree)
    return [(c0, min(c0 + ps, num_cols)) for c0 in range(0, num_cols, ps)]


def row_range(num_rows, world, rank):
    per = num_rows // world
    return rank * per, (rank + 1) * per


def global_node_index(world, rank, local_index):
    """Heap index, in the full tree, of node `local_index` (1-based heap index inside rank's subtree)."""
    depth = local_index.bit_length() - 1
    return ((world + rank) << depth) + (local_index - (1 << depth))


def exchange_partition_digests(local_digests, group=None):
    """local_digests: torch uint8 tensor [N, 32] = this rank's partition digest of every row.
    Returns [N/G, G, 32]: for the rows of this rank's range, the digests of all G partitions in partition order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = local_digests.shape[0]
    per = n // world
    send = local_digests.contiguous().view(world, per, 32)
    recv = torch.empty_like(send)
    try:
        dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
    except (RuntimeError, NotImplementedError):
        # backend without all-to-all: all-gather and keep our slice (more bytes, same result)
        bufs = [torch.empty_like(local_digests) for _ in range(world)]
        dist.all_gather(bufs, local_digests.contiguous(), group=group)
        rank = dist.get_rank(group)
        recv = torch.stack([b.view(world, per, 32)[rank] for b in bufs])
    return recv.permute(1, 0, 2).contiguous()      # [per, G, 32]


def gather_subroots(sub_root, group=None):
    """sub_root: torch uint8 [32] -> [G, 32] (all-gather of G * 32 bytes)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    bufs = [torch.empty_like(sub_root) for _ in range(world)]
    dist.all_gather(bufs, sub_root.contiguous(), group=group)
    return torch.stack(bufs)


class HipBackend:
    """Compute steps on this rank's GPU through the C ABI."""

    def __init__(self, hasher, ctx=None):
        from ._lib import default_context
        self.hasher, self.ctx = hasher, ctx or default_context()

    def lde_and_partition_digests(self, trace_shard, domain):
        """-> (polys ColMatrix, lde RowMatrix, digests torch uint8 [N, 32])"""
        from .prover.matrix import RowMatrix
        polys = trace_shard.interpolate_columns()
        lde = RowMatrix.evaluate_polys_over(polys, domain.blowup, domain.offset)
        return polys, lde, lde.hash_rows(self.hasher)          # unpartitioned hash of the shard = partition digest

    def merge_many_rows(self, digests):
        """digests: torch uint8 [rows, k, 32] -> [rows, 32]"""
        from ._lib import ptr
        rows, k = digests.shape[0], digests.shape[1]
        out = self.ctx.empty_u8(rows, 32)
        self.ctx.call("wf_hash_merge_many_batch", self.hasher.HASH_ID, ptr(digests.contiguous()), rows, k, ptr(out))
        return out

    def merkle_nodes(self, leaves):
        """leaves: torch uint8 [n, 32] -> nodes [n, 32] (heap order) — n may be 1 (a single leaf is its own root)."""
        from ._lib import ptr
        n = leaves.shape[0]
        if n == 1:
            return leaves.clone()
        nodes = self.ctx.empty_u8(n, 32)
        self.ctx.call("wf_merkle_build", self.hasher.HASH_ID, ptr(leaves.contiguous()), n, ptr(nodes))
        return nodes


def sharded_commit(backend, trace_shard, domain, group=None):
    """Column-sharded trace commitment.  `trace_shard` holds the columns of partition `rank`.
    Returns dict(polys, lde, leaves (this rank's row range), nodes (this rank's subtree, heap order), top (top-tree
    nodes, heap order, [G, 32]), root)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    polys, lde, digests = backend.lde_and_partition_digests(trace_shard, domain)
    if world == 1:
        gathered = digests.view(-1, 1, 32)
    else:
        gathered = exchange_partition_digests(digests, group)
    leaves = backend.merge_many_rows(gathered)
    nodes = backend.merkle_nodes(leaves)
    sub_root = nodes[1] if leaves.shape[0] > 1 else nodes[0]
    if world == 1:
        return dict(polys=polys, lde=lde, leaves=leaves, nodes=nodes, top=None, root=sub_root)
    roots = gather_subroots(sub_root, group)
    top = backend.merkle_nodes(roots)                 # G "leaves": the top log2(G) levels, computed redundantly
    return dict(polys=polys, lde=lde, leaves=leaves, nodes=nodes, top=top, root=top[1], rank=rank, world=world)


def assemble_nodes(world, num_rows, per_rank_nodes, top):
    """Test/debug helper: full heap-ordered node array from the per-rank subtrees and the top tree (numpy)."""
    full = np.zeros((num_rows, 32), dtype=np.uint8)
    full[1:world] = top[1:world]
    per = num_rows // world
    for g, nodes in enumerate(per_rank_nodes):
        if per == 1:
            continue
        for j in range(1, per):
            full[global_node_index(world, g, j)] = nodes[j]
    return full


def emulated_sharded_commit(backend, trace_shards, domain):
    """Run the G-way sharded commitment on ONE device (G logical shards, the exchange done by slicing): the same
    compute steps and index math as `sharded_commit`, used to validate the shard math where only one GPU is
    available and by the single-node emulation test."""
    import torch
    world = len(trace_shards)
    locals_ = [backend.lde_and_partition_digests(s, domain) for s in trace_shards]
    n_rows = locals_[0][2].shape[0]
    per = n_rows // world
    out = []
    for r in range(world):
        gathered = torch.stack([locals_[k][2][r * per:(r + 1) * per] for k in range(world)], dim=1).contiguous()   # [per, G, 32]
        leaves = backend.merge_many_rows(gathered)
        nodes = backend.merkle_nodes(leaves)
        out.append((leaves, nodes))
    roots = torch.stack([nd[1] if per > 1 else nd[0] for _, nd in out])
    top = backend.merkle_nodes(roots)
    return dict(shards=locals_, per_rank=out, top=top, root=top[1] if world > 1 else roots[0])
