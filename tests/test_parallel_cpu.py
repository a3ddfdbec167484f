"""world_size-2 (and 4) gloo tests of the multi-GPU sharding logic on CPU: the collectives, partition boundaries and
heap-index mapping are the product code (winterfell_amd/parallel.py); the compute steps are supplied by the CPU oracle.
The sharded result must equal the single-process commitment with the same PartitionOptions."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_boundaries_match_reference(oracle):
    from winterfell_amd import parallel
    for cols, parts, rate, D in ((64, 8, 8, 1), (20, 4, 4, 1), (10, 4, 8, 1), (10, 4, 8, 2), (255, 16, 4, 3), (7, 2, 1, 1)):
        got = parallel.column_partitions(cols, parts, rate, D)
        ps = oracle.partition_size(parts, rate, D, cols)
        assert got == [(c0, min(c0 + ps, cols)) for c0 in range(0, cols, ps)]
    # heap index mapping: subtree g's root is global node G + g
    assert [parallel.global_node_index(4, g, 1) for g in range(4)] == [4, 5, 6, 7]
    assert parallel.global_node_index(4, 1, 2) == 10 and parallel.global_node_index(4, 1, 3) == 11
    assert parallel.global_node_index(2, 1, 5) == 13


class _OracleBackend:
    def __init__(self, hasher_id):
        import oracle
        self.o, self.h = oracle, hasher_id

    def lde_and_partition_digests(self, trace_shard, domain):
        import torch
        polys, lde, leaves, _ = self.o.build_trace_commitment(self.h, trace_shard, domain["blowup"], domain["offset"])
        return polys, lde, torch.from_numpy(leaves.copy())

    def merge_many_rows(self, digests):
        import torch
        d = digests.numpy()
        return torch.from_numpy(np.stack([self.o.merge_many(self.h, d[r]) for r in range(d.shape[0])]))

    def merkle_nodes(self, leaves):
        import torch
        if leaves.shape[0] == 1:
            return leaves.clone()
        return torch.from_numpy(self.o.merkle_build(self.h, leaves.numpy()))


def _worker(rank, world, port, hasher_id, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import splitmix64
    import oracle
    from winterfell_amd import parallel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    n, c, blowup = 64, 8 * world // 2 + world, 4                   # column count divisible into `world` partitions
    c = world * 3
    trace = oracle.f64_from_int(splitmix64(1234, n * c)).reshape(c, n)
    parts = parallel.column_partitions(c, world, 1, 1)
    assert len(parts) == world
    c0, c1 = parts[rank]
    res = parallel.sharded_commit(_OracleBackend(hasher_id), np.ascontiguousarray(trace[c0:c1]),
                                  dict(blowup=blowup, offset=oracle.f64_new(7)))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), nodes=res["nodes"].numpy(), top=res["top"].numpy(),
             root=res["root"].numpy(), leaves=res["leaves"].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,hasher_id", [(2, 0), (2, 1), (4, 0)])
def test_sharded_commit_equals_partitioned_single_process(oracle, tmp_path, world, hasher_id):
    import torch.multiprocessing as mp
    from conftest import splitmix64
    from winterfell_amd import parallel
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, hasher_id, str(tmp_path)), nprocs=world, join=True)
    n, blowup, c = 64, 4, world * 3
    N = n * blowup
    trace = oracle.f64_from_int(splitmix64(1234, n * c)).reshape(c, n)
    _, _, leaves, nodes = oracle.build_trace_commitment(hasher_id, trace, blowup, oracle.f64_new(7), num_partitions=world, hash_rate=1)
    per = N // world
    got = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in range(world):
        assert np.array_equal(got[r]["root"], nodes[1])                            # every rank ends with the same root
        assert np.array_equal(got[r]["leaves"], leaves[r * per:(r + 1) * per])     # its row range of the leaves
    full = parallel.assemble_nodes(world, N, [g["nodes"] for g in got], got[0]["top"])
    assert np.array_equal(full, nodes)                                             # every node, reference heap layout
    # and it differs from the unpartitioned commitment (SURVEY 8e caveat)
    _, _, _, nodes1 = oracle.build_trace_commitment(hasher_id, trace, blowup, oracle.f64_new(7))
    assert not np.array_equal(nodes1[1], nodes[1])
