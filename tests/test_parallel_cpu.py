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


# ---- FRI commit phase sharded by contiguous row ranges (SURVEY 8e layout (i), BASELINE configs[4]) ---------------------
class _OracleFriBackend:
    def __init__(self, hasher_id, D):
        import oracle
        self.o, self.h, self.D = oracle, hasher_id, D

    def commit_rows(self, chunk_major, folding):
        import torch
        o, D = self.o, self.D
        buf = chunk_major.numpy()
        rows = buf.size // (folding * D)
        tr = o.transpose_slice(buf, folding, D).reshape(rows, folding * D)
        if rows >= 2:
            leaves, nodes = o.fri_layer_commit(self.h, tr.reshape(-1), folding, D)
            return torch.from_numpy(tr), torch.from_numpy(leaves), torch.from_numpy(nodes)
        leaves = np.stack([o.hash_elements(self.h, tr[0])])
        return torch.from_numpy(tr), torch.from_numpy(leaves), torch.from_numpy(leaves.copy())

    def fold_rows(self, rows_t, log_len, folding, row_start, offset_words, alpha):
        import torch
        out = self.o.apply_drp_rows(rows_t.numpy().reshape(-1), folding, 1 << log_len, row_start, int(offset_words[0]), alpha, self.D)
        return torch.from_numpy(out)

    def merkle_nodes(self, leaves):
        import torch
        if leaves.shape[0] == 1:
            return leaves.clone()
        return torch.from_numpy(self.o.merkle_build(self.h, leaves.numpy()))

    def finish_unsharded(self, options, channel, vector):
        return _oracle_fri(self.o, self.h, self.D, options, channel, vector.numpy().copy())


def _oracle_fri(o, hid, D, options, channel, ev):
    """The reference prover's layer loop (fri/src/prover/mod.rs:179-239) on the oracle; returns ([(rows, nodes)], remainder)."""
    N, off = options.folding_factor, int(options.domain_offset())
    length = ev.size // D
    layers = []
    for _ in range(options.num_fri_layers(length)):
        tr = o.transpose_slice(ev, N, D)
        leaves, nodes = o.fri_layer_commit(hid, tr, N, D)
        channel.commit_fri_layer(nodes[1])
        ev = o.apply_drp(tr, N, off, channel.draw_fri_alpha(), D)
        layers.append((tr.reshape(length // N, N * D), nodes))
        length //= N
    rem, com = o.fri_remainder(hid, ev, off, options.blowup_factor, D)
    channel.commit_fri_layer(com)
    return layers, rem.reshape(-1, D)


def _fri_case(world):
    from conftest import rand_field
    import oracle
    from winterfell_amd.fri import FriOptions
    D, log_len, blowup, N = 2, 12, 8, 4
    n = (1 << log_len) // blowup
    p = oracle.f64_from_int(rand_field(77, n * D))
    ev = oracle.evaluate_poly_with_offset(p, oracle.f64_new(7), blowup, D=D, par=True)
    return D, N, FriOptions(blowup, N, 7), ev


def _fri_worker(rank, world, port, hasher_id, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle
    from winterfell_amd import parallel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    D, N, opts, ev = _fri_case(world)
    per = ev.size // world
    piece = torch.from_numpy(ev[rank * per:(rank + 1) * per].copy())
    chan = oracle.ProverChannel(hasher_id, D)
    # world 4 runs the all-gather re-stride variant, the others the uneven all-to-all
    xchg = (lambda pc, ew, nf: parallel.fri_restride_allgather(pc, ew, world, rank, nf)) if world == 4 else None
    res = parallel.sharded_fri_build_layers(_OracleFriBackend(hasher_id, D), opts, chan, piece, D, exchange=xchg)
    np.savez(os.path.join(out_dir, "fri%d.npz" % rank), nlayers=len(res["layers"]), remainder=res["remainder"],
             commitments=np.stack(chan.commitments), ntail=len(res["tail"]),
             **{"rows%d" % k: l["rows"].numpy() for k, l in enumerate(res["layers"])},
             **{"nodes%d" % k: l["nodes"].numpy() for k, l in enumerate(res["layers"])},
             **{"top%d" % k: l["top"].numpy() for k, l in enumerate(res["layers"])},
             **{"tailnodes%d" % k: t[1] for k, t in enumerate(res["tail"])})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,hasher_id", [(2, 0), (4, 0), (2, 1)])
def test_sharded_fri_equals_single_process(oracle, tmp_path, world, hasher_id):
    """2^12 LDE domain, quadratic extension, folding 4, remainder degree 7: layers 2^12 -> 2^10 -> 2^8 -> 2^6 (remainder
    2^6 / 8 = 8 coefficients).  Every layer root, every node, every transposed row and the remainder must equal the
    single-process prover's; the first layers run sharded, the tail collapses onto every rank."""
    import torch.multiprocessing as mp
    from winterfell_amd import parallel
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_fri_worker, args=(world, port, hasher_id, str(tmp_path)), nprocs=world, join=True)
    D, N, opts, ev = _fri_case(world)
    ochan = oracle.ProverChannel(hasher_id, D)
    want_layers, want_rem = _oracle_fri(oracle, hasher_id, D, opts, ochan, ev.copy())
    got = [np.load(os.path.join(str(tmp_path), "fri%d.npz" % r)) for r in range(world)]
    nsh = int(got[0]["nlayers"])
    assert nsh >= 2 and nsh + int(got[0]["ntail"]) == len(want_layers) == 3
    for r in range(world):
        assert np.array_equal(got[r]["commitments"], np.stack(ochan.commitments))      # same transcript on every rank
        assert np.array_equal(got[r]["remainder"], want_rem)
    for k in range(nsh):
        rows, nodes = want_layers[k]
        per = rows.shape[0] // world
        for r in range(world):
            assert np.array_equal(got[r]["rows%d" % k], rows[r * per:(r + 1) * per])
        full = parallel.assemble_nodes(world, rows.shape[0], [g["nodes%d" % k] for g in got], got[0]["top%d" % k])
        assert np.array_equal(full, nodes)
    for k in range(int(got[0]["ntail"])):
        assert np.array_equal(got[0]["tailnodes%d" % k], want_layers[nsh + k][1])


def test_fri_restride_index_math():
    """chunk (j, g) = e[(j*G + g) * rc/G ...]; its owner under contiguous pieces is floor((j*G + g) / N)."""
    from winterfell_amd import parallel
    for world, N, length in ((2, 4, 64), (4, 4, 64), (8, 4, 256), (8, 2, 64), (4, 16, 256)):
        rc = length // N
        per, chunk = length // world, rc // world
        for g in range(world):
            for j in range(N):
                start = j * rc + g * chunk                       # first element of chunk (j, g): rows g*chunk.. of column j
                assert start == (j * world + g) * chunk
                assert start // per == parallel.fri_chunk_owner(j, g, world, N)
                assert (start + chunk - 1) // per == start // per  # a chunk never straddles two pieces


@pytest.mark.parametrize("world,N,length", [(2, 4, 64), (4, 4, 128), (8, 4, 256), (8, 2, 64), (8, 16, 1024), (4, 16, 256), (8, 8, 512)])
def test_fri_restride_plan_routes_every_chunk(world, N, length):
    """Simulate the uneven all-to-all from the per-rank plans: every rank must end up with exactly its chunk-major buffer
    [j][i] = e[i0 + i + j*rc], and the send / receive split sizes of every pair of ranks must agree (no deadlock)."""
    from winterfell_amd import parallel
    e = np.arange(length, dtype=np.int64)
    per, rc = length // world, length // N
    chunk = rc // world
    plans = [parallel.fri_restride_plan(world, r, N, per, chunk) for r in range(world)]
    for h in range(world):
        for g in range(world):
            assert plans[h][1][g] == plans[g][2][h]                 # what h sends to g == what g expects from h
    for g in range(world):
        recv = []
        for h in range(world):                                      # blocks arrive ordered by source rank
            piece = e[h * per:(h + 1) * per]
            for dest, start in plans[h][0]:
                if dest == g:
                    recv.append(piece[start:start + chunk])
        got = np.concatenate(recv)
        want = np.concatenate([e[j * rc + g * chunk:j * rc + (g + 1) * chunk] for j in range(N)])
        assert np.array_equal(got, want)
        assert sum(plans[g][2]) == N * chunk


# ---- row-strided sharding: bit-exact with the default (unpartitioned) commitment -------------------------------------
class _OracleStridedBackend(_OracleBackend):
    def local_commit(self, trace, sub_blowup, sub_offset_int):
        import torch
        polys, lde, leaves, _ = self.o.build_trace_commitment(self.h, trace, sub_blowup, self.o.f64_new(sub_offset_int))
        return polys, lde, torch.from_numpy(leaves.copy())


def _strided_worker(rank, world, port, hasher_id, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from conftest import splitmix64
    import oracle
    from winterfell_amd import parallel
    from winterfell_amd.math import fields
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    n, c, blowup = 64, 5, 8
    trace = oracle.f64_from_int(splitmix64(99, n * c)).reshape(c, n)
    res = parallel.strided_commit(_OracleStridedBackend(hasher_id), trace, n, blowup, 7, fields.f64)
    np.savez(os.path.join(out_dir, "strided%d.npz" % rank), nodes=res["nodes"].numpy(), top=res["top"].numpy(), root=res["root"].numpy(),
             leaves=res["leaves"].numpy(), lde=res["lde"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,hasher_id", [(2, 0), (4, 1), (8, 0)])
def test_strided_commit_equals_default_single_process(oracle, tmp_path, world, hasher_id):
    """SURVEY 8e Alternative B: rank k evaluates and hashes the LDE rows r = k (mod G) (a coset LDE with blowup b/G and
    offset s*g^k), leaves are exchanged, and the tree equals the DEFAULT num_partitions = 1 commitment node for node."""
    import torch.multiprocessing as mp
    from conftest import splitmix64
    from winterfell_amd import parallel
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_strided_worker, args=(world, port, hasher_id, str(tmp_path)), nprocs=world, join=True)
    n, c, blowup = 64, 5, 8
    N = n * blowup
    trace = oracle.f64_from_int(splitmix64(99, n * c)).reshape(c, n)
    _, lde, leaves, nodes = oracle.build_trace_commitment(hasher_id, trace, blowup, oracle.f64_new(7))
    got = [np.load(os.path.join(str(tmp_path), "strided%d.npz" % r)) for r in range(world)]
    per = N // world
    for r in range(world):
        assert np.array_equal(got[r]["root"], nodes[1])
        assert np.array_equal(got[r]["leaves"], leaves[r * per:(r + 1) * per])
        assert np.array_equal(got[r]["lde"], lde[r::world])                       # rank r holds rows r, r + G, r + 2G, ...
    full = parallel.assemble_nodes(world, N, [g["nodes"] for g in got], got[0]["top"])
    assert np.array_equal(full, nodes)


# ---- FRI commit phase in the verifier's partitioned layout (SURVEY 8e layout (ii)) --------------------------------------
def test_map_positions_to_indexes_is_the_verifiers_mapping():
    """fri/src/utils.rs:9-33: P = 1 is the identity; otherwise partition p % P, local index p // P, partition size
    (source / N) / P — a bijection of the folded domain that puts every partition's positions in one contiguous block."""
    from winterfell_amd.parallel import map_positions_to_indexes
    assert map_positions_to_indexes([5, 1, 7], 64, 4, 1) == [5, 1, 7]
    assert map_positions_to_indexes([0, 1, 2, 3, 9], 64, 4, 4) == [0, 4, 8, 12, 6]
    for src, N, P in [(64, 4, 4), (256, 2, 8), (1 << 12, 16, 2)]:
        rc = src // N
        idx = map_positions_to_indexes(list(range(rc)), src, N, P)
        assert sorted(idx) == list(range(rc))
        for k in range(P):
            assert [idx[k + P * q] for q in range(rc // P)] == list(range(k * (rc // P), (k + 1) * (rc // P)))


def _pfri_worker(rank, world, port, hasher_id, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle
    from winterfell_amd import parallel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    D, N, opts, ev = _fri_case(world)
    piece = torch.from_numpy(ev.reshape(-1, world, D)[:, rank, :].copy().reshape(-1))      # e[rank + world * m]
    chan = oracle.ProverChannel(hasher_id, D)
    res = parallel.partitioned_fri_build_layers(_OracleFriBackend(hasher_id, D), opts, chan, piece, D)
    np.savez(os.path.join(out_dir, "pfri%d.npz" % rank), nlayers=len(res["layers"]), remainder=res["remainder"],
             commitments=np.stack(chan.commitments),
             **{"rows%d" % k: l["rows"].numpy() for k, l in enumerate(res["layers"])},
             **{"leaves%d" % k: l["leaves"].numpy() for k, l in enumerate(res["layers"])},
             **{"nodes%d" % k: l["nodes"].numpy() for k, l in enumerate(res["layers"])},
             **{"top%d" % k: l["top"].numpy() for k, l in enumerate(res["layers"])})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,hasher_id", [(2, 0), (4, 1), (8, 0)])
def test_partitioned_fri_equals_verifier_layout(oracle, tmp_path, world, hasher_id):
    """Rank k folds the positions = k (mod P) with no evaluation exchange; the layer trees must be the single-process
    trees built over leaves placed by the verifier's map_positions_to_indexes, the transcript and the remainder the ones
    of that single-process run, and every queried row must sit where the verifier will look for it."""
    import torch.multiprocessing as mp
    from fri_partition_util import fold_positions, oracle_partitioned_fri
    from winterfell_amd import parallel
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_pfri_worker, args=(world, port, hasher_id, str(tmp_path)), nprocs=world, join=True)
    D, N, opts, ev = _fri_case(world)
    ochan = oracle.ProverChannel(hasher_id, D)
    want_layers, want_rem = oracle_partitioned_fri(oracle, hasher_id, D, opts, ochan, ev.copy(), world)
    got = [np.load(os.path.join(str(tmp_path), "pfri%d.npz" % r)) for r in range(world)]
    assert int(got[0]["nlayers"]) == len(want_layers) == 3
    positions, length = [5, 77, 1234, 4095, 2048, 77 + 1024], ev.size // D
    for r in range(world):
        assert np.array_equal(got[r]["commitments"], np.stack(ochan.commitments))
        assert np.array_equal(got[r]["remainder"], want_rem)
    for k, (rows, leaves, nodes) in enumerate(want_layers):
        rc = rows.shape[0]
        for r in range(world):
            assert np.array_equal(got[r]["rows%d" % k], rows[r::world])                       # global row r + P*q at local q
            assert np.array_equal(got[r]["leaves%d" % k], leaves[r * (rc // world):(r + 1) * (rc // world)])
        full = parallel.assemble_nodes(world, rc, [g["nodes%d" % k] for g in got], got[0]["top%d" % k])
        assert np.array_equal(full, nodes)
        # the verifier's look-up: folded position p -> leaf map(p), which must be the hash of transposed row p
        positions = fold_positions(positions, length, N)
        for p, i in zip(positions, parallel.map_positions_to_indexes(positions, length, N, world)):
            assert np.array_equal(leaves[i], oracle.hash_elements(hasher_id, rows[p]))
            assert np.array_equal(got[p % world]["leaves%d" % k][p // world], leaves[i])
        length = rc


def test_partitioned_fri_rejects_too_many_partitions():
    from winterfell_amd import parallel
    from winterfell_amd.fri import FriOptions
    with pytest.raises(ValueError):
        parallel._partitioned_layer_plan(FriOptions(8, 4, 7), 1 << 8, 128)      # 2^8 -> rows 64: not a multiple of 128
