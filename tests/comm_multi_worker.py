"""Worker of tests/test_gpu_comm_multi.py: ONE RANK of a multi-process RCCL communicator (one process per GPU, the deployment
include/winterfell_hip.h section "multi-device" describes).  Launched as

    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P tests/comm_multi_worker.py

The 128-byte id of rank 0's wf_comm_get_unique_id travels over a gloo process group (what the reference's host would do over its
own channel); every rank calls wf_comm_init_rank on ITS device, then
  1. wf_comm_all_gather / wf_comm_all_to_all on recognisable bytes,
  2. wf_comm_sharded_commit of a column-sharded trace: every rank's root must be the root of the single-device commitment under
     PartitionOptions::new(W, .) computed by the CPU oracle (air/src/options.rs:391-451, prover/src/matrix/row_matrix.rs:204-223),
     and the rank's leaves / sub-tree nodes the corresponding slices of the oracle's tree,
  3. the FRI commit phase sharded by row ranges (parallel.comm_sharded_fri_build_layers): every layer root, alpha, the remainder
     and the coin afterwards equal to the single-device FriProver's, on every rank.
Exit status 0 = every check passed on this rank."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    import oracle
    import winterfell_amd
    from winterfell_amd import crypto, fri as wfri, parallel
    from winterfell_amd._lib import ptr
    from winterfell_amd.math import fields
    ctx = winterfell_amd.default_context(local)
    lib = ctx.lib

    uid = (ctypes.c_uint8 * 128)()
    if rank == 0:
        st = lib.wf_comm_get_unique_id(uid)
        assert st == 0, "wf_comm_get_unique_id -> %d" % st
    idt = torch.tensor(list(bytes(uid)), dtype=torch.uint8)
    dist.broadcast(idt, 0)
    uid = (ctypes.c_uint8 * 128)(*idt.tolist())
    comm = ctypes.c_void_p()
    st = lib.wf_comm_init_rank(ctx.handle, uid, rank, world, ctypes.byref(comm))
    assert st == 0, "wf_comm_init_rank -> %d" % st
    assert lib.wf_comm_rank(comm) == rank and lib.wf_comm_size(comm) == world
    try:
        # ---- 1. collectives
        blk = 4096
        send = ctx.to_device(np.concatenate([np.full(blk, (16 * rank + k) & 0xff, dtype=np.uint8) for k in range(world)]))
        ag, a2a = ctx.empty_u8(world * blk), ctx.empty_u8(world * blk)
        torch.cuda.synchronize()
        assert lib.wf_comm_all_gather(comm, ptr(send), ptr(ag), blk) == 0
        assert lib.wf_comm_all_to_all(comm, ptr(send), ptr(a2a), blk) == 0
        ctx.sync()
        ag_h, a2a_h = ctx.to_host(ag).reshape(world, blk), ctx.to_host(a2a).reshape(world, blk)
        for k in range(world):
            assert (ag_h[k] == ((16 * k) & 0xff)).all(), "all_gather block %d on rank %d" % (k, rank)          # block 0 of rank k
            assert (a2a_h[k] == ((16 * k + rank) & 0xff)).all(), "all_to_all block %d on rank %d" % (k, rank)  # block `rank` of rank k

        # ---- 2. column-sharded trace commitment against the oracle's partitioned commitment
        f = fields.f64
        log_n, log_b, cps = 10, 3, 3
        n, N, c = 1 << log_n, 1 << (log_n + log_b), cps * world
        rng = np.random.default_rng(20260924)
        trace = oracle.f64_from_int(rng.integers(0, fields.M, n * c, dtype=np.uint64)).reshape(c, n)       # the same on every rank
        shard = np.ascontiguousarray(trace[rank * cps:(rank + 1) * cps])
        per = N // world
        rw = int(lib.wf_row_width(cps, 1))
        tr, lde = ctx.to_device(shard), ctx.empty_u64(N, rw)
        leaves, nodes, top, root = ctx.empty_u8(per, 32), ctx.empty_u8(per, 32), ctx.empty_u8(world, 32), np.zeros(32, dtype=np.uint8)
        off = f.element_words(f.new(7))
        torch.cuda.synchronize()
        st = lib.wf_comm_sharded_commit(comm, crypto.Blake3_256.HASH_ID, f.ID, 1, ptr(tr), cps, n, log_n, log_b,
                                        off.ctypes.data_as(ctypes.c_void_p), 0, ptr(lde), ptr(leaves), ptr(nodes), ptr(top),
                                        root.ctypes.data_as(ctypes.c_void_p))
        assert st == 0, "wf_comm_sharded_commit -> %d" % st
        ctx.sync()
        o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(0, trace, 1 << log_b, f.new(7), num_partitions=world, hash_rate=1)
        assert np.array_equal(ctx.to_host(tr), o_polys[rank * cps:(rank + 1) * cps]), "polys of rank %d" % rank
        assert np.array_equal(ctx.to_host(lde)[:, :cps], o_lde[:, rank * cps:(rank + 1) * cps]), "LDE of rank %d" % rank
        assert np.array_equal(ctx.to_host(leaves), o_leaves[rank * per:(rank + 1) * per]), "leaves of rank %d" % rank
        assert np.array_equal(root, o_nodes[1]), "root on rank %d" % rank
        nh = ctx.to_host(nodes)
        for j in range(1, per):
            depth = j.bit_length() - 1
            assert np.array_equal(nh[j], o_nodes[((world + rank) << depth) + (j - (1 << depth))]), (rank, j)

        # ---- 3. FRI commit phase sharded by row ranges, against the single-device prover's transcript
        D, log_len, Nf = 2, 14, 4
        nl = 1 << log_len
        ev = f.from_ints([int(v) % f.M for v in rng.integers(0, 1 << 63, nl * D, dtype=np.uint64)])            # the same on every rank
        fopts = wfri.FriOptions(8, Nf, 7, field=f)
        chan = wfri.DefaultProverChannel(nl, 8, crypto.Blake3_256, ext_degree=D, field=f, ctx=ctx)
        single = wfri.FriProver(fopts, crypto.Blake3_256, ext_degree=D, ctx=ctx)
        single.build_layers(chan, ev.copy())
        image = np.zeros(64, dtype=np.uint8)
        image[:32] = crypto.DefaultRandomCoin(crypto.Blake3_256, f, np.zeros(0, dtype=np.uint64), ctx).seed
        per_len = nl // world
        piece = ctx.to_device(np.ascontiguousarray(ev[rank * per_len * D:(rank + 1) * per_len * D]))
        state = ctx.to_device(image)
        torch.cuda.synchronize()
        out = parallel.comm_sharded_fri_build_layers(lib, comm, ctx, crypto.Blake3_256, fopts, piece, D, state, min_rows_per_rank=2)
        ctx.sync()
        total = fopts.num_fri_layers(nl)
        assert 0 < out["num_sharded"] <= total
        roots, alphas = ctx.to_host(out["roots"]), ctx.to_host(out["alphas"])
        for k in range(total + 1):
            assert np.array_equal(roots[k], chan.commitments[k]), "rank %d, FRI commitment %d" % (rank, k)
        for k in range(total):
            assert np.array_equal(alphas[k], chan.alphas[k]), "rank %d, alpha %d" % (rank, k)
        assert np.array_equal(ctx.to_host(out["remainder"]), single.remainder_poly), "remainder on rank %d" % rank
        assert np.array_equal(ctx.to_host(state)[:32], chan.public_coin.seed), "coin after the commit phase on rank %d" % rank
        dist.barrier()
    finally:
        lib.wf_comm_destroy(comm)
    print("rank %d of %d: ok" % (rank, world), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
