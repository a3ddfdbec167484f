#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X (contract: see the task statement).

Workload (BASELINE.json configs[1]): standalone 2^24-point forward + inverse NTT over f64, data resident in HBM.
A "step" = one fft::evaluate_poly followed by one fft::interpolate_poly of a 2^24-element vector (natural order in
and out, in place) => 2 * 2^24 element-transforms per step.  value = element-transforms per second, whole job.
With N GPUs every rank transforms its own vector (independent columns shard with no collective): weak scaling.

Extra fields on the same JSON line:
  roofline      HBM roofline of the NTT kernels (algorithmic bytes 2*n*8 per transform / measured kernel time)
  cpu_baseline  the CPU oracle's restatement of the reference's `concurrent` (Rayon) algorithm, timed on this host
  extra         trace-LDE+commit ms (the second half of BASELINE's metric) at 2^20 rows x 4 cols, blowup 8
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def fail(msg, code=2):
    sys.stderr.write("bench.py: " + msg + "\n")
    sys.stderr.flush()
    sys.exit(code)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-run this script as N ranks (one process per GPU) under
    torch.distributed.run on 127.0.0.1.  Never returns."""
    backend = os.environ.get("WF_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        fail("--gpus %d asked for but only %d HIP device(s) are visible; refusing to report a %d-GPU number measured on fewer "
             "devices" % (args.gpus, have, args.gpus))
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (needed by RCCL)
    import subprocess
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="control flow only (launcher, rendezvous, barrier, max-over-ranks reduction): no GPU work, no metric")
    args = ap.parse_args()
    if args.gpus < 1:
        fail("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        fail("launched with WORLD_SIZE=%d but --gpus %d: one rank per GPU is the contract" % (world, args.gpus))
    dist = None
    # one process per GPU; WF_BENCH_BACKEND=gloo (with fewer devices than ranks) only exists to exercise the N > 1 control
    # flow on a single-GPU box — the measured configuration is always nccl (= RCCL) with one device per rank
    backend = os.environ.get("WF_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and not args.dry_run and ndev < world:
        fail("rank %d: %d HIP device(s) visible, %d ranks: every rank needs its own GPU" % (rank, ndev, world))
    device_index = local_rank % max(ndev, 1)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
        if dist.get_world_size() != args.gpus:
            fail("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    if args.dry_run:
        # the N > 1 control flow without a GPU: rendezvous, barrier on both sides of the timed region, MAX over ranks, one line
        t0 = time.perf_counter()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0 + 1e-3 * (rank + 1)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.barrier()
            assert float(t.item()) >= elapsed
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "f64 NTT elements/s", "value": None, "unit": "elements/s", "n_gpus": world, "dry_run": True,
                              "backend": backend if world > 1 else None}))
        return
    torch.cuda.set_device(device_index)
    local_rank = device_index

    import winterfell_amd
    from winterfell_amd import crypto, prover
    from winterfell_amd.math import fft, fields

    ctx = winterfell_amd.default_context(local_rank)
    n = 1 << args.log_n
    rng = np.random.default_rng(0x5EED0001 + args.log_n + rank)
    host = rng.integers(0, fields.M, n, dtype=np.uint64)     # uniform canonical Montgomery residues
    data = ctx.to_device(host)
    ref = data.clone()

    def step():
        fft.evaluate_poly(data)
        fft.interpolate_poly(data)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    experiment = os.environ.get("WF_BENCH_TIMING_EXPERIMENT") == "1"       # tools/build_variant.sh builds whose results are wrong by construction
    assert experiment or torch.equal(data, ref), "forward+inverse round trip is not the identity"

    ms_per_step = elapsed * 1e3 / args.steps
    value = 2.0 * n * args.steps * world / elapsed

    out = {
        "metric": "f64 NTT elements/s",
        "value": value,
        "unit": "elements/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        **({"INVALID": "timing experiment: results not checked"} if experiment else {}),
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "standalone 2^%d-point forward+inverse NTT over f64 (BASELINE configs[1]), in place, "
                               "natural order, one vector per GPU" % args.log_n,
                   "log_n": args.log_n, "step": "evaluate_poly + interpolate_poly", "parallelism": "dp%d" % world},
    }

    # ---- N > 1 only: column-sharded trace LDE + commit (partition digests over RCCL all-to-all, sub-roots all-gather).
    # Every rank owns 4 of the 4*N f64 columns of a 2^20-row trace; this is the one place the path has an exchange step.
    sharded = None
    if world > 1 and not args.no_extra:
        # Merkle leaves/s over all ranks (north_star: reported at 1/2/4/8 GPUs): one independent 2^23-leaf BLAKE3 tree per rank,
        # no data-path collective, barrier + max over ranks like the headline
        merkle_total = None
        try:
            lv = ctx.to_device(np.random.default_rng(3 + rank).integers(0, 256, (1 << 23, 32), dtype=np.uint8))
            crypto.MerkleTree.new(crypto.Blake3_256, lv)
            barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                crypto.MerkleTree.new(crypto.Blake3_256, lv)
            barrier()
            tt = torch.tensor([(time.perf_counter() - t1) / 5], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            merkle_total = world * (1 << 23) / float(tt.item())
            del lv
        except Exception as e:
            merkle_total = repr(e)[:200]
        try:
            from winterfell_amd import parallel
            tn, tb = 1 << 20, 8
            shard = prover.ColMatrix(ctx.to_device(np.random.default_rng(7 + rank).integers(0, fields.M, (4, tn), dtype=np.uint64)))
            dom = prover.StarkDomain(tn, tb)
            backend = parallel.HipBackend(crypto.Blake3_256, ctx)
            parallel.sharded_commit(backend, shard, dom)
            barrier()
            ts = []
            for _ in range(5):
                barrier()
                t1 = time.perf_counter()
                res = parallel.sharded_commit(backend, shard, dom)
                barrier()
                ts.append((time.perf_counter() - t1) * 1e3)
            tt = torch.tensor([float(np.median(ts))], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sharded = {"sharded_lde_commit_ms_2^20x%d_b8_blake3" % (4 * world): float(tt.item())}
        except Exception as e:  # never let the optional leg break the headline measurement
            sharded = {"sharded_lde_commit_error": repr(e)[:200]}
        if isinstance(merkle_total, float):
            sharded["merkle_blake3_leaves_per_s_2^23_all_ranks"] = merkle_total
            sharded["merkle_blake3_hbm_roofline_frac_per_gpu"] = 64.0 * merkle_total / world / (HBM_PEAK_GBS * 1e9)
        elif merkle_total is not None:
            sharded["merkle_all_ranks_error"] = merkle_total
        # row-strided sharding of ONE 2^20 x 4 trace (replicated input): bit-identical to the default single-device
        # commitment; rank k evaluates and hashes the LDE rows r = k (mod N), leaves cross xGMI (equal-size all-to-all)
        try:
            if world <= 8:
                full = prover.ColMatrix(ctx.to_device(np.random.default_rng(11).integers(0, fields.M, (4, 1 << 20), dtype=np.uint64)))
                sb = parallel.HipStridedBackend(crypto.Blake3_256, fields.f64, ctx)
                srun = lambda: parallel.strided_commit(sb, full, 1 << 20, 8, 7, fields.f64)
                srun()
                ts = []
                for _ in range(5):
                    barrier()
                    t1 = time.perf_counter()
                    srun()
                    barrier()
                    ts.append((time.perf_counter() - t1) * 1e3)
                tt = torch.tensor([float(np.median(ts))], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                sharded["strided_lde_commit_ms_2^20x4_b8_blake3"] = float(tt.item())
        except Exception as e:
            sharded["strided_lde_commit_error"] = repr(e)[:200]
        # FRI commit phase of a 2^24-point quadratic-extension LDE (configs[4]) sharded by row ranges: all-to-all re-stride
        # + sub-root all-gather per layer, tail layers collapsed onto every rank
        try:
            from winterfell_amd import fri as wfri, parallel

            class _Chan:
                def __init__(self):
                    self.k = 0

                def commit_fri_layer(self, root):
                    self.k += 1

                def draw_fri_alpha(self):
                    return np.array([fields.new(12345 + self.k), fields.new(777 + self.k)], dtype=np.uint64)

            piece = ctx.to_device(np.random.default_rng(100 + rank).integers(0, fields.M, ((1 << 24) // world) * 2, dtype=np.uint64))
            fopts = wfri.FriOptions(8, 4, 31)
            fbackend = parallel.HipFriBackend(crypto.Blake3_256, fields.f64, 2, ctx)
            # equal-size all-gather re-stride here (the uneven all-to-all variant is exercised by the gloo tests): a size
            # mismatch in an optional leg must never be able to hang the headline run
            xchg = lambda pc, ew, nf: parallel.fri_restride_allgather(pc, ew, world, rank, nf)
            run = lambda: parallel.sharded_fri_build_layers(fbackend, fopts, _Chan(), piece, 2, min_rows_per_rank=1 << 12, exchange=xchg)
            run()
            ts = []
            for _ in range(3):
                barrier()
                t1 = time.perf_counter()
                run()
                barrier()
                ts.append((time.perf_counter() - t1) * 1e3)
            tt = torch.tensor([float(np.median(ts))], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sharded["sharded_fri_build_layers_ms_2^24_quad_fold4_blake3"] = float(tt.item())
            # the same commit phase in the verifier's partitioned layout (FriProof.num_partitions = N):
            # rank k folds the positions = k (mod N); only the 32-byte sub-roots cross xGMI
            prun = lambda: parallel.partitioned_fri_build_layers(fbackend, fopts, _Chan(), piece, 2)
            prun()
            ts = []
            for _ in range(3):
                barrier()
                t1 = time.perf_counter()
                prun()
                barrier()
                ts.append((time.perf_counter() - t1) * 1e3)
            tt = torch.tensor([float(np.median(ts))], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sharded["partitioned_fri_build_layers_ms_2^24_quad_fold4_blake3"] = float(tt.item())
            del piece
        except Exception as e:
            sharded["sharded_fri_error"] = repr(e)[:200]

    if rank == 0:
        # ---- roofline: per-kernel durations from HIP events on the launch stream (wf_prof_*) ----
        ctx.prof_enable(True)
        reps = max(5, min(args.steps, 20))
        for _ in range(reps):
            fft.evaluate_poly(data)
        prof = ctx.prof_collect()
        ctx.prof_enable(False)
        kern = {k: {"launches": c, "avg_us": ms * 1e3 / c} for k, (c, ms) in prof.items()}
        total_ms = sum(ms for _, ms in prof.values())
        fwd_us = total_ms * 1e3 / reps
        alg_bytes = 2.0 * n * 8                         # read once + write once per transform
        achieved = alg_bytes / (fwd_us * 1e-6) / 1e9
        # HBM bytes per transform from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2 per the gfx950
        # correction + WRITE_SIZE, summed over the transform's launches); only valid for the profiled size
        traffic = None
        try:
            pmc_path = next(pth for pth in (os.path.join(ROOT, "profiles", r, "bench_pmc_summary.json") for r in ("r02", "r01")) if os.path.exists(pth))
            with open(pmc_path) as f:
                pm = json.load(f)["ntt_2^24_f64"]
            if args.log_n == 24:
                traffic = pm["hbm_bytes_per_transform"]
        except Exception:
            traffic = None
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_unit": "HBM bytes per transform (rocprofv3 PMC, profiles/<round>/bench_pmc_summary.json)",
            "limiter": "memory-level parallelism of the tile passes, not instruction issue (round 2: removing the twiddle arithmetic changes the "
                       "time by 3 %, a 4-pass radix-64 plan at 6 waves/SIMD still takes 67 us per pass; a pure read-modify-write of the same tiles "
                       "takes 50 us per pass = 5.4 TB/s); HBM traffic = 1.02x the data per pass, three passes; see DESIGN.md section 5",
            "kernel": "ntt_pass (x%d) + ntt_pass_last per 2^%d transform; durations summed" % (
                kern.get("ntt_pass", {}).get("launches", 0) // reps, args.log_n),
            "algorithmic_bytes_per_transform": alg_bytes, "transform_us": fwd_us, "kernels": kern,
        }

        if not args.no_extra:
            ex = {}

            def timed(fn, reps=5):
                fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(reps):
                    t1 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1) * 1e3)
                return float(np.median(ts))

            # ---- trace LDE + commit (second half of BASELINE's metric): 2^20 rows x 4 cols, blowup 8 (configs[2] shape) ----
            tn, tc, tb = 1 << 20, 4, 8
            for hname, tag in (("Blake3_256", "blake3"), ("Rp64_256", "rp64")):
                hasher = getattr(crypto, hname)
                cm = prover.ColMatrix(ctx.to_device(rng.integers(0, fields.M, (tc, tn), dtype=np.uint64)))
                dom = prover.StarkDomain(tn, tb)
                ex["lde_commit_ms_2^20x4_b8_f64_" + tag] = timed(lambda: prover.build_trace_commitment(hasher, cm, dom))
            # examples::rescue as shipped (configs[2], SURVEY D2 variant 3a): f128, 4 columns, Blake3_256
            f128 = fields.f128
            tr128 = rng.integers(0, 1 << 62, (tc, tn * 2), dtype=np.uint64)      # words < 2^62 => every u128 < p
            cm128 = prover.ColMatrix(ctx.to_device(tr128), field=f128)
            dom128 = prover.StarkDomain(tn, tb, field=f128)
            ex["lde_commit_ms_2^20x4_b8_f128_blake3"] = timed(lambda: prover.build_trace_commitment(crypto.Blake3_256, cm128, dom128), 3)
            # configs[2] end to end on the device: trace LDE + commit -> constraint evaluation (Rescue AIR) -> composition
            # polynomial + constraint commitment -> OOD frames + DEEP composition + its LDE  (random trace: timing only)
            from winterfell_amd import air as wair
            rair = wair.RescueAir(tn, [1, 2], [3, 4], tb)
            ew = 2 * f128.W
            cc = prover.ConstraintCompositionCoefficients(rng.integers(1, 1 << 62, (4, ew), dtype=np.uint64),
                                                          rng.integers(1, 1 << 62, (4, ew), dtype=np.uint64))
            zpt = rng.integers(1, 1 << 62, ew, dtype=np.uint64)
            cct, ccq = rng.integers(1, 1 << 62, (4, ew), dtype=np.uint64), rng.integers(1, 1 << 62, (3, ew), dtype=np.uint64)

            def rescue_pipeline():
                lde, polys = prover.DefaultTraceLde.new(crypto.Blake3_256, cm128, dom128)
                ev_ = prover.DefaultConstraintEvaluator(rair, cc, 2).evaluate(lde, dom128)
                com, cpoly = prover.build_constraint_commitment(crypto.Blake3_256, ev_, 3, dom128, ext_degree=2, field=f128, ctx=ctx)
                table = prover.TracePolyTable(polys)
                table.get_ood_frame(zpt, 2)
                prover.composition_poly_ood_frame(cpoly, zpt, 2)
                deep = prover.DeepCompositionPoly(zpt, cct, ccq, 2)
                deep.add_trace_polys(table, cpoly)
                return deep.evaluate(dom128)

            ex["rescue_2^20_f128_quad_b8_commit+constraints+composition+deep_ms"] = timed(rescue_pipeline, 3)
            ex["grind_blake3_factor20_ms"] = timed(lambda: crypto.grind_query_seed(crypto.Blake3_256, np.arange(32, dtype=np.uint8), 20), 3)
            # Merkle leaves/s (BLAKE3, 2^23 leaves)
            lv = ctx.to_device(rng.integers(0, 256, (1 << 23, 32), dtype=np.uint8))
            ms = timed(lambda: crypto.MerkleTree.new(crypto.Blake3_256, lv))
            ex["merkle_blake3_leaves_per_s_2^23"] = (1 << 23) / (ms * 1e-3)
            ex["merkle_blake3_hbm_roofline_frac"] = 64.0 * (1 << 23) / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9)   # 64 B per leaf
            del lv
            # FRI commit phase (configs[4] shape): 2^24 LDE domain, f64 quadratic extension, folding 4, rem-deg 31
            from winterfell_amd import fri as wfri

            ev = ctx.to_device(rng.integers(0, fields.M, (1 << 24) * 2, dtype=np.uint64))

            def fri_run():
                # fri/benches/prover.rs: FriProver::build_layers against a DefaultProverChannel (its coin on the device here)
                pr = wfri.FriProver(wfri.FriOptions(8, 4, 31), crypto.Blake3_256, ext_degree=2)
                pr.build_layers(wfri.DefaultProverChannel(1 << 24, 32, crypto.Blake3_256, ext_degree=2), ev)

            ex["fri_build_layers_ms_2^24_quad_fold4_blake3"] = timed(fri_run, 3)

            def fri_run_host_coin():   # the same with the channel's coin on the host: two small device hashes + round trips per layer
                pr = wfri.FriProver(wfri.FriOptions(8, 4, 31), crypto.Blake3_256, ext_degree=2)
                pr.build_layers(wfri.DefaultProverChannel(1 << 24, 32, crypto.Blake3_256, ext_degree=2, device_coin=False), ev)

            ex["fri_build_layers_ms_2^24_quad_fold4_blake3_host_coin"] = timed(fri_run_host_coin, 3)

            # ---- HBM rooflines of the other reported rates: algorithmic bytes (SURVEY 8d / BASELINE.md section 4) over the
            # summed kernel durations (HIP events on the launch stream, wf_prof_*) of one call ----
            def kernel_ms(fn, reps=3):
                fn()
                ctx.prof_enable(True)
                for _ in range(reps):
                    fn()
                prof = ctx.prof_collect()
                ctx.prof_enable(False)
                return sum(ms for _, ms in prof.values()) / reps, {k: round(ms * 1e3 / reps, 1) for k, (c, ms) in prof.items()}

            def roof(alg_bytes, ms, kernels_us, what):
                gbs = alg_bytes / (ms * 1e-3) / 1e9
                return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes": alg_bytes, "kernel_ms": ms, "kernels_us_per_call": kernels_us, "what": what}

            rl = {}
            cm = prover.ColMatrix(ctx.to_device(rng.integers(0, fields.M, (tc, tn), dtype=np.uint64)))
            dom = prover.StarkDomain(tn, tb)
            ms, ks = kernel_ms(lambda: prover.build_trace_commitment(crypto.Blake3_256, cm, dom))
            rl["lde_commit_2^20x4_b8_f64_blake3"] = roof(tn * tc * 8 * (2 + tb) + 64 * tb * tn, ms, ks,
                                                         "n c s (2 + b) + 64 b n: read trace, write polys + LDE + leaves + nodes")
            lv = ctx.to_device(rng.integers(0, 256, (1 << 23, 32), dtype=np.uint8))
            ms, ks = kernel_ms(lambda: crypto.MerkleTree.new(crypto.Blake3_256, lv))
            rl["merkle_blake3_2^23_leaves"] = roof(64 * (1 << 23), ms, ks, "64 B per leaf: read leaves, write nodes")
            del lv
            fri_bytes, ln = 0, 1 << 24
            while ln > 256:                      # FriOptions(blowup 8, folding 4, remainder degree 31): layers down to 2^8 evaluations
                fri_bytes += ln * 16 + (ln // 4) * 16 + 64 * (ln // 4)
                ln //= 4
            ms, ks = kernel_ms(fri_run, 2)
            rl["fri_build_layers_2^24_quad_fold4_blake3"] = roof(fri_bytes, ms, ks,
                                                                 "per layer: len e (read) + len/4 e (folded) + 64 len/4 (leaves + nodes)")
            out["rooflines"] = rl
            del ev
            if sharded:
                ex.update(sharded)
            out["extra"] = ex

        if not args.no_cpu_baseline:
            # ---- CPU baseline (rank 0, N = 1 semantics): the oracle's OpenMP restatement of the reference's `concurrent`
            # (Rayon) algorithms, on this host.  As in math/benches/fft.rs the twiddles are computed once, outside the timed
            # body; the transform runs in place at the bench's own size.  The reference's 4-step
            # transposes are single-threaded (math/src/fft/concurrent.rs:177-218) and so are ours.
            import oracle
            nthreads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
            cn = n
            cp = host.copy()
            tw, itw = oracle.get_twiddles(cn), oracle.get_inv_twiddles(cn)
            oracle.evaluate_poly(cp[:1 << 16], par=True)          # load the library, spin up the OpenMP pool
            t1 = time.perf_counter()
            reps_cpu = 0
            while reps_cpu < 1 or (time.perf_counter() - t1 < 10.0 and reps_cpu < 8):
                oracle.evaluate_poly(cp, par=True, twiddles=tw, inplace=True)
                oracle.interpolate_poly(cp, par=True, twiddles=itw, inplace=True)
                reps_cpu += 1
            cpu_s = time.perf_counter() - t1
            assert np.array_equal(cp, host)
            cpu = {
                "value": 2.0 * cn * reps_cpu / cpu_s, "unit": "elements/s", "cores": nthreads, "kind": "port",
                "sample": "%d x (evaluate_poly + interpolate_poly) in place at 2^%d points, twiddles precomputed; OpenMP restatement of "
                          "math/src/fft/concurrent.rs (oracle/fft_f64.c), %d threads" % (reps_cpu, args.log_n, nthreads),
            }
            if not args.no_extra:
                # the second metric beside its CPU path: trace LDE + commit (spans extend_execution_trace +
                # compute_execution_trace_commitment, trace_lde/default/mod.rs:258-278) and the Merkle build, BLAKE3 (portable C)
                tr = rng.integers(0, fields.M, (4, 1 << 20), dtype=np.uint64)
                t1 = time.perf_counter()
                oracle.build_trace_commitment(0, tr, 8, fields.new(7), par=True)
                cpu["lde_commit_ms_2^20x4_b8_f64_blake3"] = (time.perf_counter() - t1) * 1e3
                lvh = rng.integers(0, 256, (1 << 23, 32), dtype=np.uint8)
                t1 = time.perf_counter()
                oracle.merkle_build(0, lvh, par=True)
                cpu["merkle_blake3_leaves_per_s_2^23"] = (1 << 23) / (time.perf_counter() - t1)
                # FRI commit phase (fri/benches/prover.rs shape of the GPU number above): all cores, DefaultProverChannel
                evh = rng.integers(0, fields.M, (1 << 24) * 2, dtype=np.uint64)
                t1 = time.perf_counter()
                oracle.fri_build_layers_par(0, evh, 4, 8, 31, fields.new(7), 2)
                cpu["fri_build_layers_ms_2^24_quad_fold4_blake3"] = (time.perf_counter() - t1) * 1e3
                del evh
                cpu["extras_note"] = ("one call each (buffers allocated inside the call); concurrent interpolate_columns + 8-column-segment "
                                      "LDE + commit_to_rows + subtree-per-thread Merkle (oracle/commit.c), FRI layers with parallel transpose / "
                                      "row hashing / folding (oracle/fri.c), portable-C BLAKE3 (the Rust crate is AVX2/AVX-512)")
            out["cpu_baseline"] = cpu
        print(json.dumps(out))

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
