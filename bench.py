#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X (contract: see the task statement).

Workload (BASELINE.json configs[1]): standalone 2^24-point forward + inverse NTT over f64, data resident in HBM.
A "step" = one fft::evaluate_poly followed by one fft::interpolate_poly of a 2^24-element vector (natural order in
and out, in place) => 2 * 2^24 element-transforms per step.  value = element-transforms per second, whole job.
With N GPUs every rank transforms its own vector (independent columns shard with no collective): weak scaling.

The ONE stdout line is compact (compact_line, < 6 KB); the full detail goes to --detail (default gpurun_out/bench_detail_n<N>.json).
Fields of the detail object (the line carries the scalar part of each):
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
# issue rates of the two VALU instruction classes of this chip, cycles per wave-instruction on one SIMD (tools/microbench_isa.hip,
# profiles/r02/microbench_isa_*.txt: 2.3-2.7 for plain 32-bit add / sub / and / xor / arithmetic shift / mov, 4.1-4.6 for the rest)
VALU_FAST_CYCLES, VALU_SLOW_CYCLES = 2.3, 4.3


LINE_LIMIT = 6144     # the driver keeps a bounded tail of stdout: the ONE JSON line it parses stays well under it (round 5: 20 KB -> parsed null)
EXTRA_KEYS = (        # the scalars of `extra` that go on the line, in this order, at most ten (the rest: --detail file)
    "lde_commit_ms_2^20x4_b8_f64_blake3", "lde_commit_ms_2^22x32_b8_f64_blake3", "lde_commit_ms_2^24x4_b8_f64_blake3",
    "lde_commit_ms_2^22x64_b8_f128_blake3_p8", "lde_commit_ms_2^19x96_b8_f64_blake3", "lde_commit_ms_2^20x4_b8_f64_rp64",
    "merkle_blake3_leaves_per_s_2^23", "fri_build_layers_ms_2^24_quad_fold4_blake3",
    "rescue_2^20_f128_quad_b8_commit+constraints+composition+deep_ms", "rp64_permutations_per_s")
EXTRA_PREFIXES_N = (  # N > 1: the sharded legs first (their keys carry N, so they are matched by prefix)
    "config3_sharded_commit_", "config4_sharded_fri_", "sharded_lde_commit_ms_", "merkle_blake3_leaves_per_s_2^23_all_ranks",
    "strided_lde_commit_ms_", "sharded_fri_build_layers_ms_", "partitioned_fri_build_layers_ms_", "comm_abi_error")


def _scalar(v):
    return v is None or isinstance(v, (bool, int, float)) or (isinstance(v, str) and len(v) <= 200)


def compact_line(out, detail=None, limit=LINE_LIMIT):
    """The ONE stdout line: the contract's fields, a scalar-only `roofline` and `cpu_baseline`, at most ten scalars of `extra`, the
    fraction of every other roofline case, and where the full detail (every case's kernels, counters, per-thread CPU tables) went.
    Everything on it is copied from `out`; nothing is recomputed.  Raises if the result would not fit `limit` bytes."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_cold", "cold_steps", "spinup_s",
           "spinup_converged", "higher_is_better", "INVALID", "scaling", "vs_baseline", "dtype", "data", "config", "dry_run", "backend")
    line = {k: out[k] for k in top if k in out}
    rf = out.get("roofline")
    if rf:
        keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_transform", "transform_us", "kernel",
                "sclk_mhz_under_load", "reps")
        line["roofline"] = {k: rf[k] for k in keep if k in rf and _scalar(rf[k])}
        v = rf.get("valu") or {}
        for k in ("insts_per_element_per_transform", "valu_busy_frac_at_4_clocks_per_inst", "issue_us_if_every_inst_were_fast",
                  "issue_us_if_every_inst_were_slow"):
            if _scalar(v.get(k)) and v.get(k) is not None:
                line["roofline"]["valu_" + k] = v[k]
        line["roofline"]["limiter"] = "VALU issue, not HBM (DESIGN.md section 5)"
    rls = out.get("rooflines")
    if rls:      # one number per further case: frac of the HBM roofline from the summed kernel durations (detail file: everything else)
        line["roofline_frac_by_case"] = {k: round(v["frac"], 4) for k, v in rls.items() if isinstance(v, dict) and isinstance(v.get("frac"), float)}
    ex = out.get("extra")
    if ex:
        picked = {}
        if out.get("n_gpus", 1) > 1:
            for pre in EXTRA_PREFIXES_N:
                for k in ex:
                    if k.startswith(pre) and (k.endswith("_ms") or "_ms_" in k or "per_s" in k or k == "comm_abi_error") and len(picked) < 10 and _scalar(ex[k]):
                        picked.setdefault(k, ex[k])
        for k in EXTRA_KEYS:
            if k in ex and len(picked) < 10 and _scalar(ex[k]):
                picked.setdefault(k, ex[k])
        line["extra"] = picked
    pc = out.get("pcie")
    if isinstance(pc, dict) and isinstance(pc.get("2^22x32_f64"), dict) and "pipelined_total_ms" in pc["2^22x32_f64"]:
        q = pc["2^22x32_f64"]     # host Vec -> host TracePolyTable + root, PCIe included (never part of `value`)
        line["pcie_2^22x32_f64"] = {k: q[k] for k in ("h2d_trace_ms", "d2h_polys_ms", "d2h_leaves_nodes_ms", "kernels_ms", "serial_total_ms_polys_only",
                                                      "pipelined_total_ms") if k in q}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "host_hardware_threads") if k in cb}
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:400]
    if detail:
        line["detail"] = detail
    s = json.dumps(line)
    if len(s) >= limit:
        raise ValueError("bench line is %d bytes, limit %d" % (len(s), limit))
    return s


def write_detail(out, path):
    """Everything measured, as one JSON object, to `path` (relative to the repository root unless absolute).  Returns the path as
    given, or None when the directory cannot be written (the line is then all there is: never a reason to fail the bench)."""
    try:
        full = path if os.path.isabs(path) else os.path.join(ROOT, path)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        return path
    except OSError:
        return None


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


def spin_up(fn, sync, min_s=2.0, max_s=8.0, tol=0.02, block=8):
    """Bring the chip to the clocks it sustains under `fn` BEFORE anything is timed: the first steps after an idle period run at
    lower clocks (round 3: the driver's `--warmup 5` measured every workload 7-20 % slower than the same binary after 50 steps).
    Runs blocks of `block` calls until at least `min_s` seconds have passed AND three consecutive block times agree within `tol`
    (or `max_s` is reached).  Returns (seconds spent, converged, last block ms per call).  Outside every timed region."""
    t_start = time.perf_counter()
    hist = []
    while True:
        t1 = time.perf_counter()
        for _ in range(block):
            fn()
        sync()
        hist.append((time.perf_counter() - t1) / block)
        el = time.perf_counter() - t_start
        ok = len(hist) >= 3 and max(hist[-3:]) <= (1.0 + tol) * min(hist[-3:])
        if (el >= min_s and ok) or el >= max_s:
            return el, ok, hist[-1] * 1e3


def comm_abi_legs(ctx, dist, rank, world, barrier, timeout_s=240.0, log_rows=22, total_cols=64, fri_log_len=24):
    """BASELINE configs[3] and configs[4] on N ranks through the product's multi-GPU C ABI (include/winterfell_hip.h wf_comm_*,
    INTEGRATION.md section 6): rank 0's wf_comm_get_unique_id travels over the existing process group, every rank calls
    wf_comm_init_rank (RCCL on the context's device), then
      configs[3]: wf_comm_sharded_commit — f128, 64 columns x 2^22 rows, blowup 8, Blake3_256, the columns sharded by partition
                  (PartitionOptions(N, .): 64 / N columns per rank; digest all-to-all + sub-root all-gather on xGMI): STRONG scaling,
      configs[4]: the FRI commit phase of a 2^24-point quadratic-extension LDE sharded by row ranges
                  (parallel.comm_sharded_fri_build_layers: wf_comm_sharded_fri_layers + wf_comm_all_gather + wf_fri_build_layers).
    Every timed call sits between barriers, MAX over ranks; per-rank kernel time comes from the library's HIP events (rank 0's),
    exchanged bytes from the shapes.  The legs run on a watchdog thread: a transport that hangs costs the legs, never the headline."""
    import ctypes
    import threading

    import winterfell_amd  # noqa: F401
    from winterfell_amd import crypto, fri as wfri, parallel
    from winterfell_amd._lib import ptr
    from winterfell_amd.math import fields
    res, lib = {}, ctx.lib

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def body():
        torch.cuda.set_device(ctx.device)                       # the current device is per thread
        # ---- communicator: the 128-byte id over torch's process group, then ncclCommInitRank inside the library
        uid = (ctypes.c_uint8 * 128)()
        if rank == 0:
            st = lib.wf_comm_get_unique_id(uid)
            if st != 0:
                raise RuntimeError("wf_comm_get_unique_id -> %d" % st)
        idt = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device="cuda")
        dist.broadcast(idt, 0)
        uid = (ctypes.c_uint8 * 128)(*idt.cpu().tolist())
        comm = ctypes.c_void_p()
        ctx.use_torch_stream()
        st = lib.wf_comm_init_rank(ctx.handle, uid, rank, world, ctypes.byref(comm))
        if st != 0:
            raise RuntimeError("wf_comm_init_rank -> %d" % st)
        res["transport"] = "RCCL via wf_comm_init_rank, %d ranks" % lib.wf_comm_size(comm)
        try:
            # ---- configs[3]: f128, 64 columns x 2^22 rows sharded by columns
            f = fields.f128
            log_n, log_b = log_rows, 3
            if total_cols % world == 0:
                c = total_cols // world
                n, N = 1 << log_n, 1 << (log_n + log_b)
                g = torch.Generator(device=ctx.device)
                g.manual_seed(0x5EED0400 + rank)
                trace = torch.randint(0, 1 << 62, (c, n * 2), dtype=torch.int64, device=ctx.device, generator=g)
                work = trace.clone()
                rw = int(lib.wf_row_width(c, 1))
                lde, leaves, nodes = ctx.empty_u64(N, rw * f.W), ctx.empty_u8(N // world, 32), ctx.empty_u8(N // world, 32)
                top, root = ctx.empty_u8(world, 32), np.zeros(32, dtype=np.uint8)
                off = f.element_words(f.new(f.GENERATOR))

                def commit():
                    work.copy_(trace)
                    ctx.use_torch_stream()
                    st_ = lib.wf_comm_sharded_commit(comm, crypto.Blake3_256.HASH_ID, f.ID, 1, ptr(work), c, n, log_n, log_b,      # col_stride in base ELEMENTS
                                                     off.ctypes.data_as(ctypes.c_void_p), 0, ptr(lde), ptr(leaves), ptr(nodes), ptr(top),
                                                     root.ctypes.data_as(ctypes.c_void_p))
                    if st_ != 0:
                        raise RuntimeError("wf_comm_sharded_commit -> %d" % st_)

                commit()
                ts = []
                for _ in range(3):
                    barrier()
                    t1 = time.perf_counter()
                    commit()
                    barrier()
                    ts.append((time.perf_counter() - t1) * 1e3)
                key = "config3_sharded_commit_f128_2^%dx%d_b8_blake3_p%d" % (log_n, total_cols, world)
                res[key + "_ms"] = max_over_ranks(float(np.median(ts)))
                ctx.prof_enable(True)
                commit()
                prof = ctx.prof_collect()
                ctx.prof_enable(False)
                res[key + "_rank0_kernel_ms"] = sum(ms for _, ms in prof.values())
                res[key + "_exchanged_bytes_per_rank"] = 32 * N * (world - 1) // world + 32 * world
                roots = [torch.zeros(32, dtype=torch.uint8, device="cuda") for _ in range(world)]
                dist.all_gather(roots, torch.from_numpy(root).cuda())
                res[key + "_roots_agree"] = bool(all(torch.equal(r_, roots[0]) for r_ in roots))
                res[key + "_root"] = bytes(root).hex()
                del trace, work, lde, leaves, nodes
                torch.cuda.empty_cache()
            # ---- configs[4]: FRI commit phase, 2^24-point quadratic extension, folding 4, remainder degree 31
            f64 = fields.f64
            D, log_len = 2, fri_log_len
            piece = ctx.to_device(np.random.default_rng(100 + rank).integers(0, fields.M, ((1 << log_len) // world) * D, dtype=np.uint64))
            fopts = wfri.FriOptions(8, 4, 31)
            coin0 = crypto.DefaultRandomCoin(crypto.Blake3_256, f64, np.zeros(0, dtype=np.uint64), ctx).to_device()
            coin0.draw(1)                                        # uploads the state
            image = coin0.state.clone()
            state = image.clone()

            def fri_run():
                state.copy_(image)
                return parallel.comm_sharded_fri_build_layers(lib, comm, ctx, crypto.Blake3_256, fopts, piece, D, state, min_rows_per_rank=1 << 12)

            out_ = fri_run()
            ts = []
            for _ in range(3):
                barrier()
                t1 = time.perf_counter()
                out_ = fri_run()
                barrier()
                ts.append((time.perf_counter() - t1) * 1e3)
            key = "config4_sharded_fri_2^%d_quad_fold4_blake3_n%d" % (log_len, world)
            res[key + "_ms"] = max_over_ranks(float(np.median(ts)))
            ctx.prof_enable(True)
            fri_run()
            prof = ctx.prof_collect()
            ctx.prof_enable(False)
            res[key + "_rank0_kernel_ms"] = sum(ms for _, ms in prof.values())
            res[key + "_sharded_layers"] = int(out_["num_sharded"])
            ew, ln, xb = 16, 1 << log_len, 0
            for _ in range(int(out_["num_sharded"])):            # re-stride of the layer (all-to-all of equal blocks) + sub-roots
                xb += (ln // world) * ew * (world - 1) // world + 32 * world
                ln //= 4
            res[key + "_exchanged_bytes_per_rank"] = xb + (ln // world) * ew * (world - 1)
            roots = [torch.zeros_like(out_["roots"]) for _ in range(world)]
            dist.all_gather(roots, out_["roots"])
            res[key + "_roots_agree"] = bool(all(torch.equal(r_, roots[0]) for r_ in roots))
        finally:
            lib.wf_comm_destroy(comm)

    err = []

    def guarded():
        try:
            body()
        except BaseException as e:  # noqa: BLE001 - an optional leg must never break the headline
            err.append(repr(e)[:300])

    th = threading.Thread(target=guarded, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        res["comm_abi_error"] = "timed out after %.0f s inside the wf_comm legs" % timeout_s
        res["_hung"] = True
    elif err:
        res["comm_abi_error"] = err[0]
    return res


def pcie_legs(timeout_s=240):
    """SURVEY 8(d) "Timing definition": H2D of the trace and D2H of the polynomials (and of leaves + nodes) reported separately, never
    inside `value` — and what Prover::new_trace_lde (prover/src/lib.rs:182-190) costs a host caller end to end, serial against
    pipelined (wf::new_trace_lde_from_host, include/winterfell_hip.hpp).  Measured by the C++ driver tools/host_pipeline_bench.cpp over
    page-locked host columns, in a process of its own (its three contexts share this GPU after this bench's legs have finished)."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "host_pipeline_bench.bin")
    if not os.path.exists(exe):
        return {"error": "tools/host_pipeline_bench.bin not built (__graft_entry__.build())"}
    res = {}
    for key, argv in (("2^20x4_f64", ["0", "20", "4", "1", "5"]), ("2^22x32_f64", ["0", "22", "32", "1", "3"]),
                      ("2^22x64_f128_p8", ["1", "22", "64", "8", "2"])):
        try:
            r = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=timeout_s)
            res[key] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-200:]}
        except Exception as e:  # noqa: BLE001 - an optional leg
            res[key] = {"error": repr(e)[:200]}
    return res


def stall_breakdown(c):
    """Where a wave's cycles go, from the committed counter passes of the 2^24 transform (sums over its three launches): the three
    disjoint buckets of SQ_WAVE_CYCLES — parked (SQ_WAIT_ANY: s_waitcnt / barrier), stalled at issue (SQ_WAIT_INST_ANY, of which
    SQ_WAIT_INST_LDS is the LDS part) and issuing (the rest) —, and the instruction mix per wave."""
    wc = c.get("SQ_WAVE_CYCLES")
    if not wc:
        return None
    frac = lambda k: (c[k] / wc) if k in c else None
    out = {"of_wave_cycles": {"parked_waitcnt_or_barrier (SQ_WAIT_ANY)": frac("SQ_WAIT_ANY"),
                              "issue_stall (SQ_WAIT_INST_ANY)": frac("SQ_WAIT_INST_ANY"),
                              "issue_stall_lds (SQ_WAIT_INST_LDS, part of the above)": frac("SQ_WAIT_INST_LDS"),
                              "vmem_inst_cycles (SQ_INST_CYCLES_VMEM)": frac("SQ_INST_CYCLES_VMEM"),
                              "active_inst_any (SQ_ACTIVE_INST_ANY)": frac("SQ_ACTIVE_INST_ANY")},
           "raw_per_transform": {k: v for k, v in sorted(c.items())}}
    if c.get("SQ_WAVES"):
        w = c["SQ_WAVES"]
        out["per_wave"] = {k: c[k] / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
                                                  "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT") if k in c}
    if c.get("SQ_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
        out["sq_busy_over_gui_active"] = c["SQ_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--spinup-s", type=float, default=2.0,
                    help="seconds of untimed steps before --warmup, until the step time is steady (0 = none); reported as spinup_s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--detail", default=None,
                    help="file for the full detail (every roofline case, kernels, counters, per-thread CPU tables); "
                         "default gpurun_out/bench_detail_n<N>.json; '-' = none")
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
    dist, ctrl_group = None, None
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
        # a CPU side channel for decisions every rank must take alike when a GPU transport may be wedged (see the wf_comm legs)
        ctrl_group = dist.new_group(backend="gloo") if backend == "nccl" and not args.dry_run else None
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
            print(compact_line({"metric": "f64 NTT elements/s", "value": None, "unit": "elements/s", "n_gpus": world, "dry_run": True,
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

    # the first five steps on a box as it comes (tables built by an untimed step first): reported beside the steady number as
    # ms_per_step_cold, so that what the untimed spin-up below is worth can be read from the line itself
    step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    barrier()
    cold_ms = (time.perf_counter() - t0) * 1e3 / 5
    if dist is not None:
        t = torch.tensor([cold_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cold_ms = float(t.item())
    # steady clocks first (not part of --warmup, not timed): see spin_up
    spinup_s, spinup_ok, _ = spin_up(step, torch.cuda.synchronize, min_s=args.spinup_s) if args.spinup_s > 0 else (0.0, False, None)
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
        "ms_per_step_cold": cold_ms, "cold_steps": 5,
        "spinup_s": round(spinup_s, 3), "spinup_converged": bool(spinup_ok),
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
    hung = False
    if world > 1 and not args.no_extra and backend == "nccl":
        # the product's multi-GPU entry points (C ABI) on BASELINE configs[3] / configs[4]
        sharded = comm_abi_legs(ctx, dist, rank, world, barrier)
        hung = bool(sharded.pop("_hung", False))
        # one rank's failure is every rank's: a rank that threw early leaves the others waiting inside a collective until their
        # watchdog fires, so the ranks agree (over gloo, on the CPU) whether ANY of them is wedged before anybody touches the GPU
        # process group again
        flag = torch.tensor([1 if hung else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=ctrl_group)
        if int(flag.item()) and not hung:
            hung = True
            sharded.setdefault("comm_abi_error", "a peer rank timed out inside the wf_comm legs")
    if world > 1 and not args.no_extra and not hung:
        sharded = sharded or {}
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
            sharded["sharded_lde_commit_ms_2^20x%d_b8_blake3" % (4 * world)] = float(tt.item())
        except Exception as e:  # never let the optional leg break the headline measurement
            sharded["sharded_lde_commit_error"] = repr(e)[:200]
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
        fwd = lambda: fft.evaluate_poly(data)
        spin_up(fwd, torch.cuda.synchronize, min_s=0.5, max_s=3.0)
        # the shader clock WHILE the transform's kernels run: a probe wavefront on its own stream beside 40 queued transforms
        sclk_mhz = None
        try:
            for _ in range(40):
                fwd()
            sclk_mhz = ctx.shader_clock_mhz(2000)
            torch.cuda.synchronize()
        except Exception:
            sclk_mhz = None
        reps = 20
        per_rep = []
        for _ in range(reps):
            ctx.prof_enable(True)
            fwd()
            pr = ctx.prof_collect()
            per_rep.append(pr)
        ctx.prof_enable(False)
        names = sorted({k for pr in per_rep for k in pr})
        kern = {k: {"launches": per_rep[0].get(k, (0, 0.0))[0],
                    "avg_us": float(np.median([pr[k][1] * 1e3 / pr[k][0] for pr in per_rep if k in pr]))} for k in names}
        fwd_us = float(np.median([sum(ms for _, ms in pr.values()) * 1e3 for pr in per_rep]))
        alg_bytes = 2.0 * n * 8                         # read once + write once per transform
        achieved = alg_bytes / (fwd_us * 1e-6) / 1e9
        # HBM bytes per transform and the VALU counters from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2 per
        # the gfx950 correction + WRITE_SIZE, summed over the transform's launches); only valid for the profiled size
        traffic, valu, pmc_round = None, None, None
        try:
            pmc_round = next(r for r in ("r06", "r05", "r04", "r03", "r02", "r01") if os.path.exists(os.path.join(ROOT, "profiles", r, "bench_pmc_summary.json")))
            with open(os.path.join(ROOT, "profiles", pmc_round, "bench_pmc_summary.json")) as f:
                pmf = json.load(f)
            if args.log_n == 24:
                traffic = pmf["ntt_2^24_f64"]["hbm_bytes_per_transform"]
                # the second bound: wave-instructions per element (SQ_INSTS_VALU; on this chip SQ_ACTIVE_INST_VALU reports the same
                # number, i.e. it counts instructions, not cycles), priced at the two issue rates below
                insts = 0.0
                stall = {}
                for kname, cs in pmf["kernels"].items():
                    targs = kname.split("(")[0].split("<", 1)[-1].rstrip("> ").split(", ")    # F, LOG_A, LOG_B, LAST, TWTAB, PF, RH, VT
                    if "ntt_pass<F64, 4, 4" not in kname or "SQ_INSTS_VALU" not in cs or any(a not in ("false", "0") for a in targs[6:]):
                        continue                       # (the rows + leaves and vector-tile variants belong to the LDE, not to a transform)
                    per_transform = 1 if "ntt_pass<F64, 4, 4, true," in kname else 2       # the last pass once, the other shape twice
                    insts += per_transform * cs["SQ_INSTS_VALU"]["avg"] * 64 / n
                    for cname, cv in cs.items():           # every SQ / GRBM counter of the committed passes, summed over the transform's launches
                        if cname.startswith(("SQ_", "GRBM_")):
                            stall[cname] = stall.get(cname, 0.0) + per_transform * cv["avg"]
                if insts:
                    ghz = (sclk_mhz or 2400.0) * 1e-3
                    wave_insts = insts * n / 64.0
                    valu = {"insts_per_element_per_transform": insts, "wave_insts_per_transform": wave_insts,
                            "clock_ghz": ghz, "clock_source": "measured beside the kernels (wf_debug_shader_clock: s_memtime over s_memrealtime)"
                            if sclk_mhz else "assumed (maximum clock; the probe failed)",
                            "issue_us_if_every_inst_were_fast": wave_insts * VALU_FAST_CYCLES / (1024 * ghz * 1e9) * 1e6,
                            "issue_us_if_every_inst_were_slow": wave_insts * VALU_SLOW_CYCLES / (1024 * ghz * 1e9) * 1e6,
                            "measured_us": fwd_us,
                            # SQ_ACTIVE_INST_VALU counts one quad-cycle (4 clocks) per wave-instruction: the share of the kernels' SIMD
                            # time in which the vector ALU is issuing.  Near 1 = bound by instruction issue, whatever the waves wait for.
                            "valu_busy_frac_at_4_clocks_per_inst": wave_insts * 4.0 / (1024 * ghz * 1e9) / (fwd_us * 1e-6),
                            "stall_breakdown": stall_breakdown(stall),
                            "what": "SQ_INSTS_VALU (wave-instructions, summed over the transform's launches) priced at the two issue rates "
                                    "tools/microbench_isa.hip measures on this chip: %.1f cycles per wave-instruction on a SIMD for plain 32-bit "
                                    "add / sub / and / xor / shift / mov, %.1f for everything else (multiply-adds, carry producers and consumers, "
                                    "64-bit and three-operand forms).  The first is a hard lower bound of the kernel time, the second what this "
                                    "stream would cost if none of it were of the fast class; about a third of it is." % (VALU_FAST_CYCLES, VALU_SLOW_CYCLES),
                            "source": "profiles/%s/bench_pmc_summary.json (SQ_INSTS_VALU)" % pmc_round}
        except Exception:
            traffic = None
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_unit": "HBM bytes per transform (rocprofv3 PMC, profiles/%s/bench_pmc_summary.json)" % pmc_round,
            "valu": valu,
            "limiter": "VALU issue, not HBM: a pass executes ~90-115 wave-instructions per element (limb DFTs, two multiply-accumulate "
                       "exits, the Montgomery chain of the twiddle progression; `valu` prices them).  `valu.valu_busy_frac_at_4_clocks_per_inst`: "
                       "the vector ALU issues in ~0.84-0.88 of the kernels' cycles; `valu.stall_breakdown` splits a WAVE's cycles into issuing / "
                       "stalled at issue / parked on a wait or barrier (counters of this bench's own --pmc passes; four isolated transforms give "
                       "38 / 27.5 / 34 % for a non-last pass, profiles/r05/ntt_2p24_stall_counters.json) - with three to four waves per SIMD the "
                       "issue stalls are mostly the other waves' instructions, the LDS part is ~1 %.  Round 4's experiments (DESIGN.md 5.R4): "
                       "with the arithmetic removed the same loads, LDS exchange and stores take 24.5 us per pass (the working set is served by "
                       "the Infinity Cache) against 59; the time follows the instruction count linearly.  A two-pass plan (three-step passes of "
                       "radix 4096, csrc/ntt_big.cuh) was built and measured in round 5: same instruction count, one 1024-lane workgroup per CU, "
                       "229-236 us against 183-193 us at 2^24 (it wins at 2^21 / 2^22: -13 % / -9 %, and is the default there)",
            "kernel": "%s per 2^%d transform; durations summed" % (
                " + ".join("%s x%d" % (k.split("(")[0][:60], v["launches"]) for k, v in kern.items()), args.log_n),
            "algorithmic_bytes_per_transform": alg_bytes, "transform_us": fwd_us, "kernels": kern,
            "sclk_mhz_under_load": sclk_mhz, "reps": reps, "statistic": "median over reps of the summed kernel durations of one transform",
        }

        if not args.no_extra:
            ex = {}

            def timed(fn, reps=10):
                fn()
                torch.cuda.synchronize()
                spin_up(fn, torch.cuda.synchronize, min_s=0.4, max_s=2.5, block=2)      # steady clocks for THIS leg too (untimed)
                reps = max(reps, 10)
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
            last_clock = [None]

            def kernel_ms(fn, reps=10):
                """median over >= 10 calls of the summed kernel durations of one call (HIP events on the launch stream), steady clocks"""
                fn()
                spin_up(fn, torch.cuda.synchronize, min_s=0.4, max_s=2.5, block=2)
                try:                                   # shader clock under THIS leg's load (probe wavefront beside queued calls)
                    fn()
                    fn()
                    last_clock[0] = ctx.shader_clock_mhz(1000)
                    torch.cuda.synchronize()
                except Exception:
                    last_clock[0] = None
                reps = max(reps, 10)
                runs = []
                for _ in range(reps):
                    ctx.prof_enable(True)
                    fn()
                    runs.append(ctx.prof_collect())
                ctx.prof_enable(False)
                tot = float(np.median([sum(ms for _, ms in pr.values()) for pr in runs]))
                names = sorted({k for pr in runs for k in pr})
                return tot, {k: round(float(np.median([pr[k][1] for pr in runs if k in pr])) * 1e3, 1) for k in names}

            # HBM bytes per call of these workloads from the committed counter runs (tools/pmc_workloads.py under rocprofv3 --pmc
            # FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x2 per the gfx950 correction; tools/summarize_workloads_pmc.py)
            wl_traffic, wl_round = {}, None
            try:
                wl_round = next(r for r in ("r06", "r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", r, "workloads_pmc_summary.json")))
                with open(os.path.join(ROOT, "profiles", wl_round, "workloads_pmc_summary.json")) as f:
                    wl_traffic = json.load(f)["workloads"]
            except Exception:
                wl_traffic = {}

            def issue(key, ms):
                """VALU issue time of a workload from its committed counter run, as a bracket: SQ_INSTS_VALU (wave-instructions over
                all dispatches of a call) at the fast class's issue rate (a hard lower bound of the kernel time) and at the slow class's
                (tools/microbench_isa.hip), over 1024 SIMDs at the clock measured under the leg's load.  None without a counter run.
                (Round 3 and the first half of round 4 quoted insts x 4 cycles as a "floor": Rescue then showed 1.2 — not a floor.)"""
                sq = wl_traffic.get(key, {}).get("sq") if key else None
                if not sq or not sq.get("SQ_INSTS_VALU"):
                    return None
                ghz = (last_clock[0] or 2400.0) * 1e-3
                per = lambda cyc: sq["SQ_INSTS_VALU"] * cyc / (1024 * ghz * 1e9) * 1e3
                return {"valu_wave_insts_per_call": sq["SQ_INSTS_VALU"], "clock_ghz": ghz, "clock_measured": last_clock[0] is not None,
                        "issue_ms_if_every_inst_were_fast": per(VALU_FAST_CYCLES), "issue_ms_if_every_inst_were_slow": per(VALU_SLOW_CYCLES),
                        "measured_ms": ms, "measured_over_fast_bound": ms / per(VALU_FAST_CYCLES),
                        "cycles_per_wave_inst": {"fast": VALU_FAST_CYCLES, "slow": VALU_SLOW_CYCLES},
                        "source": "profiles/%s/workloads_pmc_summary.json (SQ_INSTS_VALU)" % wl_round}

            def roof(alg_bytes, ms, kernels_us, what, key=None):
                gbs = alg_bytes / (ms * 1e-3) / 1e9
                tr = wl_traffic.get(key, {}).get("hbm_bytes_per_call") if key else None
                return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes": alg_bytes, "kernel_ms": ms, "kernels_us_per_call": kernels_us, "what": what,
                        "traffic": tr, "traffic_over_algorithmic": (tr / alg_bytes) if tr else None,
                        "traffic_source": ("profiles/%s/workloads_pmc_summary.json" % wl_round) if tr else None,
                        "sclk_mhz_under_load": last_clock[0], "valu": issue(key, ms)}

            rl = {}
            lde_what = "n c s (2 + b) + 64 b n: read trace, write polys + LDE + leaves + nodes"

            def lde_commit_case(key, field, hasher, log_rows, cols, parts=1, reps=3):
                """one wf_build_trace_commitment shape: end-to-end ms into `extra`, kernel-event roofline into `rooflines`"""
                rows = 1 << log_rows
                try:
                    if field is fields.f64:
                        tr_ = ctx.to_device(rng.integers(0, fields.M, (cols, rows), dtype=np.uint64))
                    else:
                        g_ = torch.Generator(device=ctx.device)
                        g_.manual_seed(log_rows * 100 + cols)
                        tr_ = torch.randint(0, 1 << 62, (cols, rows * 2), dtype=torch.int64, device=ctx.device, generator=g_)
                    cm_ = prover.ColMatrix(tr_, field=field)
                    dom_ = prover.StarkDomain(rows, tb, field=field)
                    po_ = prover.PartitionOptions(parts, 1)
                    run_ = lambda: prover.build_trace_commitment(hasher, cm_, dom_, po_)
                    ex["lde_commit_ms_" + key] = timed(run_, reps)
                    ms_, ks_ = kernel_ms(run_, 2)
                    rl["lde_commit_" + key] = roof(rows * cols * 8 * field.W * (2 + tb) + 64 * tb * rows, ms_, ks_, lde_what, "lde_commit_" + key)
                except Exception as e:  # an allocation failure on a smaller part must not cost the line
                    ex["lde_commit_" + key + "_error"] = repr(e)[:160]
                torch.cuda.empty_cache()

            lde_commit_case("2^20x4_b8_f64_blake3", fields.f64, crypto.Blake3_256, 20, 4)
            # BASELINE's metric range "2^20 - 2^24 rows"  and configs[3] on ONE GPU (M4: f128, 64 columns x 2^22 rows,
            # PartitionOptions(8, .)): the widths the reference's row_matrix bench uses
            lde_commit_case("2^22x32_b8_f64_blake3", fields.f64, crypto.Blake3_256, 22, 32, reps=2)
            lde_commit_case("2^24x4_b8_f64_blake3", fields.f64, crypto.Blake3_256, 24, 4, reps=2)
            lde_commit_case("2^22x64_b8_f128_blake3_p8", fields.f128, crypto.Blake3_256, 22, 64, parts=8, reps=2)
            # the rest of the reference's row_matrix bench widths (prover/benches/row_matrix.rs: 2^19 rows x 64 / 96 columns)
            lde_commit_case("2^19x64_b8_f64_blake3", fields.f64, crypto.Blake3_256, 19, 64, reps=2)
            lde_commit_case("2^19x96_b8_f64_blake3", fields.f64, crypto.Blake3_256, 19, 96, reps=2)

            # standalone 2^20-point transforms over the other fields of math/benches/fft.rs:16-116 (f128, f62, and the quadratic /
            # cubic extensions of f64): forward transform, kernel events, algorithmic bytes 2 n e (read once, write once)
            def ntt_case(key, field, D, log_rows=20):
                try:
                    rows = 1 << log_rows
                    if field is fields.f64:
                        d_ = ctx.to_device(rng.integers(0, fields.M, rows * D, dtype=np.uint64))
                    elif field is fields.f128:
                        d_ = ctx.to_device(rng.integers(0, 1 << 62, rows * D * 2, dtype=np.uint64))
                    else:
                        d_ = ctx.to_device(rng.integers(0, 1 << 61, rows * D, dtype=np.uint64))
                    run_ = lambda: fft.evaluate_poly(d_, ext_degree=D, field=field)
                    ms_, ks_ = kernel_ms(run_, 10)
                    rl["ntt_" + key] = roof(2.0 * rows * D * 8 * field.W, ms_, ks_, "2 n e: read once, write once (one forward transform, in place)",
                                            "ntt_" + key)
                except Exception as e:
                    ex["ntt_" + key + "_error"] = repr(e)[:160]

            ntt_case("2^20_f128", fields.f128, 1)
            ntt_case("2^20_f62", fields.f62, 1)
            ntt_case("2^20_f64_quad", fields.f64, 2)
            ntt_case("2^20_f64_cubic", fields.f64, 3)
            # Rescue: hopeless against HBM, so its rate is permutations per second against the measured modular-multiplication
            # ceiling.  2^20 x 4, blowup 8: one permutation per row (4 elements < rate 8) + one per Merkle merge.
            perms = 2.0 * (1 << 23) - 1
            cm_r = prover.ColMatrix(ctx.to_device(rng.integers(0, fields.M, (tc, tn), dtype=np.uint64)))
            ms_r, ks_r = kernel_ms(lambda: prover.build_trace_commitment(crypto.Rp64_256, cm_r, prover.StarkDomain(tn, tb)), 2)
            hash_ms = sum(v for k, v in ks_r.items() if "ntt" not in k and "transpose" not in k) * 1e-3
            rl["lde_commit_2^20x4_b8_f64_rp64"] = {
                "bound": "valu", "kernel_ms": ms_r, "kernels_us_per_call": ks_r, "permutations": perms, "hash_kernel_ms": hash_ms,
                "rp64_permutations_per_s": perms / (hash_ms * 1e-3), "sclk_mhz_under_load": last_clock[0],
                "valu": issue("lde_commit_2^20x4_b8_f64_rp64", ms_r),
                "what": "Rp64_256: one permutation per row (4 elements < rate 8) + one per Merkle merge.  The bound is VALU issue: `valu` "
                        "brackets the issue time of the whole call from its counter run (SQ_INSTS_VALU at the fast and at the slow class's "
                        "issue rate, 1024 SIMDs at the measured clock): the kernel time must lie above the first and lies below the second "
                        "when a good part of the stream is plain 32-bit arithmetic.  (Round 3 quoted a 'modmul ceiling' fraction of 1.05, "
                        "the first half of round 4 'insts x 4 cycles' with a fraction of 1.2: neither was a ceiling, both dropped.)"}
            ex["rp64_permutations_per_s"] = rl["lde_commit_2^20x4_b8_f64_rp64"]["rp64_permutations_per_s"]
            del cm_r
            lv = ctx.to_device(rng.integers(0, 256, (1 << 23, 32), dtype=np.uint8))
            ms, ks = kernel_ms(lambda: crypto.MerkleTree.new(crypto.Blake3_256, lv))
            rl["merkle_blake3_2^23_leaves"] = roof(64 * (1 << 23), ms, ks, "64 B per leaf: read leaves, write nodes", "merkle_blake3_2^23_leaves")
            del lv
            fri_bytes, ln = 0, 1 << 24
            while ln > 256:                      # FriOptions(blowup 8, folding 4, remainder degree 31): layers down to 2^8 evaluations
                fri_bytes += ln * 16 + (ln // 4) * 16 + 64 * (ln // 4)
                ln //= 4
            ms, ks = kernel_ms(fri_run, 2)
            rl["fri_build_layers_2^24_quad_fold4_blake3"] = roof(fri_bytes, ms, ks,
                                                                 "per layer: len e (read) + len/4 e (folded) + 64 len/4 (leaves + nodes)",
                                                                 "fri_build_layers_2^24_quad_fold4_blake3")
            # the same call with ONE event pair around all its launches (wf_prof_enable(ctx, 2)): sixteen launches, most of them a few
            # microseconds long, so the per-launch brackets above and the span are reported side by side
            spans = []
            for _ in range(3):
                ctx.prof_enable(2)
                fri_run()
                sp = ctx.prof_collect().get("__span__")
                if sp:
                    spans.append(sp)
            ctx.prof_enable(False)
            if spans:
                rl["fri_build_layers_2^24_quad_fold4_blake3"]["gpu_span_ms"] = float(np.median([m for _, m in spans]))
                rl["fri_build_layers_2^24_quad_fold4_blake3"]["launches"] = int(spans[0][0])
            out["rooflines"] = rl
            del ev
            if sharded:
                ex.update(sharded)
            out["extra"] = ex

        if not args.no_cpu_baseline and world == 1:
            # ---- CPU baseline (rank 0, N = 1 only: the N > 1 runs are about scaling, and the other ranks would wait for it): the oracle's OpenMP restatement of the reference's `concurrent`
            # (Rayon) algorithms, on this host.  As in math/benches/fft.rs the twiddles are computed once, outside the timed
            # body; the transform runs in place at the bench's own size.  The reference's 4-step
            # transposes are single-threaded (math/src/fft/concurrent.rs:177-218) and so are ours.
            import oracle
            ncpu = os.cpu_count() or 1
            env_threads = int(os.environ.get("OMP_NUM_THREADS", "0"))
            # the reference's Rayon pool is as wide as the host; a 256-thread team is not automatically the fastest for these sizes, so
            # every number is the BEST over team sizes {8, 32, 64, all} (round-3 review), buffers allocated and touched before the clock
            # starts, one untimed call first
            teams = sorted({t for t in (8, 32, 64, ncpu) if t <= ncpu}) if not env_threads else [env_threads]
            cn = n
            cp = host.copy()
            tw, itw = oracle.get_twiddles(cn), oracle.get_inv_twiddles(cn)
            oracle.evaluate_poly(cp[:1 << 16], par=True)          # load the library, spin up the OpenMP pool
            best = None
            per_team = {}
            for tcount in teams:
                oracle.set_num_threads(tcount)
                oracle.evaluate_poly(cp, par=True, twiddles=tw, inplace=True)         # untimed: pool of this size, pages, caches
                oracle.interpolate_poly(cp, par=True, twiddles=itw, inplace=True)
                t1 = time.perf_counter()
                reps_cpu = 0
                while reps_cpu < 1 or (time.perf_counter() - t1 < 2.5 and reps_cpu < 4):
                    oracle.evaluate_poly(cp, par=True, twiddles=tw, inplace=True)
                    oracle.interpolate_poly(cp, par=True, twiddles=itw, inplace=True)
                    reps_cpu += 1
                rate = 2.0 * cn * reps_cpu / (time.perf_counter() - t1)
                per_team[str(tcount)] = rate
                if best is None or rate > best[0]:
                    best = (rate, tcount, reps_cpu)
            assert np.array_equal(cp, host)
            cpu = {
                "value": best[0], "unit": "elements/s", "cores": best[1], "kind": "port", "host_hardware_threads": ncpu,
                "elements_per_s_by_threads": per_team,
                "sample": "best of team sizes %s: %d x (evaluate_poly + interpolate_poly) in place at 2^%d points after one untimed pair, twiddles "
                          "precomputed; OpenMP restatement of math/src/fft/concurrent.rs (oracle/fft_f64.c)" % (teams, best[2], args.log_n),
            }
            if not args.no_extra:
                # the second metric beside its CPU path: trace LDE + commit (spans extend_execution_trace +
                # compute_execution_trace_commitment, trace_lde/default/mod.rs:258-278), the Merkle build and the FRI commit phase, BLAKE3
                # (portable C).  Result buffers are allocated once and touched by an untimed call; best over team sizes.
                tr = rng.integers(0, fields.M, (4, 1 << 20), dtype=np.uint64)
                lvh = rng.integers(0, 256, (1 << 23, 32), dtype=np.uint8)
                evh = rng.integers(0, fields.M, (1 << 24) * 2, dtype=np.uint64)
                bufs = oracle.build_trace_commitment(0, tr, 8, fields.new(7), par=True)
                nodes_h = oracle.merkle_build(0, lvh, par=True)
                res_cpu = {"lde_commit_ms_2^20x4_b8_f64_blake3": {}, "merkle_blake3_leaves_per_s_2^23": {}, "fri_build_layers_ms_2^24_quad_fold4_blake3": {}}
                for tcount in teams:
                    oracle.set_num_threads(tcount)
                    t1 = time.perf_counter()
                    oracle.build_trace_commitment(0, tr, 8, fields.new(7), par=True, out=bufs)
                    res_cpu["lde_commit_ms_2^20x4_b8_f64_blake3"][str(tcount)] = (time.perf_counter() - t1) * 1e3
                    t1 = time.perf_counter()
                    oracle.merkle_build(0, lvh, par=True, out=nodes_h)
                    res_cpu["merkle_blake3_leaves_per_s_2^23"][str(tcount)] = (1 << 23) / (time.perf_counter() - t1)
                # FRI commit phase (fri/benches/prover.rs shape of the GPU number above): DefaultProverChannel; one untimed call, then the
                # two widest teams (its layers are allocated inside the call: the arena keeps them between calls)
                oracle.fri_build_layers_par(0, evh, 4, 8, 31, fields.new(7), 2)
                for tcount in teams[-2:]:
                    oracle.set_num_threads(tcount)
                    t1 = time.perf_counter()
                    oracle.fri_build_layers_par(0, evh, 4, 8, 31, fields.new(7), 2)
                    res_cpu["fri_build_layers_ms_2^24_quad_fold4_blake3"][str(tcount)] = (time.perf_counter() - t1) * 1e3
                oracle.set_num_threads(ncpu)
                del evh
                for k, by_t in res_cpu.items():
                    pick = max if "per_s" in k else min
                    bt = pick(by_t, key=lambda q: by_t[q]) if "per_s" in k else min(by_t, key=lambda q: by_t[q])
                    cpu[k] = by_t[bt]
                    cpu[k + "_threads"] = int(bt)
                    cpu[k + "_by_threads"] = by_t
                cpu["extras_note"] = ("best over the team sizes, result buffers allocated and touched before the clock starts (one untimed call); "
                                      "concurrent interpolate_columns + 8-column-segment LDE + commit_to_rows + subtree-per-thread Merkle "
                                      "(oracle/commit.c), FRI layers with parallel transpose / row hashing / folding (oracle/fri.c), portable-C "
                                      "BLAKE3 (the Rust crate is AVX2/AVX-512).  The reference publishes 2.5 s for the WHOLE f128 rescue proof of "
                                      "2^20 rows on 8 laptop cores (README.md:411-465); these are f64 stages on this host")
            out["cpu_baseline"] = cpu
        if not args.no_extra and world == 1:
            out["pcie"] = pcie_legs()
        # the full detail to a file, ONE compact line (< LINE_LIMIT bytes) to stdout, last
        dpath = args.detail if args.detail is not None else os.path.join("gpurun_out", "bench_detail_n%d.json" % world)
        dpath = write_detail(out, dpath) if dpath != "-" else None
        sys.stdout.flush()
        print(compact_line(out, dpath))
        sys.stdout.flush()

    if hung:
        # a thread is still inside a collective that never completed: no orderly shutdown is possible
        sys.stdout.flush()
        os._exit(0)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
