"""Plan selection, pinned by measurement: for every shape of a grid the DEFAULT context's plan is timed against the
forced alternatives on the same box, interleaved, and the table goes to profiles/rNN/plan_sweep.csv.

  trace LDE + commit (wf_build_trace_commitment, f64, Blake3_256): 2^18 .. 2^24 rows x {1, 4, 8, 16, 32, 64, 96} columns x blowup {2, 4, 8}
  single / batched transforms (fft::evaluate_poly): 2^18 .. 2^24 points x {1, 8, 32} vectors

Variants (each a context of its own; the environment is read once, by wf_ctx_create):
  default                       what a caller gets
  three-pass                    WF_NTT_BIG=0              radix <= 256 passes only
  two-pass                      WF_NTT_BIG=1              three-step passes of radix 2^10 .. 2^12 wherever eligible
  separate-row-hash             WF_ROWS_HASH_WIDE=0       rows of 9 .. 32 columns hashed by hash_rows_wide, not by the last pass
  three-pass+separate-row-hash  both
  f64-tables                    WF_NTT_F64_TABLES=1       inter-pass twiddles from one-word tables instead of the per-lane progression
  f64-tables+three-pass         with WF_NTT_BIG=0
  no-vector-tiles               WF_LDE_VT=0               the wide-trace LDE on position-major tiles (per-lane twiddle progressions), as before round 6

  python tools/plan_sweep.py [out.csv] [reps=5] [max_lde_gib=24] [quick]

Every variant must produce the same Merkle root (checked).  Timing: wall clock around the call + sync, minimum of `reps` after one untimed
call per variant, variants interleaved repetition by repetition (so that a clock drift hits all of them alike)."""
import csv
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import winterfell_amd  # noqa: E402
from winterfell_amd import crypto, prover  # noqa: E402
from winterfell_amd._lib import Context  # noqa: E402
from winterfell_amd.math import fft  # noqa: E402

VARIANTS = (("default", {}), ("three-pass", {"WF_NTT_BIG": "0"}), ("two-pass", {"WF_NTT_BIG": "1"}),
            ("separate-row-hash", {"WF_ROWS_HASH_WIDE": "0"}), ("three-pass+separate-row-hash", {"WF_NTT_BIG": "0", "WF_ROWS_HASH_WIDE": "0"}),
            ("f64-tables", {"WF_NTT_F64_TABLES": "1"}), ("f64-tables+three-pass", {"WF_NTT_F64_TABLES": "1", "WF_NTT_BIG": "0"}),
            ("no-vector-tiles", {"WF_LDE_VT": "0"}))


def make_contexts(device=0):
    out = {}
    for name, env in VARIANTS:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            out[name] = Context(device)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return out


def time_variants(ctxs, run, reps, check=None):
    """run(ctx) -> object; returns {variant: min ms}.  One untimed call each (and `check` on its result), then interleaved repetitions."""
    ts = {name: [] for name in ctxs}
    for name, ctx in ctxs.items():
        out = run(ctx)
        ctx.sync()
        if check is not None:
            check(name, out)
        del out
    for _ in range(reps):
        for name, ctx in ctxs.items():
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = run(ctx)
            ctx.sync()
            ts[name].append((time.perf_counter() - t) * 1e3)
            del out
    return {name: float(np.min(v)) for name, v in ts.items()}


def lde_shapes(quick):
    if quick:   # the shapes BETWEEN the ones earlier rounds measured (tests/test_gpu_plan_default.py)
        return [(21, 8, 8), (21, 16, 8), (21, 32, 8), (23, 4, 8), (23, 16, 4), (20, 16, 2), (22, 8, 4), (19, 32, 8), (20, 64, 4), (22, 4, 2), (18, 96, 8), (24, 8, 2)]
    return [(L, c, b) for L in range(18, 25) for c in (1, 4, 8, 16, 32, 64, 96) for b in (2, 4, 8)]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/plan_sweep.csv"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    max_gib = float(sys.argv[3]) if len(sys.argv) > 3 else 24.0
    quick = len(sys.argv) > 4 and sys.argv[4] == "quick"
    base = winterfell_amd.default_context(0)
    x = torch.from_numpy(np.random.default_rng(1).integers(0, 1 << 62, 1 << 24, dtype=np.int64)).to(base.device)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:          # clocks up
        fft.evaluate_poly(x, ctx=base)
    torch.cuda.synchronize()
    del x
    rows = []
    g = torch.Generator(device=base.device)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    fcsv = open(out_path, "w", newline="")                       # written row by row: a late failure keeps what was measured
    wcsv = csv.writer(fcsv)
    wcsv.writerow(["workload", "log_rows", "cols_or_vectors", "blowup"] + ["ms_" + v for v, _ in VARIANTS] + ["best", "default_over_best"])

    def record(row):
        rows.append(row)
        wcsv.writerow(row)
        fcsv.flush()
        print(row, flush=True)
    for L, c, b in lde_shapes(quick):
        n = 1 << L
        rw = 8 * ((c + 7) // 8)
        if n * b * rw * 8 / 2**30 > max_gib:
            continue
        g.manual_seed(L * 1000 + c * 10 + b)
        trace = torch.randint(0, 1 << 62, (c, n), dtype=torch.int64, device=base.device, generator=g)
        dom = prover.StarkDomain(n, b)
        roots = {}
        ctxs = make_contexts(0)          # per shape: a context keeps its grow-only scratch (seven of them at 2^23 x 96 x 8: out of memory)

        def run(ctx):
            return prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(trace.clone(), 1, ctx), dom)

        def check(name, out):
            roots[name] = out[1].root().tobytes()

        ms = time_variants(ctxs, run, reps, check)
        assert len(set(roots.values())) == 1, "the plans disagree on the Merkle root at 2^%d x %d, blowup %d" % (L, c, b)
        best = min(ms, key=ms.get)
        record(["lde_commit", L, c, b] + [round(ms[v], 4) for v, _ in VARIANTS] + [best, round(ms["default"] / ms[best], 4)])
        del trace
        for ctx in ctxs.values():
            ctx.sync()
            ctx.close()
        torch.cuda.empty_cache()
    ctxs = make_contexts(0)
    nt_ctxs = {k: v for k, v in ctxs.items() if k in ("default", "three-pass", "two-pass", "f64-tables", "f64-tables+three-pass")}
    for L in (range(18, 25) if not quick else (21, 23)):
        for nvec in (1, 8, 32):
            n = 1 << L
            if n * nvec * 8 / 2**30 > max_gib:
                continue
            g.manual_seed(L * 100 + nvec)
            d = torch.randint(0, 1 << 62, (nvec * n,), dtype=torch.int64, device=base.device, generator=g)
            ref = {}

            def run(ctx):
                return fft.evaluate_poly(d.clone(), ctx=ctx, batch=nvec) if nvec > 1 else fft.evaluate_poly(d.clone(), ctx=ctx)

            def check(name, out):
                ref[name] = out.clone()

            ms = time_variants(nt_ctxs, run, reps, check)
            assert all(torch.equal(v, ref["default"]) for v in ref.values()), "the plans disagree at 2^%d x %d" % (L, nvec)
            best = min(ms, key=ms.get)
            record(["evaluate_poly", L, nvec, 1] + [round(ms.get(v, float("nan")), 4) for v, _ in VARIANTS] + [best, round(ms["default"] / ms[best], 4)])
            del d, ref
    fcsv.close()
    worst = max(rows, key=lambda r: r[-1])
    print("worst default/best: %.3f at %s" % (worst[-1], worst[:4]))


if __name__ == "__main__":
    main()
