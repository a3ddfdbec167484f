"""Two-pass (ntt_big.cuh: three-step passes of radix 2^10 .. 2^12) against three-pass plans, same box, alternating runs:
single transforms 2^20 .. 2^24 (kernel time from the library's per-launch events) and trace LDE + commit over wide f64 traces.
WF_NTT_BIG is read once per context, so each plan gets a context of its own.   python tools/time_two_pass.py [reps=5]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd import crypto, prover
from winterfell_amd._lib import Context
from winterfell_amd.math import fft

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = sys.argv[2] if len(sys.argv) > 2 else ""          # "lde22": only the 2^22-row LDE + commit shapes
base = winterfell_amd.default_context(0)


def make(**env):
    for k, v in env.items():
        os.environ[k] = v
    c = Context(0)
    for k in env:
        del os.environ[k]
    return c


# r04 = the round-4 behaviour: three passes, wide rows hashed by the separate kernel
ctxs = {"r04": make(WF_NTT_BIG="0", WF_ROWS_HASH_WIDE="0"), "two-pass": make(WF_NTT_BIG="1", WF_ROWS_HASH_WIDE="0"),
        "rows+hash": make(WF_NTT_BIG="0"), "default": make(), "linear-order": make(WF_NTT_COSET_ORDER="0")}
# spin the clocks up
x = torch.from_numpy(np.random.default_rng(1).integers(0, 1 << 62, 1 << 24, dtype=np.int64)).to(base.device)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    fft.evaluate_poly(x, ctx=base)
torch.cuda.synchronize()


def kernel_ms(ctx, fn):
    ctx.prof_enable(True)
    out = fn()
    agg = ctx.prof_collect()
    ctx.prof_enable(False)
    del out
    return sum(v[1] for v in agg.values()), agg


print("== single transforms (kernel us, median of %d) ==" % reps)
for log_n in (() if only else (20, 21, 22, 23, 24)):
    n = 1 << log_n
    d = torch.from_numpy(np.random.default_rng(log_n).integers(0, 1 << 62, n, dtype=np.int64)).to(base.device)
    row = []
    for name, ctx in ctxs.items():
        if name in ("rows+hash", "linear-order"):
            continue
        for _ in range(3):
            fft.evaluate_poly(d, ctx=ctx)
        ts, last = [], None
        for _ in range(reps):
            ms, agg = kernel_ms(ctx, lambda: fft.evaluate_poly(d, ctx=ctx))
            ts.append(ms)
            last = agg
        row.append("%s %.1f us (%s)" % (name, 1e3 * float(np.median(ts)), " ".join("%s=%.1f" % (k, 1e3 * v[1] / v[0]) for k, v in sorted(last.items()))))
    print("2^%d: %s" % (log_n, " | ".join(row)))
    del d

print("== trace LDE + commit, f64, blowup 8, Blake3_256 (wall ms, median of %d; kernels of the last run) ==" % reps)
for log_n, c in (((22, 32), (22, 24)) if only == "lde22" else ((20, 32), (22, 32), (20, 96), (22, 16), (20, 4))):
    n, b = 1 << log_n, 8
    trace = torch.from_numpy(np.random.default_rng(7).integers(0, 1 << 62, (c, n), dtype=np.int64)).to(base.device)
    dom = prover.StarkDomain(n, b)
    roots = {}
    for name, ctx in ctxs.items():
        run = lambda: prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(trace.clone(), 1, ctx), dom)
        out = run()
        roots[name] = out[1].root().tobytes()
        del out
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = run()
            ctx.sync()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
            del out
        ms, agg = kernel_ms(ctx, run)
        alg = n * c * 8 * (2 + b) + 64 * b * n
        print("2^%d x %d %-10s wall %.2f ms  kernels %.2f ms  frac %.3f   %s" % (log_n, c, name, float(np.median(ts)), ms, alg / (np.median(ts) * 1e-3) / 8e12,
              " ".join("%s=%.2f" % (k, v[1]) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]))))
        ctx.call("wf_ctx_trim")
    assert len(set(roots.values())) == 1, "the plans disagree on the Merkle root"
    del trace
    torch.cuda.empty_cache()
