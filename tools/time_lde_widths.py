"""trace LDE + commit (wf_build_trace_commitment: interpolate -> coset LDE -> row hashes -> Merkle tree) over
the reference's row_matrix bench widths, f64, blowup 8, Blake3_256 — ms per call, algorithmic bytes n*c*s*(2+b) + 64*b*n,
and the resulting fraction of the 8 TB/s HBM roofline.   python tools/time_lde_widths.py [max_log_n=24]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd import crypto, prover

ctx = winterfell_amd.default_context(0)
max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2)
print("%-6s %-5s %10s %10s %8s   %s" % ("log_n", "cols", "ms", "alg GB", "TB/s", "kernels (ms)"))
for log_n, widths in ((20, (4, 32, 64, 96)), (22, (4, 32, 64, 96)), (24, (4, 32))):
    if log_n > max_log:
        continue
    n, b = 1 << log_n, 8
    for c in widths:
        trace = torch.from_numpy(rng.integers(0, 1 << 62, (c, n), dtype=np.int64)).to(ctx.device)
        dom = prover.StarkDomain(n, b)
        run = lambda: prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(trace.clone(), 1, ctx), dom)
        out = run()
        del out
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = run()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
            del out
        ctx.prof_enable(True)
        out = run()
        agg = ctx.prof_collect()
        ctx.prof_enable(False)
        del out
        ms = float(np.median(ts))
        alg = n * c * 8 * (2 + b) + 64 * b * n
        print("%-6d %-5d %10.3f %10.2f %8.2f   %s" % (log_n, c, ms, alg / 1e9, alg / ms / 1e9,
              " ".join("%s=%.2f" % (k, v[1]) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]))))
        del trace
        torch.cuda.empty_cache()
