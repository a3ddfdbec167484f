"""Per-pass kernel time (library HIP events) of a 2^24-point f64 forward NTT, median of 20: for A/B runs of variant libraries
(WF_HIP_LIBRARY=...), including the timing-experiment builds whose results are wrong by construction (no parity check here)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd.math import fft, fields

ctx = winterfell_amd.default_context(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
data = ctx.to_device(np.random.default_rng(0).integers(0, fields.M, 1 << log_n, dtype=np.uint64))
for _ in range(300):
    fft.evaluate_poly(data)
torch.cuda.synchronize()
runs = []
for _ in range(20):
    ctx.prof_enable(True)
    fft.evaluate_poly(data)
    runs.append(ctx.prof_collect())
    ctx.prof_enable(False)
names = sorted({k for r in runs for k in r})
tot = float(np.median([sum(ms for _, ms in r.values()) for r in runs])) * 1e3
print("transform %.1f us  " % tot + "  ".join("%s x%d %.1f us" % (k, runs[0][k][0], float(np.median([r[k][1] / r[k][0] for r in runs])) * 1e3) for k in names))
