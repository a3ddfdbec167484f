"""Per-kernel times of batched f64 transforms: `batch` vectors of 2^log_n points (the shape of the LDE passes), plain
evaluate_poly.  usage: python tools/time_batch_ntt.py [log_n=20] [batch=32]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import winterfell_amd
from winterfell_amd.math import fft, fields
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = winterfell_amd.default_context(0)
x = ctx.to_device(np.random.default_rng(1).integers(0, fields.M, batch << log_n, dtype=np.uint64))
fft.evaluate_poly(x, batch=batch); torch.cuda.synchronize()
ctx.prof_enable(True)
for _ in range(10): fft.evaluate_poly(x, batch=batch)
agg = ctx.prof_collect(); ctx.prof_enable(False)
tot = 0.0
for k, (c, ms) in agg.items():
    print("%-16s %3d launches  %8.1f us each" % (k, c // 10, ms * 1e3 / c)); tot += ms / 10
print("2^%d x %d: %.1f us per batch transform, %.2f TB/s algorithmic" % (log_n, batch, tot * 1e3, 16.0 * (batch << log_n) / (tot * 1e-3) / 1e12))
