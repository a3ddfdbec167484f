"""One workload for counter runs: K forward 2^log_n-point f64 transforms after a warm-up.  python tools/wl_ntt.py [log_n=24] [K=3]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd.math import fft, fields

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = winterfell_amd.default_context(0)
d = ctx.to_device(np.random.default_rng(1).integers(0, fields.M, 1 << log_n, dtype=np.uint64))
for _ in range(K + 1):
    fft.evaluate_poly(d)
ctx.sync()
