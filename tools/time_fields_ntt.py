"""Forward-NTT time per field (2^22 points, single vector) through the Python binding: wf_prof per-kernel events."""
import sys

import numpy as np

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd.math import fft, fields

ctx = winterfell_amd.default_context(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
rng = np.random.default_rng(0)
for f in (fields.f64, fields.f62, fields.f128):
    n = 1 << log_n
    host = rng.integers(0, 1 << 61, n * f.W, dtype=np.uint64)
    d = ctx.to_device(host)
    fft.evaluate_poly(d, None, field=f)
    ctx.sync()
    ctx.prof_enable(True)
    for _ in range(10):
        fft.evaluate_poly(d, None, field=f)
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    tot = sum(ms for _, ms in prof.values()) / 10
    print("%-5s 2^%d forward NTT: %8.1f us  (%.2f ns/element)  %s" % (f.name, log_n, tot * 1e3, tot * 1e6 / n,
          {k: round(v[1] / 10 * 1e3, 1) for k, v in prof.items()}))
