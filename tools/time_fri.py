"""FRI commit phase over a 2^24-point LDE domain (fri/benches/prover.rs shape), per-kernel HIP-event totals.
   python tools/time_fri.py [log_len=24]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd import crypto, fri
from winterfell_amd.math import fields

ctx = winterfell_amd.default_context(0)
log_len = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(4)


class Chan:
    def __init__(self, f, D):
        self.k, self.f, self.D = 0, f, D

    def commit_fri_layer(self, root):
        self.k += 1

    def draw_fri_alpha(self):
        return self.f.pack([self.f.new(1234 + self.k + d) for d in range(self.D)])


for fname, D, N, hname in (("f64", 2, 4, "Blake3_256"), ("f64", 2, 8, "Blake3_256"), ("f128", 1, 4, "Blake3_256"), ("f64", 3, 4, "Blake3_256"), ("f64", 2, 4, "Rp64_256")):
    f = getattr(fields, fname)
    hasher = getattr(crypto, hname)
    ev = ctx.to_device(rng.integers(0, 1 << 62, (1 << log_len) * D * f.W, dtype=np.uint64))
    opts = fri.FriOptions(8, N, 31, field=f)
    run = lambda: fri.FriProver(opts, hasher, ext_degree=D, ctx=ctx).build_layers(Chan(f, D), ev)
    run()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    ctx.prof_enable(True)
    run()
    agg = ctx.prof_collect()
    ctx.prof_enable(False)
    print("%s D=%d fold %d %s 2^%d: %.3f ms | " % (fname, D, N, hname, log_len, float(np.median(ts))),
          " ".join("%s=%.3f(%d)" % (k, v[1], v[0]) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])))
    del ev
