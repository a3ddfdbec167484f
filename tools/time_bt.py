"""Block tiles (single three-pass f64 transforms: transposed first pass, tile-shared twiddles in the middle pass) against the standard
plans, same box: parity (forward, inverse, round trip) and kernel time per transform.  python tools/time_bt.py [reps=20]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import winterfell_amd  # noqa: E402
from winterfell_amd._lib import Context  # noqa: E402
from winterfell_amd.math import fft  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
base = winterfell_amd.default_context(0)


def make(**env):
    os.environ.update(env)
    try:
        return Context(0)
    finally:
        for k in env:
            del os.environ[k]


ctxs = {"bt": make(WF_NTT_BT="1", WF_NTT_BIG="0"), "three-pass": make(WF_NTT_BT="0", WF_NTT_BIG="0"), "default": make()}
x = torch.randint(0, 1 << 62, (1 << 24,), dtype=torch.int64, device=base.device)
for _ in range(300):
    fft.evaluate_poly(x, ctx=base)
torch.cuda.synchronize()
for L in (18, 19, 20, 21, 22, 23, 24):
    d = torch.randint(0, 1 << 62, (1 << L,), dtype=torch.int64, device=base.device)
    ref_f = fft.evaluate_poly(d.clone(), ctx=ctxs["three-pass"])
    ref_i = fft.interpolate_poly(d.clone(), ctx=ctxs["three-pass"])
    row = []
    for name, ctx in ctxs.items():
        f = fft.evaluate_poly(d.clone(), ctx=ctx)
        i = fft.interpolate_poly(d.clone(), ctx=ctx)
        ok = torch.equal(f, ref_f) and torch.equal(i, ref_i) and torch.equal(fft.interpolate_poly(f.clone(), ctx=ctx), d)
        ts, last = [], None
        w = d.clone()
        for _ in range(5):
            fft.evaluate_poly(w, ctx=ctx)
        for _ in range(reps):
            ctx.prof_enable(True)
            fft.evaluate_poly(w, ctx=ctx)
            pr = ctx.prof_collect()
            ts.append(sum(v[1] for v in pr.values()))
            last = pr
        ctx.prof_enable(False)
        row.append("%s %s %.1f us (%s)" % (name, "ok" if ok else "MISMATCH", 1e3 * float(np.median(ts)),
                                           " ".join("%s=%.1f" % (k, 1e3 * v[1] / v[0]) for k, v in sorted(last.items()))))
    print("2^%d: %s" % (L, " | ".join(row)), flush=True)
