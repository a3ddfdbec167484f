"""Pass plans for a single 2^24-point f64 transform, kernel by kernel (WF_NTT_PLAN, one context per plan): every radix <= 256 pass costs
~46-55 us whatever its radix, so four radix-64 passes (213-234 us) lose to three radix-256 passes (171 us).  python tools/time_plans_2p24.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import winterfell_amd
from winterfell_amd._lib import Context
from winterfell_amd.math import fft
base = winterfell_amd.default_context(0)
def make(**env):
    os.environ.update(env)
    try: return Context(0)
    finally:
        for k in env: del os.environ[k]
x = torch.randint(0, 1 << 62, (1 << 24,), dtype=torch.int64, device=base.device)
for _ in range(300): fft.evaluate_poly(x, ctx=base)
torch.cuda.synchronize()
for plan in ("24:8,8,8", "24:6,6,6,6", "24:7,6,6,5", "24:6,6,6,6", "24:8,8,8", "24:7,7,5,5"):
    ctx = make(WF_NTT_PLAN=plan, WF_NTT_BT="0")
    for _ in range(5): fft.evaluate_poly(x, ctx=ctx)
    ts=[]; last=None
    for _ in range(20):
        ctx.prof_enable(True); fft.evaluate_poly(x, ctx=ctx); pr = ctx.prof_collect(); ts.append(sum(v[1] for v in pr.values())); last=pr
    ctx.prof_enable(False)
    print(plan, "%.1f us" % (1e3*float(np.median(ts))), {k:(v[0], round(1e3*v[1]/v[0],1)) for k,v in last.items()}, flush=True)
    ctx.sync(); ctx.close()
