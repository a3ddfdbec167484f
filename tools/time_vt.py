"""Vector tiles (ntt_pass<..., VT>) against position-major tiles for the coset LDE of wide f64 traces, kernel by kernel, same box:
python tools/time_vt.py [reps=7]   (WF_LDE_VT is read once per context)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import winterfell_amd  # noqa: E402
from winterfell_amd import crypto, prover  # noqa: E402
from winterfell_amd._lib import Context  # noqa: E402
from winterfell_amd.math import fft  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
base = winterfell_amd.default_context(0)


def make(**env):
    os.environ.update(env)
    try:
        return Context(0)
    finally:
        for k in env:
            del os.environ[k]


x = torch.randint(0, 1 << 62, (1 << 24,), dtype=torch.int64, device=base.device)
for _ in range(300):
    fft.evaluate_poly(x, ctx=base)
torch.cuda.synchronize()
del x
for L, c, b in ((22, 32, 8), (21, 32, 8), (20, 32, 8), (19, 64, 8), (19, 96, 8), (20, 64, 8), (22, 64, 2), (18, 32, 8)):
    n = 1 << L
    trace = torch.randint(0, 1 << 62, (c, n), dtype=torch.int64, device=base.device)
    dom = prover.StarkDomain(n, b)
    ctxs = {"vt": make(WF_LDE_VT="1"), "no-vt": make(WF_LDE_VT="0")}
    roots = {}
    for name, ctx in ctxs.items():
        run = lambda: prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(trace.clone(), 1, ctx), dom)
        out = run()
        roots[name] = out[1].root().tobytes()
        del out
        tot, per = [], {}
        for _ in range(reps):
            ctx.prof_enable(True)
            out = run()
            pr = ctx.prof_collect()
            del out
            tot.append(sum(v[1] for v in pr.values()))
            for k, v in pr.items():
                per.setdefault(k, []).append(v[1])
        ctx.prof_enable(False)
        print("2^%d x %d b%d %-6s kernels %.3f ms   %s" % (L, c, b, name, float(np.median(tot)),
                                                          " ".join("%s=%.3f" % (k, float(np.median(v))) for k, v in sorted(per.items()))), flush=True)
    assert len(set(roots.values())) == 1, "roots differ"
    for ctx in ctxs.values():
        ctx.sync()
        ctx.close()
    del trace
    torch.cuda.empty_cache()
