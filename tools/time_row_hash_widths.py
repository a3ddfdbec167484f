import sys
sys.path.insert(0, ".")
import numpy as np, torch
import winterfell_amd
from winterfell_amd import crypto, prover
from winterfell_amd.math import fields
ctx = winterfell_amd.default_context(0)
rng = np.random.default_rng(3)
for cols in (8, 32, 96):
    rows = 1 << 22
    data = ctx.to_device(rng.integers(0, fields.M, (rows, cols), dtype=np.uint64))
    m = prover.RowMatrix(data, cols, cols, 1, ctx, fields.f64)
    for h in (crypto.Sha3_256, crypto.Blake3_256):
        m.hash_rows(h); torch.cuda.synchronize()
        ctx.prof_enable(True)
        for _ in range(3): m.hash_rows(h)
        agg = ctx.prof_collect(); ctx.prof_enable(False)
        print(cols, h.__name__, {k: round(v[1]/3, 3) for k, v in agg.items()})
    del data, m
