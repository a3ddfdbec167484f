import sys
sys.path.insert(0,'.')
import numpy as np, torch, time
import winterfell_amd
from winterfell_amd import crypto, prover
from winterfell_amd.math import fields
ctx = winterfell_amd.default_context(0)
f = fields.f62
rng = np.random.default_rng(3)
rows = 1 << 20
data = ctx.to_device(rng.integers(0, f.M, (rows, 8), dtype=np.uint64))
m = prover.RowMatrix(data, 8, 8, 1, ctx, f)
h = crypto.Rp62_248
m.commit_to_rows(h); torch.cuda.synchronize()
ctx.prof_enable(True)
for _ in range(3): m.commit_to_rows(h)
agg = ctx.prof_collect()
print({k: round(v[1]/3, 3) for k, v in agg.items()})
