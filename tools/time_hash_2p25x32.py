import sys
sys.path.insert(0, ".")
import numpy as np, torch
import winterfell_amd
from winterfell_amd import crypto, prover
from winterfell_amd.math import fields
ctx = winterfell_amd.default_context(0)
rng = np.random.default_rng(3)
rows, cols = 1 << 25, 32
data = torch.randint(0, 1 << 62, (rows, cols), dtype=torch.int64, device=ctx.device)
m = prover.RowMatrix(data, cols, cols, 1, ctx, fields.f64)
m.hash_rows(crypto.Blake3_256); torch.cuda.synchronize()
ctx.prof_enable(True)
for _ in range(5): m.hash_rows(crypto.Blake3_256)
agg = ctx.prof_collect(); ctx.prof_enable(False)
print({k: round(v[1]/5, 3) for k, v in agg.items()})
