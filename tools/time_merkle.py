import sys
sys.path.insert(0, ".")
import numpy as np, torch
import winterfell_amd
from winterfell_amd import crypto
ctx = winterfell_amd.default_context(0)
rng = np.random.default_rng(3)
for lg in (23, 20, 16):
    lv = ctx.to_device(rng.integers(0, 256, (1 << lg, 32), dtype=np.uint8))
    crypto.MerkleTree.new(crypto.Blake3_256, lv); torch.cuda.synchronize()
    ctx.prof_enable(True)
    for _ in range(5): crypto.MerkleTree.new(crypto.Blake3_256, lv)
    agg = ctx.prof_collect(); ctx.prof_enable(False)
    print(lg, {k: (v[0]//5, round(v[1]/5, 4)) for k, v in agg.items()})
