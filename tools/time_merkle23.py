import sys
sys.path.insert(0, ".")
import numpy as np, torch
import winterfell_amd
from winterfell_amd import crypto
ctx = winterfell_amd.default_context(0)
rng = np.random.default_rng(3)
lv = ctx.to_device(rng.integers(0, 256, (1 << 23, 32), dtype=np.uint8))
for _ in range(6): crypto.MerkleTree.new(crypto.Blake3_256, lv)
torch.cuda.synchronize()
