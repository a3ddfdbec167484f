"""Per-hasher timing of the commitment kernels (wf_prof events): hash_rows over a 2^LOG x COLS f64 row-major matrix and
the Merkle build over its leaves.  A/B a kernel variant with WF_HIP_LIBRARY=winterfell_amd/variants/<name>/libwinterfell_hip.so.
    python tools/time_hashers.py [log_rows=22] [cols=8] [hashers=Rp64_256,RpJive64_256,Blake3_256,Sha3_256]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd import crypto, prover
from winterfell_amd.math import fields

log_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 22
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 8
names = (sys.argv[3] if len(sys.argv) > 3 else "Rp64_256,RpJive64_256,Blake3_256,Sha3_256").split(",")
ctx = winterfell_amd.default_context(0)
rng = np.random.default_rng(3)
f = fields.f64
rows = 1 << log_rows
data = ctx.to_device(rng.integers(0, f.M, (rows, cols), dtype=np.uint64))
m = prover.RowMatrix(data, cols, cols, 1, ctx, f)
for name in names:
    h = getattr(crypto, name)
    m.commit_to_rows(h)
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    for _ in range(3):
        m.commit_to_rows(h)
    agg = ctx.prof_collect()
    ctx.prof_enable(False)
    print(name, "rows 2^%d x %d:" % (log_rows, cols), "  ".join("%s %d launches, %.3f ms per commit" % (k.split("<")[0], v[0] // 3, v[1] / 3) for k, v in sorted(agg.items())))
    # small trees: latency of the upper levels (what FRI layers and partition sub-trees look like)
    import time
    for lg in (8, 12, 15, 18):
        if lg >= log_rows:
            continue
        lv = ctx.empty_u8(1 << lg, 32)
        lv.copy_(m.hash_rows(h)[: 1 << lg])
        crypto.MerkleTree.new(h, lv, ctx)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            crypto.MerkleTree.new(h, lv, ctx)
        torch.cuda.synchronize()
        print("   merkle 2^%d leaves: %.3f ms" % (lg, (time.perf_counter() - t) / 5 * 1e3))
