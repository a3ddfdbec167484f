"""One-off timing of BASELINE configs[3] on one GPU: f128, 64 columns x 2^22 rows, blowup 8, Blake3, PartitionOptions(8, 1)."""
import sys, time
sys.path.insert(0, '.')
import torch, winterfell_amd
from winterfell_amd import crypto, prover
from winterfell_amd.math import fields
ctx = winterfell_amd.default_context()
f = fields.f128
n, cols, b = 1 << 22, 64, 8
trace = torch.randint(0, 1 << 62, (cols, n * 2), dtype=torch.int64, device=ctx.device)
cm, dom, po = prover.ColMatrix(trace, field=f), prover.StarkDomain(n, b, field=f), prover.PartitionOptions(8, 1)
out = prover.build_trace_commitment(crypto.Blake3_256, cm, dom, po); torch.cuda.synchronize(); del out
ctx.prof_enable(True)
t0 = time.perf_counter()
out = prover.build_trace_commitment(crypto.Blake3_256, cm, dom, po); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("config4 f128 64x2^22 b8 LDE+commit: %.1f ms" % (dt * 1e3))
for k, (c, ms) in sorted(ctx.prof_collect().items()):
    print("  %-24s %3d launches %9.2f ms" % (k, c, ms))
alg = n * cols * 16 * (2 + b) + 64 * b * n
print("  algorithmic bytes %.2f GB -> %.2f TB/s" % (alg / 1e9, alg / dt / 1e12))
