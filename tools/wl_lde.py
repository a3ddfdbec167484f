"""One workload for counter runs: trace LDE + commit over f64 (Blake3_256, blowup 8), K calls after a warm-up.
python tools/wl_lde.py [log_n=22] [cols=32] [K=2]      (WF_NTT_BIG=0 / 1 selects the three-pass / two-pass plan)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd import crypto, prover

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 32
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = winterfell_amd.default_context(0)
n = 1 << log_n
trace = torch.from_numpy(np.random.default_rng(7).integers(0, 1 << 62, (cols, n), dtype=np.int64)).to(ctx.device)
cm, dom = prover.ColMatrix(trace, 1, ctx), prover.StarkDomain(n, 8)
for _ in range(K + 1):
    out = prover.build_trace_commitment(crypto.Blake3_256, cm, dom)
    ctx.sync()
    del out
