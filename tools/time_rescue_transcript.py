"""fib_small over f64 with Rp64_256 (SURVEY D2 variant 3b's hasher), whole proof, host transcript against device transcript (the
Rescue coin on 16-lane groups, csrc/coin.hip): wall ms, median of 5.   python tools/time_rescue_transcript.py [log_n=16]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import oracle
import winterfell_amd
from winterfell_amd import air as wair, crypto, prover
from winterfell_amd.math import fields

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, blowup = 1 << log_n, 8
ctx = winterfell_amd.default_context(0)
fld = fields.f64
trace = oracle.f64t.fib_small_build_trace(n)
result = fld.unpack(trace[1])[n - 1]
air = wair.FibSmall(n, result, blowup, fld)
options = prover.ProofOptions(28, blowup, 16, ext_degree=2, fri_folding_factor=4, fri_remainder_max_degree=31)
out = {}
for mode in ("host", "device"):
    ts, proof = [], None
    for _ in range(6):
        ctx.sync()
        t = time.perf_counter()
        proof = prover.prove(air, prover.ColMatrix(trace, 1, ctx, fld), options, crypto.Rp64_256, [result], transcript=mode)
        ctx.sync()
        ts.append((time.perf_counter() - t) * 1e3)
    out[mode] = (float(np.median(ts[1:])), proof.to_bytes())
    print("%-6s transcript: %.2f ms  (%s)" % (mode, out[mode][0], ", ".join("%s %.2f" % kv for kv in sorted(proof.timings_ms.items()))[:400]))
assert out["host"][1] == out["device"][1], "the two transcripts produced different proofs"
print("same proof bytes (%d)" % len(out["host"][1]))
