"""Per-kernel breakdown (wf_prof events) of the examples::rescue-shaped pipeline: f128, 4 columns x 2^20 rows, blowup 8,
Blake3_256, quadratic extension — trace commitment, constraint evaluation, composition poly + commitment, OOD, DEEP, FRI."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd import air as wair, crypto, fri, prover
from winterfell_amd.math import fields

ctx = winterfell_amd.default_context(0)
f128 = fields.f128
rng = np.random.default_rng(1)
tn, tb, D = 1 << 20, 8, 2
ew = D * f128.W
cm = prover.ColMatrix(ctx.to_device(rng.integers(0, 1 << 62, (4, tn * 2), dtype=np.uint64)), field=f128)
dom = prover.StarkDomain(tn, tb, field=f128)
rair = wair.RescueAir(tn, [1, 2], [3, 4], tb)
cc = prover.ConstraintCompositionCoefficients(rng.integers(1, 1 << 62, (4, ew), dtype=np.uint64), rng.integers(1, 1 << 62, (4, ew), dtype=np.uint64))
z = rng.integers(1, 1 << 62, ew, dtype=np.uint64)
cct, ccq = rng.integers(1, 1 << 62, (4, ew), dtype=np.uint64), rng.integers(1, 1 << 62, (3, ew), dtype=np.uint64)


class Chan:
    def __init__(self):
        self.k = 0

    def commit_fri_layer(self, root):
        self.k += 1

    def draw_fri_alpha(self):
        return f128.pack([f128.new(1234 + self.k), f128.new(99 + self.k)])


def run():
    lde, polys = prover.DefaultTraceLde.new(crypto.Blake3_256, cm, dom)
    ev = prover.DefaultConstraintEvaluator(rair, cc, D).evaluate(lde, dom)
    com, cpoly = prover.build_constraint_commitment(crypto.Blake3_256, ev, 3, dom, ext_degree=D, field=f128, ctx=ctx)
    table = prover.TracePolyTable(polys)
    table.get_ood_frame(z, D)
    prover.composition_poly_ood_frame(cpoly, z, D)
    deep = prover.DeepCompositionPoly(z, cct, ccq, D)
    deep.add_trace_polys(table, cpoly)
    dev = deep.evaluate(dom)
    p = fri.FriProver(fri.FriOptions(tb, 4, 31, field=f128), crypto.Blake3_256, ext_degree=D)
    p.build_layers(Chan(), dev)


run()
torch.cuda.synchronize()
t = time.perf_counter()
run()
torch.cuda.synchronize()
print("total %.2f ms" % ((time.perf_counter() - t) * 1e3))
ctx.prof_enable(True)
run()
agg = ctx.prof_collect()
for name, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %4d launches %8.3f ms" % (name, cnt, ms))
print("sum of kernels %.2f ms" % sum(v[1] for v in agg.values()))
