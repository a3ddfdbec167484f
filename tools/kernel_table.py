"""Per-kernel roofline table for DESIGN.md: representative launches of every kernel on the path, HIP-event time per launch
(wf_prof), algorithmic bytes per launch and the resulting fraction of the 8 TB/s HBM roofline.
   python tools/kernel_table.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd import air as wair, crypto, fri, prover
from winterfell_amd.math import fft, fields

ctx = winterfell_amd.default_context(0)
rng = np.random.default_rng(9)
rows = []


def measure(label, fn, alg_bytes, note="", reps=5):
    """alg_bytes: {kernel name: algorithmic bytes per launch}"""
    fn()
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    for _ in range(reps):
        fn()
    agg = ctx.prof_collect()
    ctx.prof_enable(False)
    for k, b in alg_bytes.items():
        if k not in agg:
            continue
        cnt, ms = agg[k]
        per = ms / reps
        rows.append((label, k, cnt // reps, per, b, b / (per * 1e-3) / 1e9 if per else 0, note))      # GB/s


f64, f128 = fields.f64, fields.f128
# ---- NTT 2^24 f64
d = ctx.to_device(rng.integers(0, f64.M, 1 << 24, dtype=np.uint64))
measure("2^24-point f64 NTT", lambda: fft.evaluate_poly(d), {"ntt_pass": 2 * 2 * (1 << 24) * 8, "ntt_pass_last": 2 * (1 << 24) * 8}, "VALU-issue bound")
del d
# ---- LDE + commit, narrow and wide
for c, log_n in ((4, 20), (64, 20)):
    n = 1 << log_n
    tr = torch.from_numpy(rng.integers(0, 1 << 62, (c, n), dtype=np.int64)).to(ctx.device)
    dom = prover.StarkDomain(n, 8)
    N = n * 8
    rw = 8 * ((c + 7) // 8)
    b = {"lde_transpose_hash": n * 8 * c * 8 + N * rw * 8 + N * 32, "merkle_stage_blake3": 64 * N, "hash_rows_blake3": N * c * 8 + N * 32,
         "ntt_pass_last": n * c * 8 * 8 + N * rw * 8}
    measure("LDE+commit 2^%d x %d f64 Blake3" % (log_n, c), lambda: prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(tr.clone(), 1, ctx), dom), b,
            "hash: compression rate" if c == 4 else "")
    del tr
# ---- Rescue hashing
data = ctx.to_device(rng.integers(0, f64.M, (1 << 22, 8), dtype=np.uint64))
m = prover.RowMatrix(data, 8, 8, 1, ctx, f64)
measure("Rp64_256 row hash + tree, 2^22 rows x 8", lambda: m.commit_to_rows(crypto.Rp64_256), {"hash_rows_rp64": (1 << 22) * 96, "merkle_stage_rp64": 64 << 22},
        "VALU: 6384 modmuls per permutation", reps=3)
del data, m
# ---- FRI 2^24 quadratic extension, folding 4
ev = ctx.to_device(rng.integers(0, f64.M, (1 << 24) * 2, dtype=np.uint64))


class Chan:
    def __init__(self):
        self.k = 0

    def commit_fri_layer(self, root):
        self.k += 1

    def draw_fri_alpha(self):
        return np.array([f64.new(7 + self.k), f64.new(9)], dtype=np.uint64)


layer_bytes = sum(((1 << 24) >> (2 * k)) * 16 for k in range(9))
measure("FRI commit phase 2^24 quad ext, fold 4", lambda: fri.FriProver(fri.FriOptions(8, 4, 31), crypto.Blake3_256, ext_degree=2, ctx=ctx).build_layers(Chan(), ev),
        {"fri_transpose_hash": 2 * layer_bytes + layer_bytes // 2, "fri_fold": layer_bytes + layer_bytes // 4, "merkle_stage_blake3": 64 * layer_bytes // 64},
        "all layers summed")
del ev
# ---- rescue pipeline pieces (f128, 2^20 x 4, D = 2)
tn, D = 1 << 20, 2
cm = prover.ColMatrix(ctx.to_device(rng.integers(0, 1 << 62, (4, tn * 2), dtype=np.uint64)), field=f128)
dom = prover.StarkDomain(tn, 8, field=f128)
lde, polys = prover.DefaultTraceLde.new(crypto.Blake3_256, cm, dom)
rair = wair.RescueAir(tn, [1, 2], [3, 4], 8)
ew = D * 2
cc = prover.ConstraintCompositionCoefficients(rng.integers(1, 1 << 62, (4, ew), dtype=np.uint64), rng.integers(1, 1 << 62, (4, ew), dtype=np.uint64))
ce = tn * 4
measure("Rescue AIR constraints, 2^22 ce steps (f128, D=2)", lambda: prover.DefaultConstraintEvaluator(rair, cc, D).evaluate(lde, dom),
        {"evaluate_constraints": ce * (2 * 4 * 16 + D * 16 + 2 * 16), "divisor_inv": 2 * ce * 16}, "VALU (f128 products)")
evc = prover.DefaultConstraintEvaluator(rair, cc, D).evaluate(lde, dom)
com, cpoly = prover.build_constraint_commitment(crypto.Blake3_256, evc, 3, dom, ext_degree=D, field=f128, ctx=ctx)
z = rng.integers(1, 1 << 62, ew, dtype=np.uint64)
table = prover.TracePolyTable(polys)
measure("OOD frame, 4 f128 columns x 2^20 at z, z*g", lambda: table.get_ood_frame(z, D), {"poly_eval_at": 4 * tn * 16}, "VALU / latency")
cct, ccq = rng.integers(1, 1 << 62, (4, ew), dtype=np.uint64), rng.integers(1, 1 << 62, (3, ew), dtype=np.uint64)


def deep():
    dp = prover.DeepCompositionPoly(z, cct, ccq, D)
    dp.add_trace_polys(table, cpoly)


measure("DEEP composition (4 + 3 columns, 2^20, f128 D=2)", deep, {"deep_acc": (4 * 16 + 3 * 32 + 32) * tn, "syndiv_final": 2 * 2 * 32 * tn}, "")
print("| workload | kernel | launches | ms per launch set | algorithmic MB | TB/s | % of 8 TB/s | note |")
print("|---|---|---|---|---|---|---|---|")
for label, k, cnt, per, b, gbs, note in rows:
    print("| %s | `%s` | %d | %.3f | %.0f | %.2f | %.0f | %s |" % (label, k, cnt, per, b / 1e6, gbs / 1e3, gbs / 80.0, note))
