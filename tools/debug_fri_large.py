import sys, ctypes, numpy as np
sys.path.insert(0, '.')
import oracle, winterfell_amd
from winterfell_amd import crypto, fri
from winterfell_amd.math import fft, fields
from winterfell_amd._lib import ptr
ctx = winterfell_amd.default_context()
P = oracle.M
D, N, blowup = 2, 4, 8
for log_len in (18, 20, 22, 24):
    n = (1 << log_len) // blowup
    rng = np.random.default_rng(log_len)
    p = oracle.f64_from_int(rng.integers(0, P, n * D, dtype=np.uint64))
    ev = fft.evaluate_poly_with_offset(ctx.to_device(p), None, fields.new(7), blowup, ext_degree=D)
    o_ev = oracle.evaluate_poly_with_offset(p, fields.new(7), blowup, D=D, par=True)
    h_ev = ctx.to_host(ev)
    print(log_len, "lde equal:", np.array_equal(h_ev, o_ev), flush=True)
    rows = (1 << log_len) // N
    tr, leaves, nodes = ctx.empty_u64(rows, N * D), ctx.empty_u8(rows, 32), ctx.empty_u8(rows, 32)
    root = np.empty(32, dtype=np.uint8)
    ctx.call("wf_fri_layer_commit", 0, 0, D, ptr(ev), log_len, N, ptr(tr), ptr(leaves), ptr(nodes), root.ctypes.data_as(ctypes.c_void_p))
    o_tr = oracle.transpose_slice(o_ev, N, D)
    print("  transposed equal:", np.array_equal(ctx.to_host(tr).reshape(-1), o_tr), flush=True)
    o_leaves = oracle.hash_rows(0, o_tr.reshape(rows, N * D), N * D)
    print("  leaves equal:", np.array_equal(ctx.to_host(leaves), o_leaves), flush=True)
    o_nodes = oracle.merkle_build(0, o_leaves, par=True)
    print("  nodes equal:", np.array_equal(ctx.to_host(nodes), o_nodes), flush=True)
    alpha = oracle.f64_from_int(rng.integers(0, P, D, dtype=np.uint64))
    off = ctypes.c_uint64(fields.new(7))
    folded = ctx.empty_u64(rows * D)
    ctx.call("wf_fri_apply_drp", 0, D, ptr(tr), log_len, N, ctypes.cast(ctypes.byref(off), ctypes.c_void_p), alpha.ctypes.data_as(ctypes.c_void_p), ptr(folded))
    o_f = oracle.apply_drp(o_tr, N, fields.new(7), alpha, D)
    hf = ctx.to_host(folded)
    eq = np.array_equal(hf, o_f)
    print("  folded equal:", eq, flush=True)
    if not eq:
        bad = np.nonzero(hf != o_f)[0]
        print("   first bad idx", bad[:10], "count", len(bad), "of", len(hf))
