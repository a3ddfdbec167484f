import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import rand_field
import oracle, winterfell_amd
from winterfell_amd import crypto, fri
from winterfell_amd.math import fft, fields
ctx = winterfell_amd.default_context()
D, N, blowup, log_len = 2, 4, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = (1 << log_len) // blowup
p = oracle.f64_from_int(rand_field(5, n * D))
ev = fft.evaluate_poly_with_offset(ctx.to_device(p), None, fields.new(7), blowup, ext_degree=D)
opts = fri.FriOptions(blowup, N, 31)
chan = oracle.ProverChannel(0, D)
prover = fri.FriProver(opts, crypto.Blake3_256, ext_degree=D)
prover.build_layers(chan, ev)
cur = ctx.to_host(ev).copy()
ochan = oracle.ProverChannel(0, D)
for k in range(prover.num_layers()):
    tr = oracle.transpose_slice(cur, N, D)
    leaves, nodes = oracle.fri_layer_commit(0, tr, N, D)
    print(k, "evals", np.array_equal(ctx.to_host(prover.layers[k].evaluations).reshape(-1), tr),
          "root", np.array_equal(prover.layers[k].commitment.root(), nodes[1]), flush=True)
    ochan.commit_fri_layer(nodes[1])
    alpha = ochan.draw_fri_alpha()
    cur = oracle.apply_drp(tr, N, fields.new(7), alpha, D)
rem, com = oracle.fri_remainder(0, cur, fields.new(7), blowup, D)
print("remainder", np.array_equal(prover.remainder_poly.reshape(-1), rem), "commitments", all(np.array_equal(a, b) for a, b in zip(chan.commitments, ochan.commitments + [com])))
# ---- coefficient model, layer by layer
P = oracle.M
c = fields.to_ints(p).reshape(n, D).astype(object)
mchan = oracle.ProverChannel(0, D)
length = 1 << log_len
for k in range(prover.num_layers()):
    mchan.commit_fri_layer(prover.layers[k].commitment.root())
    a0, a1 = (int(v) for v in fields.to_ints(mchan.draw_fri_alpha()))
    acc0, acc1 = c[N - 1::N, 0], c[N - 1::N, 1]
    for j in reversed(range(N - 1)):
        t0 = (acc0 * a0 - 2 * acc1 * a1) % P
        t1 = (acc0 * a1 + acc1 * a0 + acc1 * a1) % P
        acc0, acc1 = (t0 + c[j::N, 0]) % P, (t1 + c[j::N, 1]) % P
    s_ = pow(7, N - 1, P)
    scale, cu = np.empty(len(acc0), dtype=object), 1
    for m in range(len(acc0)):
        scale[m] = cu
        cu = cu * s_ % P
    c = np.stack([(acc0 * scale) % P, (acc1 * scale) % P], axis=1)
    length //= N
    # compare with the GPU's next-layer input at 3 points via Horner on the oracle
    if k + 1 < prover.num_layers():
        nxt = ctx.to_host(prover.layers[k + 1].evaluations).reshape(-1, N, D)   # transposed: [i][j] = e[i + j*rc]
        cm = fields.from_ints(c.astype(np.uint64))
        g = oracle.f64_root_of_unity(length.bit_length() - 1)
        ok = True
        for i in (0, 1, 5):
            x = oracle.f64_mul(fields.new(7), oracle.f64_exp(g, i))
            for d in range(D):
                ok &= int(nxt[i, 0, d]) == oracle.poly_eval(np.ascontiguousarray(cm[:, d]), x)
        print("model layer", k, "matches GPU evaluations:", ok, "coeffs", c.shape, flush=True)
want = fields.from_ints(c[::-1].astype(np.uint64))
print("model remainder equal:", np.array_equal(prover.remainder_poly, want))
