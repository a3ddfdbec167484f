"""End-to-end composition of the device pieces in the order of Prover::generate_proof (prover/src/lib.rs:275-470) for the two
example AIRs, driven by a host Fiat-Shamir coin, followed by an independent re-check of what a verifier would check from
the queried data (verifier/src/lib.rs:139-330): Merkle openings, the out-of-domain constraint equation, DEEP composition at
every query position, and the FRI fold chain down to the remainder.  The pipeline itself is the product's
prover.prove() (coin and channel in winterfell_amd/crypto/random.py, winterfell_amd/prover/channel.py); the exact transcript
serialisation of the reference's ProverChannel (context / proof bytes) is out of scope, so this is a consistency test of
the pipeline, not a byte-level proof comparison."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


from verifier_util import Ext as ExtOps, ood_constraint_equation_holds  # noqa: E402


@pytest.mark.parametrize("example,fname,hname,n,D", [("fib_small", "f64", "Blake3_256", 1 << 10, 2), ("fib_small", "f64", "Rp64_256", 1 << 8, 1),
                                                      ("rescue", "f128", "Blake3_256", 1 << 9, 2), ("rescue", "f128", "Sha3_256", 1 << 8, 1),
                                                      ("mulfib8", "f128", "Blake3_256", 1 << 8, 2), ("vdf_exempt", "f128", "Blake3_192", 1 << 9, 1)])
def test_prove_then_check(oracle, example, fname, hname, n, D):
    import winterfell_amd
    from winterfell_amd import air as wair, crypto, fri, prover
    from winterfell_amd.math import fields
    ctx = winterfell_amd.default_context()
    hasher = getattr(crypto, hname)
    fld, ofld = {"f64": (fields.f64, oracle.f64t), "f128": (fields.f128, oracle.f128)}[fname]
    blowup, folding, rem_deg, num_queries, grinding = 8, 4, 7, 12, 8
    W, ew = fld.W, D * fld.W
    one = fld.new(1)
    E = ExtOps(ofld, D, one)
    # ---- 1. trace, AIR, public inputs -> coin seed (prover/src/lib.rs:284-297)
    if example == "fib_small":
        trace = ofld.fib_small_build_trace(n)
        result = fld.unpack(trace[1])[n - 1]
        air = wair.FibSmall(n, result, blowup, fld)
        pub = [result]
        air_id = 0
    elif example == "rescue":
        trace = ofld.rescue_build_trace([42, 43], n // 16)
        t0, t1 = fld.unpack(trace[0]), fld.unpack(trace[1])
        air = wair.RescueAir(n, [t0[0], t1[0]], [t0[n - 1], t1[n - 1]], blowup)
        pub = [t0[0], t1[0], t0[n - 1], t1[n - 1]]
        air_id = 1
    elif example == "mulfib8":
        trace = ofld.mulfib8_build_trace(n)
        result = fld.unpack(trace[6])[n - 1]
        air = wair.MulFib8(n, result, blowup, fld)
        pub = [result]
        air_id = 4
    else:                                          # vdf with two exempt steps: the last row of the trace is garbage
        trace = ofld.vdf_build_trace(31337, n, exempt=True)
        result = fld.unpack(trace[0])[n - 2]
        air = wair.Vdf(n, 31337, result, blowup, exempt=True, field=fld)
        pub = [31337, result]
        air_id = 6
    domain = prover.StarkDomain(n, blowup, field=fld)
    N = n * blowup
    # ---- 2..6: the product's prove() (winterfell_amd/prover/prove.py), everything data-parallel on the device
    options = prover.ProofOptions(num_queries, blowup, grinding, ext_degree=D, fri_folding_factor=folding, fri_remainder_max_degree=rem_deg)
    proof = prover.prove(air, prover.ColMatrix(trace, 1, ctx, fld), options, hasher, pub)
    nt, ncols, width = air.num_transition_constraints(), air.num_constraint_composition_columns(), air.TRACE_WIDTH
    cc, z, positions, nonce = proof.constraint_coefficients, proof.ood_point, proof.query_positions, proof.pow_nonce
    (ood_cur, ood_next), (q_cur, q_next) = proof.ood_trace_frame, proof.ood_constraint_frame
    cc_t, cc_c = proof.deep_coefficients
    fopts = proof.fri_options
    trace_root, constraint_root = proof.trace_commitment, proof.constraint_commitment
    assert 0 < len(positions) <= num_queries and len(proof.commitments) == 2 + len(proof.fri_layers) + 1
    (t_rows, (t_leaves, t_proof)), = proof.trace_queries
    c_rows, (c_leaves, c_proof) = proof.constraint_queries

    class _F:                                       # the names the checks below were written against
        layers, remainder_poly = proof.fri_layers, proof.fri_remainder

        @staticmethod
        def num_layers():
            return len(proof.fri_layers)

    class _C:
        commitments, alphas = proof.commitments[2:], proof.fri_alphas
    fprover, fchan = _F, _C

    # ================= the checks a verifier would make =================
    # (0) Fiat-Shamir replay: a fresh channel fed the proof's commitments and frames re-derives every challenge
    #     (verifier/src/channel.rs + lib.rs:139-260), and the nonce satisfies the grinding condition
    replay = prover.ProverChannel(air, options, hasher, pub)
    replay.commit_trace(trace_root)
    rcc = replay.get_constraint_composition_coeffs()
    assert np.array_equal(rcc.transition, cc.transition) and np.array_equal(rcc.boundary, cc.boundary)
    replay.commit_constraints(constraint_root)
    assert np.array_equal(replay.get_ood_point(), z)
    replay.send_ood_evaluations(proof.ood_trace_frame, proof.ood_constraint_frame)
    rt, rc_ = replay.get_deep_composition_coeffs()
    assert np.array_equal(rt, cc_t) and np.array_equal(rc_, cc_c)
    for li in range(len(proof.fri_layers)):
        replay.commit_fri_layer(proof.commitments[2 + li])
        assert np.array_equal(replay.draw_fri_alpha(), proof.fri_alphas[li])
    replay.commit_fri_layer(proof.commitments[-1])
    assert np.array_equal(replay.public_coin.seed, proof.pow_seed)
    assert crypto.check_leading_zeros(hasher, proof.pow_seed, nonce) >= grinding
    assert nonce == 1 or all(crypto.check_leading_zeros(hasher, proof.pow_seed, v) < grinding for v in range(max(1, nonce - 8), nonce))
    replay.pow_nonce = nonce
    assert replay.get_query_positions() == positions
    # (a) Merkle openings of the queried rows against the two commitments
    assert crypto.MerkleTree.verify_batch(hasher, trace_root, positions, t_leaves, t_proof) is None
    assert crypto.MerkleTree.verify_batch(hasher, constraint_root, positions, c_leaves, c_proof) is None
    lv = hasher.hash_elements(np.ascontiguousarray(t_rows), field=fld)
    assert all(np.array_equal(lv[k], t_leaves[k]) for k in range(len(positions)))
    # (b) the OOD constraint equation (verifier/src/evaluator.rs:16-89 vs sum_i z^(i n) H_i(z))
    zi = fld.unpack(z)
    g = fld.new(fld.get_root_of_unity(n.bit_length() - 1))
    zn = E.pow(zi, n)
    H, zp = [0] * D, E.lift(one)
    for i in range(ncols):
        H = E.add(H, E.mul(zp, fld.unpack(q_cur[i])))
        zp = E.mul(zp, zn)
    per = np.zeros(0, dtype=np.uint64)
    if air_id == 1:
        per = ofld.evaluate_columns_at(ofld.air_periodic_polys(1), 9, fld.pack(E.pow(zi, n // 16)), D, 1).reshape(-1)
    tev = fld.unpack(ofld.air_evaluate_transition(air_id, D, ood_cur.reshape(-1), ood_next.reshape(-1), per))
    assert ood_constraint_equation_holds(E, one, g, n, zi, H, [tev[k * D:(k + 1) * D] for k in range(nt)],
                                         [fld.unpack(c) for c in cc.transition], [fld.unpack(r) for r in ood_cur],
                                         [(a.column, a.first_step, a.value) for a in proof.assertions],
                                         [fld.unpack(c) for c in cc.boundary], num_exemptions=air.num_transition_exemptions())
    # (c) DEEP composition at every query position from the opened rows (verifier/src/composer.rs)
    g_lde = fld.new(fld.get_root_of_unity(N.bit_length() - 1))
    zg = E.mul(zi, E.lift(g))
    layer0 = ctx.to_host(fprover.layers[0].evaluations)                          # [rc][folding * ew]
    rc0 = N // folding
    for k, p in enumerate(positions):
        x = E.lift(ofld.mul(int(domain.offset), ofld.exp(g_lde, p)))
        dz, dzg = E.sub(x, zi), E.sub(x, zg)
        acc = [0] * D
        trow = fld.unpack(t_rows[k])
        for i in range(width):
            tx = E.lift(trow[i])
            term = E.add(E.mul(E.sub(tx, fld.unpack(ood_cur[i])), dzg), E.mul(E.sub(tx, fld.unpack(ood_next[i])), dz))
            acc = E.add(acc, E.mul(fld.unpack(cc_t[i]), term))
        crow = fld.unpack(c_rows[k])
        for i in range(ncols):
            hx = crow[i * D:(i + 1) * D]
            term = E.add(E.mul(E.sub(hx, fld.unpack(q_cur[i])), dzg), E.mul(E.sub(hx, fld.unpack(q_next[i])), dz))
            acc = E.add(acc, E.mul(fld.unpack(cc_c[i]), term))
        got = fld.unpack(layer0[p % rc0][(p // rc0) * ew:(p // rc0 + 1) * ew])
        assert E.mul(E.mul(got, dz), dzg) == acc, p
    # (d) FRI: layer commitments open, each queried row folds into the next layer, the last one into the remainder
    pos = positions
    length = N
    for li, layer in enumerate(fprover.layers):
        rc = length // folding
        # what FriProver::build_proof put into the proof for this layer: the queried rows, in fold_positions order, and
        # the batch opening; the verifier hashes the rows into the leaves itself (fri/src/proof.rs:284-330)
        rows_idx = fri.fold_positions(pos, length, folding)
        assert sorted(rows_idx) == sorted(set(p % rc for p in pos))
        player = proof.fri_proof.layers[li]
        q_leaves = hasher.hash_elements(np.ascontiguousarray(player.values), field=fld)
        assert crypto.MerkleTree.verify_batch(hasher, fchan.commitments[li], rows_idx, q_leaves, player.proof) is None
        rows = ctx.to_host(layer.evaluations)
        assert all(np.array_equal(player.values[k], rows[r]) for k, r in enumerate(rows_idx))
        nxt_rows = ctx.to_host(fprover.layers[li + 1].evaluations) if li + 1 < len(fprover.layers) else None
        for r in rows_idx:
            if fname == "f64":
                folded = oracle.apply_drp_rows(rows[r], folding, length, r, int(fopts.domain_offset()), fchan.alphas[li], D)
            else:
                # generic-field oracle folds whole layers only: fold the layer once and pick the row
                folded = None
            if folded is not None:
                if nxt_rows is not None:
                    rc2 = rc // folding
                    assert np.array_equal(folded, nxt_rows[r % rc2][(r // rc2) * ew:(r // rc2 + 1) * ew])
                else:
                    # remainder polynomial (reversed coefficients, prover/mod.rs:230-239) at offset * g_rc^r
                    gl_ = fld.new(fld.get_root_of_unity(rc.bit_length() - 1))
                    x = E.lift(ofld.mul(int(fopts.domain_offset()), ofld.exp(gl_, r)))
                    acc = [0] * D
                    for coef in fprover.remainder_poly.reshape(-1, ew):       # highest degree first
                        acc = E.add(E.mul(acc, x), fld.unpack(coef))
                    assert acc == fld.unpack(folded)
        if fname != "f64":
            full = ofld.apply_drp(rows.reshape(-1), folding, int(fopts.domain_offset()), fchan.alphas[li], D).reshape(rc, ew)
            if nxt_rows is not None:
                rc2 = rc // folding
                for r in rows_idx:
                    assert np.array_equal(full[r], nxt_rows[r % rc2][(r // rc2) * ew:(r // rc2 + 1) * ew])
            else:
                gl_ = fld.new(fld.get_root_of_unity(rc.bit_length() - 1))
                for r in rows_idx:
                    x = E.lift(ofld.mul(int(fopts.domain_offset()), ofld.exp(gl_, r)))
                    acc = [0] * D
                    for coef in fprover.remainder_poly.reshape(-1, ew):
                        acc = E.add(E.mul(acc, x), fld.unpack(coef))
                    assert acc == fld.unpack(full[r])
        pos, length = rows_idx, rc
    assert len(fchan.commitments) == fprover.num_layers() + 1
    assert proof.fri_proof.num_layers() == fprover.num_layers() and proof.fri_proof.num_partitions() == 1
    assert np.array_equal(proof.fri_proof.remainder, fprover.remainder_poly)
