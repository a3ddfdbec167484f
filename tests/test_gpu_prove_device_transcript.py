"""prove(..., transcript="device") (winterfell_amd/prover/prove.py: prove_device_transcript): the Fiat-Shamir coin on the device from the
first commitment to the query positions — every reseed, every draw, the proof-of-work search and the query draw queued on the stream
(wf_coin_*, wf_evaluate_constraints_dev, wf_polys_evaluate_at_dev, wf_deep_compose_dev, wf_fri_build_layers, wf_coin_grind,
wf_coin_draw_integers) — must produce THE SAME PROOF, byte for byte, as the host-coin prove(), whose bytes are pinned to the CPU
restatement of the whole prover and accepted by the independent verifier (tests/test_gpu_proof_artefacts.py, test_gpu_verifier.py).
Reference: Prover::generate_proof, prover/src/lib.rs:282-492; ProverChannel, prover/src/channel.rs:87-185."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(oracle, example, fname, n, blowup):
    from winterfell_amd import air as wair
    from winterfell_amd.math import fields
    fld, ofld = {"f64": (fields.f64, oracle.f64t), "f128": (fields.f128, oracle.f128)}[fname]
    if example == "fib_small":
        trace = ofld.fib_small_build_trace(n)
        result = fld.unpack(trace[1])[n - 1]
        return fld, trace, wair.FibSmall(n, result, blowup, fld), [result]
    if example == "rescue":
        trace = ofld.rescue_build_trace([42, 43], n // 16)
        t0, t1 = fld.unpack(trace[0]), fld.unpack(trace[1])
        return fld, trace, wair.RescueAir(n, [t0[0], t1[0]], [t0[n - 1], t1[n - 1]], blowup), [t0[0], t1[0], t0[n - 1], t1[n - 1]]
    if example == "mulfib8":
        trace = ofld.mulfib8_build_trace(n)
        result = fld.unpack(trace[6])[n - 1]
        return fld, trace, wair.MulFib8(n, result, blowup, fld), [result]
    trace = ofld.vdf_build_trace(31337, n, exempt=True)
    result = fld.unpack(trace[0])[n - 2]
    return fld, trace, wair.Vdf(n, 31337, result, blowup, exempt=True, field=fld), [31337, result]


@pytest.mark.parametrize("example,fname,hname,n,D,grinding,queries", [
    ("fib_small", "f64", "Blake3_256", 1 << 10, 2, 8, 12), ("fib_small", "f64", "Blake3_256", 1 << 8, 1, 0, 5), ("fib_small", "f64", "Blake3_256", 1 << 9, 3, 12, 20),
    ("rescue", "f128", "Blake3_256", 1 << 9, 2, 8, 12), ("rescue", "f128", "Blake3_256", 1 << 8, 1, 16, 28), ("mulfib8", "f128", "Blake3_256", 1 << 8, 2, 4, 9),
    ("vdf_exempt", "f128", "Blake3_192", 1 << 9, 1, 8, 12), ("fib_small", "f64", "Sha3_256", 1 << 8, 2, 6, 7),
    ("fib_small", "f64", "Rp64_256", 1 << 9, 2, 6, 9), ("fib_small", "f64", "Rp64_256", 1 << 8, 3, 0, 70), ("fib_small", "f64", "RpJive64_256", 1 << 8, 1, 4, 5)])
def test_device_transcript_proof_equals_the_host_transcript_proof(oracle, example, fname, hname, n, D, grinding, queries):
    import winterfell_amd
    from winterfell_amd import crypto, prover
    ctx = winterfell_amd.default_context()
    hasher = getattr(crypto, hname)
    blowup = 8
    fld, trace, air, pub = _setup(oracle, example, fname, n, blowup)
    options = prover.ProofOptions(queries, blowup, grinding, ext_degree=D, fri_folding_factor=4, fri_remainder_max_degree=7)
    host = prover.prove(air, prover.ColMatrix(trace, 1, ctx, fld), options, hasher, pub)
    dev = prover.prove(air, prover.ColMatrix(trace, 1, ctx, fld), options, hasher, pub, transcript="device")
    if not hasher.DEVICE_COIN:
        pytest.skip("no device coin for this hasher: prove() kept the host transcript")
    assert "queue_whole_transcript" in dev.timings_ms, "prove() fell back to the host transcript"
    # the transcript, value for value ...
    assert len(dev.commitments) == len(host.commitments)
    for k, (a, b) in enumerate(zip(dev.commitments, host.commitments)):
        assert np.array_equal(a, b), "commitment %d" % k
    assert np.array_equal(dev.constraint_coefficients.transition.reshape(-1), host.constraint_coefficients.transition.reshape(-1))
    assert np.array_equal(dev.constraint_coefficients.boundary.reshape(-1), host.constraint_coefficients.boundary.reshape(-1))
    assert np.array_equal(np.asarray(dev.ood_point).reshape(-1), np.asarray(host.ood_point).reshape(-1))
    for a, b in zip(dev.ood_trace_frame + dev.ood_constraint_frame, host.ood_trace_frame + host.ood_constraint_frame):
        assert np.array_equal(np.asarray(a).reshape(-1), np.asarray(b).reshape(-1))
    for a, b in zip(dev.deep_coefficients, host.deep_coefficients):
        assert np.array_equal(np.asarray(a).reshape(-1), np.asarray(b).reshape(-1))
    assert len(dev.fri_alphas) == len(host.fri_alphas) and all(np.array_equal(a, b) for a, b in zip(dev.fri_alphas, host.fri_alphas))
    assert np.array_equal(dev.fri_remainder, host.fri_remainder)
    assert np.array_equal(np.asarray(dev.pow_seed).reshape(-1), np.asarray(host.pow_seed).reshape(-1))
    assert dev.pow_nonce == host.pow_nonce, "the device search must return the serial reference's nonce (the smallest)"
    assert dev.query_positions == host.query_positions
    # ... and the serialised proof
    assert dev.to_bytes() == host.to_bytes()


def test_device_transcript_with_a_rescue_coin(oracle):
    """Rp64_256 (SURVEY D2 variant 3b: f64 + Rp64_256): the coin's steps run on 16-lane groups (csrc/coin.hip), so the whole transcript
    stays on the device for the Rescue family too; the proof is the host-transcript proof byte for byte (which
    tests/test_gpu_proof_artefacts.py holds against the in-repo CPU prover and the independent verifier)"""
    import winterfell_amd
    from winterfell_amd import crypto, prover
    ctx = winterfell_amd.default_context()
    for hasher in (crypto.Rp64_256, crypto.RpJive64_256):
        fld, trace, air, pub = _setup(oracle, "fib_small", "f64", 1 << 8, 8)
        options = prover.ProofOptions(6, 8, 4, ext_degree=1, fri_folding_factor=4, fri_remainder_max_degree=7)
        a = prover.prove(air, prover.ColMatrix(trace, 1, ctx, fld), options, hasher, pub, transcript="device")
        b = prover.prove(air, prover.ColMatrix(trace, 1, ctx, fld), options, hasher, pub)
        assert "queue_whole_transcript" in a.timings_ms and "queue_whole_transcript" not in b.timings_ms
        assert a.pow_nonce == b.pow_nonce and a.query_positions == b.query_positions
        assert a.to_bytes() == b.to_bytes()


def test_device_transcript_falls_back_where_it_does_not_apply(oracle):
    """a proof without FRI layers keeps the host transcript"""
    import winterfell_amd
    from winterfell_amd import crypto, prover
    ctx = winterfell_amd.default_context()
    fld, trace, air, pub = _setup(oracle, "fib_small", "f64", 1 << 4, 8)
    options = prover.ProofOptions(6, 8, 0, ext_degree=1, fri_folding_factor=4, fri_remainder_max_degree=127)
    a = prover.prove(air, prover.ColMatrix(trace, 1, ctx, fld), options, crypto.Blake3_256, pub, transcript="device")
    b = prover.prove(air, prover.ColMatrix(trace, 1, ctx, fld), options, crypto.Blake3_256, pub)
    assert "queue_whole_transcript" not in a.timings_ms and a.to_bytes() == b.to_bytes()


def test_coin_grind_and_draw_integers_against_the_host_coin(oracle):
    """wf_coin_grind / wf_coin_draw_integers alone: the nonce is the host search's (the minimum), the positions are
    DefaultRandomCoin::draw_integers' (crypto/src/random/default.rs:209-248), the coin afterwards is the host coin afterwards; a
    search range without a nonce sets the coin's failed flag (WF_ERR_NOT_FOUND on read)."""
    import winterfell_amd
    from winterfell_amd import crypto
    from winterfell_amd._lib import ptr
    from winterfell_amd.math import fields
    ctx = winterfell_amd.default_context()
    f = fields.f64
    for hname, factor, nq, log_dom in (("Blake3_256", 10, 27, 20), ("Blake3_192", 6, 5, 9), ("Sha3_256", 8, 40, 33), ("Blake3_256", 0, 3, 4),
                                        ("Rp64_256", 6, 70, 20), ("RpJive64_256", 4, 5, 9)):
        hasher = getattr(crypto, hname)
        seed_elems = f.pack([f.new(v) for v in (3, 1, 4, 1, 5, 9, 2, 6)])
        host = crypto.DefaultRandomCoin(hasher, f, seed_elems, ctx)
        host.reseed(hasher.hash_elements(f.pack([f.new(7)]), ctx, field=f))
        dev = host.to_device()
        dev.draw(1, 1)                                           # uploads the state and moves the counter: grinding must not care
        host.draw(1)
        d_nonce, d_pos = ctx.empty_u64(1), ctx.empty_u64(nq)
        ctx.call("wf_coin_grind", hasher.HASH_ID, ptr(dev.state), factor, factor + 10, ptr(d_nonce))
        ctx.call("wf_coin_draw_integers", hasher.HASH_ID, ptr(dev.state), ptr(d_nonce), nq, log_dom, ptr(d_pos))
        nonce = int(ctx.to_host(d_nonce)[0])
        want = crypto.grind_query_seed(hasher, host.seed, factor, ctx=ctx)
        assert nonce == want, (hname, nonce, want)
        assert [int(v) for v in ctx.to_host(d_pos)] == host.draw_integers(nq, 1 << log_dom, nonce)
        seed, counter = dev.read()
        assert np.array_equal(seed, host.seed) and counter == nq
    # no nonce with 30 trailing zero bits among the first 2^10: the failed flag
    hasher = crypto.Blake3_256
    dev = crypto.DefaultRandomCoin(hasher, f, f.pack([f.new(1)]), ctx).to_device()
    dev.draw(1, 1)
    d_nonce = ctx.empty_u64(1)
    ctx.call("wf_coin_grind", hasher.HASH_ID, ptr(dev.state), 30, 10, ptr(d_nonce))
    with pytest.raises(RuntimeError):
        dev.read()
