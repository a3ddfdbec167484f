"""The acceptance oracle (oracle/verifier.py: an independent restatement of winterfell::verify over Proof::to_bytes) on the CPU:
its field / wire-format pieces against the reference's own vectors, acceptance of the proofs the oracle's CPU prover serialises
(two codes written independently from the reference must agree on the transcript, the frame layout, the position folding and
the wire format for a proof to be accepted), and rejection of corrupted proofs with the reference's error for each corruption."""
import numpy as np
import pytest


def _pub(name, fld_name, art, oracle):
    canon = (lambda v: int(oracle.f64_as_int(v))) if fld_name == "f64t" else int
    pub = [canon(v) for v in art["pub_inputs"]]
    if name == "rescue":
        return dict(seed=pub[:2], result=pub[2:])
    if name == "rescue_raps":
        return dict(result=[pub[:2], pub[2:]])
    return pub


def _cpu_proof(oracle, name, fname, hid, n, D, queries=16, blowup=8, grinding=5, folding=4, rem_deg=7):
    from oracle import prover as op
    fld = getattr(oracle, fname)
    opts = op.Options(queries, blowup, grinding, D, folding, rem_deg)
    art = op.prove(name, fld, hid, n, opts)
    return op.proof_to_bytes(art, fld, hid, opts), _pub(name, fname, art, oracle), art


def test_extension_arithmetic_against_the_reference_vectors(oracle, golden):
    """the verifier's own field code (python integers, polynomial reduction) on math/src/field/f64/tests.rs:228-346"""
    from oracle import verifier as ov
    E2, E3 = ov.Ext(ov.F64, 2), ov.Ext(ov.F64, 3)
    for case in golden["reference"]["f64_quad_mul"]:
        assert list(E2.mul(tuple(v % ov.M64 for v in case["a"]), tuple(v % ov.M64 for v in case["b"]))) == [v % ov.M64 for v in case["out"]]
    for case in golden["reference"]["f64_cube_mul"]:
        assert list(E3.mul(tuple(v % ov.M64 for v in case["a"]), tuple(v % ov.M64 for v in case["b"]))) == [v % ov.M64 for v in case["out"]]
    # x^2 = x + 1 over f128 (f128/mod.rs:267-272): (a0 + a1 phi)(b0 + b1 phi) = a0 b0 + a1 b1 + (a0 b1 + a1 b0 + a1 b1) phi
    Q = ov.Ext(ov.F128, 2)
    a, b = (5, 7), (11, 13)
    assert Q.mul(a, b) == ((5 * 11 + 7 * 13) % ov.M128, (5 * 13 + 7 * 11 + 7 * 13) % ov.M128)
    rng = np.random.default_rng(1)
    for E in (E2, E3, Q):
        for _ in range(5):
            x = tuple(int(v) % E.M for v in rng.integers(1, 2**62, E.D))
            assert E.mul(x, E.inv(x)) == E.one
    with pytest.raises(ov.VerifierError):
        ov.Ext(ov.F128, 3)                                                             # f128/mod.rs:288-308: no cubic extension
    # the two-adic roots are what their definition says (f64/mod.rs:258-267, f128/mod.rs:40-43,162)
    for f in (ov.F64, ov.F128):
        assert pow(f.two_adic_root, 1 << f.two_adicity, f.M) == 1 and pow(f.two_adic_root, 1 << (f.two_adicity - 1), f.M) == f.M - 1
        assert (f.M - 1) % (1 << f.two_adicity) == 0 and ((f.M - 1) >> f.two_adicity) % 2 == 1
    assert ov.F64.root_of_unity(3) == 1 << 24 and ov.F64.root_of_unity(6) == 8       # the reference's choice of root: w_64 = 8 (f64/mod.rs:17)


def test_wire_primitives(oracle):
    from oracle import verifier as ov
    # read_usize against write_usize's definition (utils/core/src/serde/byte_writer.rs:77-91), round trip over the length classes
    for v in list(range(0, 70000, 37)) + [2**k + d for k in range(7, 64, 7) for d in (-1, 0, 1)] + [2**64 - 1]:
        nbytes = 1
        while nbytes < 9 and v >> (7 * nbytes):
            nbytes += 1
        enc = b"\x00" + v.to_bytes(8, "little") if nbytes == 9 else (((v << 1) | 1) << (nbytes - 1)).to_bytes(nbytes, "little")
        r = ov.Reader(enc + b"\xAA")
        assert r.usize() == v and r.u8() == 0xAA and not r.has_more()
    with pytest.raises(ov.VerifierError):
        ov.Reader(b"\x02").usize()                                                     # a two-byte encoding cut short
    # Context::to_elements: the reference's own test vector (air/src/proof/context.rs tests: main width 20, aux 9 with 12 random
    # elements, length 4096, 128 constraints, 30 queries, blowup 8, grinding 20, no extension, folding 8, remainder degree 127)
    raw = bytes([20, 9, 12, 12]) + (0).to_bytes(2, "little")
    info = ov.TraceInfo(ov.Reader(raw))
    assert info.to_elements(8) == [(((20 << 8 | 1) << 8 | 9) << 8) | 12, 4096]
    opts = ov.ProofOptions(ov.Reader(bytes([30, 8, 20, 1, 8, 127, 0, 0, 1, 1])))
    assert opts.to_elements() == [int.from_bytes(bytes([8, 127, 8, 1]), "little"), 20, 30]
    # fold_positions / map_positions_to_indexes (fri/src/folding/mod.rs:159-176, fri/src/utils.rs:9-33)
    assert ov.fold_positions([1, 9, 300, 44, 513], 1024, 4) == [1, 9, 44]
    assert ov.map_positions_to_indexes([0, 1, 2, 5], 64, 4, 4) == [0, 4, 8, 5]
    assert ov.map_positions_to_indexes([3, 7], 64, 4, 1) == [3, 7]


@pytest.mark.parametrize("name,fname,hid,n,D", [("fib_small", "f64t", 0, 64, 1), ("fib_small", "f64t", 1, 32, 2), ("fib_small", "f64t", 0, 64, 3),
                                               ("fib_small", "f64t", 0, 1 << 16, 1),       # BASELINE configs[0]: 2^16 rows, the CPU path end to end
                                               ("rescue", "f128", 0, 64, 2), ("rescue", "f128", 0, 128, 1), ("rescue_raps", "f128", 0, 64, 2),
                                               ("rescue_raps", "f128", 0, 128, 1)])
def test_cpu_prover_proofs_are_accepted(oracle, name, fname, hid, n, D):
    from oracle import verifier as ov
    pb, pub, art = _cpu_proof(oracle, name, fname, hid, n, D)
    out = ov.verify(pb, name, pub, ["Blake3_256", "Rp64_256"][hid])
    assert out["query_positions"] == art["query_positions"] and out["trace_length"] == n and out["ext_degree"] == D
    assert out["pow_nonce"] == art["pow_nonce"]
    # AcceptableOptions::OptionSet
    ov.verify(pb, name, pub, ["Blake3_256", "Rp64_256"][hid], acceptable_options=[out["options"].as_tuple()])
    with pytest.raises(ov.VerifierError) as e:
        ov.verify(pb, name, pub, ["Blake3_256", "Rp64_256"][hid], acceptable_options=[(28, 8, 16, D, 4, 31, 1, 1)])
    assert e.value.kind == "UnacceptableProofOptions"


def test_other_folding_factors_and_remainders(oracle):
    from oracle import verifier as ov
    for folding, rem_deg, blowup in ((2, 3, 4), (8, 15, 8), (16, 7, 16)):
        pb, pub, art = _cpu_proof(oracle, "fib_small", "f64t", 0, 256, 2, queries=12, blowup=blowup, grinding=3, folding=folding, rem_deg=rem_deg)
        out = ov.verify(pb, "fib_small", pub, "Blake3_256")
        assert out["num_fri_layers"] == len(art["fri_layers"]) and len(out["remainder"]) == art["fri_remainder"].shape[0]


def _flip(pb, at, mask=0x01):
    b = bytearray(pb)
    b[at] ^= mask
    return bytes(b)


@pytest.mark.parametrize("name,fname,hid,n,D,hname", [("rescue", "f128", 0, 128, 2, "Blake3_256"), ("fib_small", "f64t", 1, 64, 2, "Rp64_256"),
                                                      ("rescue_raps", "f128", 0, 64, 2, "Blake3_256")])
def test_single_byte_corruptions_are_rejected_with_the_reference_error(oracle, name, fname, hid, n, D, hname):
    from oracle import verifier as ov
    pb, pub, art = _cpu_proof(oracle, name, fname, hid, n, D)
    ov.verify(pb, name, pub, hname)
    lay = ov.layout(pb)
    assert lay["pow_nonce"][1] == len(pb)
    nb = 16 if fname == "f128" else 8

    def rejected(mutated, kinds):
        with pytest.raises(ov.VerifierError) as e:
            ov.verify(mutated, name, pub, hname)
        assert e.value.kind in kinds, e.value
        return e.value

    c0 = lay["commitments"][0]
    rejected(_flip(pb, c0 + 3), {"InconsistentOodConstraintEvaluations"})              # trace root: every later draw changes
    nseg = 2 if name == "rescue_raps" else 1
    rejected(_flip(pb, c0 + 32 * nseg + 5), {"InconsistentOodConstraintEvaluations"})  # constraint root: z changes
    rejected(_flip(pb, c0 + 32 * (nseg + 1) + 7), {"QuerySeedProofOfWorkVerificationFailed", "TraceQueryDoesNotMatchCommitment"})   # first FRI root
    rejected(_flip(pb, lay["commitments"][1] - 1), {"QuerySeedProofOfWorkVerificationFailed", "TraceQueryDoesNotMatchCommitment"})  # remainder commitment
    rejected(_flip(pb, lay["ood_trace_states"][0] + 1 + 2), {"InconsistentOodConstraintEvaluations"})      # an out-of-domain trace value
    rejected(_flip(pb, lay["ood_quotient_states"][0] + 1), {"InconsistentOodConstraintEvaluations"})       # H_0(z)
    # the NEXT-row halves of the frames are not in the OOD equation for these AIRs' quotient columns: they reach the coin and DEEP
    rejected(_flip(pb, lay["ood_quotient_states"][1] - 1, 0x01), {"QuerySeedProofOfWorkVerificationFailed", "TraceQueryDoesNotMatchCommitment",
                                                                  "ProofDeserializationError"})
    rejected(_flip(pb, lay["trace_queries_0_values"][0] + nb), {"TraceQueryDoesNotMatchCommitment"})       # an opened trace row
    rejected(_flip(pb, lay["trace_queries_0_paths"][1] - 1), {"TraceQueryDoesNotMatchCommitment"})         # a node of the batch opening
    if nseg == 2:
        rejected(_flip(pb, lay["trace_queries_1_values"][0] + 1), {"TraceQueryDoesNotMatchCommitment"})    # an opened auxiliary row
    rejected(_flip(pb, lay["constraint_queries_values"][0]), {"ConstraintQueryDoesNotMatchCommitment"})    # an opened quotient row
    rejected(_flip(pb, lay["constraint_queries_paths"][1] - 2), {"ConstraintQueryDoesNotMatchCommitment"})
    e = rejected(_flip(pb, lay["fri_layer_0_values"][0]), {"FriVerificationFailed"})                       # a FRI layer-0 evaluation
    assert "LayerCommitmentMismatch" in str(e)
    e = rejected(_flip(pb, lay["fri_layer_0_paths"][1] - 1), {"FriVerificationFailed"})                    # a FRI layer node
    assert "LayerCommitmentMismatch" in str(e)
    e = rejected(_flip(pb, lay["fri_remainder"][0] + 1), {"FriVerificationFailed"})                        # a remainder coefficient
    assert "InvalidRemainderFolding" in str(e)
    rejected(_flip(pb, lay["pow_nonce"][0]), {"QuerySeedProofOfWorkVerificationFailed", "TraceQueryDoesNotMatchCommitment"})   # the nonce
    rejected(_flip(pb, lay["fri_num_partitions"][0]), {"FriVerificationFailed"})       # P = 2: leaves looked up at the partitioned indexes
    rejected(_flip(pb, lay["num_constraints"][0], 0x02), {"InconsistentOodConstraintEvaluations"})         # Context reaches the coin seed only
    rejected(_flip(pb, lay["modulus"][0] + 1), {"InconsistentBaseField"})
    rejected(_flip(pb, lay["num_unique_queries"][0]), {"ProofDeserializationError", "TraceQueryDoesNotMatchCommitment"})
    rejected(pb + b"\x00", {"ProofDeserializationError"})
    rejected(pb[:-1], {"ProofDeserializationError"})
    # a field element that is not canonical (f64/mod.rs:672-682, f128/mod.rs:394-404)
    s = lay["constraint_queries_values"][0]
    rejected(pb[:s] + b"\xff" * nb + pb[s + nb:], {"ProofDeserializationError"})
    # the proof is for these public inputs only
    if name == "fib_small":
        bad = [pub[0] ^ 1]
    elif name == "rescue":
        bad = dict(seed=pub["seed"], result=[pub["result"][0] + 1, pub["result"][1]])
    else:
        bad = dict(result=[[pub["result"][0][0], pub["result"][0][1] + 1], pub["result"][1]])
    with pytest.raises(ov.VerifierError) as ex:
        ov.verify(pb, name, bad, hname)
    assert ex.value.kind == "InconsistentOodConstraintEvaluations"
    with pytest.raises(ov.VerifierError):
        ov.verify(pb, name, pub, "Rp64_256" if hname == "Blake3_256" else "Blake3_256")


def test_a_sweep_of_random_corruptions_never_passes(oracle):
    """every byte of a proof is bound — by the coin, a Merkle root, the OOD equation, the DEEP / FRI consistency checks or the
    parser — except ProofOptions' hash_rate byte while num_partitions = 1 (air/src/options.rs:428-444: then the partition size is
    the row width whatever the rate)"""
    from oracle import verifier as ov
    pb, pub, art = _cpu_proof(oracle, "fib_small", "f64t", 0, 32, 2, queries=6, grinding=2)
    lay = ov.layout(pb)
    hash_rate_byte = lay["options"][1] - 1
    ov.verify(_flip(pb, hash_rate_byte, 0x06), "fib_small", pub, "Blake3_256")
    rng = np.random.default_rng(11)
    for at in sorted(set(int(v) for v in rng.integers(0, len(pb), 300))):
        if at == hash_rate_byte:
            continue
        with pytest.raises(ov.VerifierError):
            ov.verify(_flip(pb, at, 1 << int(rng.integers(0, 8))), "fib_small", pub, "Blake3_256")


def test_a_dishonest_prover_is_caught(oracle):
    """a trace that breaks the transition constraint at one step: the CPU prover still produces a well-formed proof (it commits to
    whatever it is given), and the verifier rejects it — at the FRI degree check, since the quotient is no polynomial"""
    from oracle import prover as op, verifier as ov
    fld = oracle.f64t
    n = 64
    real = op.example

    def broken(name, f, nn):
        ex = real(name, f, nn)
        t = ex["trace"].copy()
        t[0, 17] = oracle.f64_new(12345)
        ex["trace"] = t
        return ex

    op.example = broken
    try:
        opts = op.Options(16, 8, 3, 2, 4, 7)
        try:
            art = op.prove("fib_small", fld, 0, n, opts)
        except AssertionError:
            return                                                                     # the prover itself noticed (composition degree)
        pb = op.proof_to_bytes(art, fld, 0, opts)
    finally:
        op.example = real
    with pytest.raises(ov.VerifierError):
        ov.verify(pb, "fib_small", _pub("fib_small", "f64t", art, oracle), "Blake3_256")
