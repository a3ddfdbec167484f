"""GPU parity: constraint evaluation for the built-in example AIRs (wf_evaluate_constraints) against the CPU oracle's
restatement of DefaultConstraintEvaluator + ConstraintEvaluationTable::combine, and the prover pipeline around it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wf():
    import winterfell_amd
    from winterfell_amd import air, crypto, prover
    from winterfell_amd.math import fields
    return winterfell_amd.default_context(), prover, fields, air, crypto


def _rand_words(fld, count, seed):
    rng = np.random.default_rng(seed)
    if fld.W == 1:
        return rng.integers(1, fld.M, size=count, dtype=np.uint64)
    vals = [(int(a) << 64 | int(b)) % fld.M for a, b in zip(rng.integers(0, 2**63, size=count), rng.integers(1, 2**63, size=count, dtype=np.uint64))]
    return fld.pack(vals)


def _setup(oracle, fields, air_mod, fname, air_id, n, blowup):
    fld, ofld = {"f64": (fields.f64, oracle.f64t), "f128": (fields.f128, oracle.f128), "f62": (fields.f62, oracle.f62)}[fname]
    if air_id == 0:
        trace = ofld.fib_small_build_trace(n)
        air = air_mod.FibSmall(n, fld.unpack(trace[1])[n - 1], blowup, fld)
    elif air_id == 1:
        trace = ofld.rescue_build_trace([42, 43], n // 16)
        t0, t1 = fld.unpack(trace[0]), fld.unpack(trace[1])
        air = air_mod.RescueAir(n, [t0[0], t1[0]], [t0[n - 1], t1[n - 1]], blowup)
    elif air_id == 2:
        trace = ofld.fib8_build_trace(n)
        air = air_mod.Fib8(n, fld.unpack(trace[1])[n - 1], blowup, fld)
    elif air_id == 3:
        trace = ofld.mulfib2_build_trace(n)
        air = air_mod.MulFib2(n, fld.unpack(trace[0])[n - 1], blowup, fld)
    elif air_id == 4:
        trace = ofld.mulfib8_build_trace(n)
        air = air_mod.MulFib8(n, fld.unpack(trace[6])[n - 1], blowup, fld)
    else:
        ex = air_id == 6
        trace = ofld.vdf_build_trace(987654321, n, exempt=ex)
        air = air_mod.Vdf(n, 987654321, fld.unpack(trace[0])[n - 2 if ex else n - 1], blowup, exempt=ex, field=fld)
    return fld, ofld, trace, air


def _gpu_eval(ctx, prover, crypto, fld, trace, air, D, blowup, seed):
    domain = prover.StarkDomain(air.trace_length(), blowup, field=fld)
    lde, polys = prover.DefaultTraceLde.new(crypto.Blake3_256, prover.ColMatrix(trace, 1, ctx, fld), domain)
    ew = D * fld.W
    cc = prover.ConstraintCompositionCoefficients(_rand_words(fld, air.num_transition_constraints() * D, seed).reshape(-1, ew),
                                                  _rand_words(fld, air.num_assertions() * D, seed + 1).reshape(-1, ew))
    ev = prover.DefaultConstraintEvaluator(air, cc, D)
    return ev.evaluate(lde, domain), cc, ev, lde, polys, domain


CASES = [("f64", 0, 8, 1, 8), ("f64", 0, 64, 2, 8), ("f64", 0, 4096, 3, 2), ("f62", 0, 256, 2, 4), ("f128", 0, 128, 1, 8),
         ("f128", 1, 32, 1, 8), ("f128", 1, 1024, 2, 4), ("f128", 1, 16, 2, 16),
         # the other example AIRs of the reference (fib8, mulfib2, mulfib8, vdf regular / exempt)
         ("f128", 2, 64, 1, 8), ("f64", 2, 1024, 2, 2), ("f128", 3, 128, 2, 8), ("f64", 3, 32, 3, 4), ("f62", 3, 64, 1, 8),
         ("f128", 4, 256, 1, 8), ("f64", 4, 64, 2, 2), ("f128", 5, 64, 1, 8), ("f128", 5, 512, 2, 4), ("f128", 6, 128, 2, 8),
         ("f128", 6, 32, 1, 4)]


@pytest.mark.parametrize("fname,air_id,n,D,blowup", CASES)
def test_evaluate_constraints_vs_oracle(wf, oracle, fname, air_id, n, D, blowup):
    ctx, prover, fields, air_mod, crypto = wf
    fld, ofld, trace, air = _setup(oracle, fields, air_mod, fname, air_id, n, blowup)
    out, cc, ev, lde, polys, domain = _gpu_eval(ctx, prover, crypto, fld, trace, air, D, blowup, 7 * n + D)
    o_lde = ofld.build_trace_commitment(0, trace, blowup, int(domain.offset))[1]
    want = ofld.evaluate_constraints(air.AIR_ID, o_lde, o_lde.shape[1] // fld.W, n, blowup, air.ce_blowup_factor(), int(domain.offset), D,
                                     cc.transition.reshape(-1), [(a.column, a.first_step, fld.pack([a.value])) for a in ev.assertions],
                                     cc.boundary.reshape(-1))
    assert np.array_equal(ctx.to_host(out), want)


@pytest.mark.parametrize("fname,air_id,n,D,blowup", [("f64", 0, 64, 2, 8), ("f128", 0, 256, 1, 8), ("f62", 0, 64, 3, 4), ("f128", 1, 128, 2, 8),
                                                     ("f64", 4, 1 << 14, 2, 4)])
def test_periodic_and_sequence_assertions_vs_oracle(wf, oracle, fname, air_id, n, D, blowup):
    """Assertion::periodic / Assertion::sequence (air/src/air/assertions/mod.rs:84-120) in the device evaluator
    (wf_evaluate_constraints_assertions): an AIR whose get_assertions mixes single, periodic and sequence assertions over several
    (stride, first_step) groups — strides down to 2 (n / 2 asserted steps) and sequences of up to n / 4 values — against the oracle's
    evaluator, which tests/test_oracle_constraints.py pins to the reference's definition with python integers."""
    ctx, prover, fields, air_mod, crypto = wf
    fld, ofld, trace, air = _setup(oracle, fields, air_mod, fname, air_id, n, blowup)
    cols = [fld.unpack(trace[c]) for c in range(2)]
    A = air_mod.Assertion
    extra = [A.periodic(1, 3, 16, fld.new(123456789)), A.periodic(0, 1, 2, fld.new(5)),
             A.sequence(1, 1, 8, [cols[1][1 + 8 * j] for j in range(n // 8)]), A.sequence(0, 0, 4, [cols[0][4 * j] for j in range(n // 4)]),
             A.sequence(0, 1, 8, [cols[0][1 + 8 * j] for j in range(n // 8)]), A.sequence(1, 0, n, [cols[1][0]])]      # the last one is a single assertion
    base = air.get_assertions()
    air.get_assertions = lambda: base + extra
    air._num_main_assertions = air._num_assertions = len(base) + len(extra)
    out, cc, ev, lde, polys, domain = _gpu_eval(ctx, prover, crypto, fld, trace, air, D, blowup, 3 * n + D)
    assert [a.stride for a in ev.assertions] == sorted(a.stride for a in ev.assertions) and any(a.is_sequence() for a in ev.assertions)
    o_lde = ofld.build_trace_commitment(0, trace, blowup, int(domain.offset))[1]
    want = ofld.evaluate_constraints_multi(air.AIR_ID, o_lde, o_lde.shape[1] // fld.W, n, blowup, air.ce_blowup_factor(), int(domain.offset), D,
                                           cc.transition.reshape(-1), [(a.column, a.first_step, a.stride, fld.pack(a.values)) for a in ev.assertions],
                                           cc.boundary.reshape(-1))
    assert np.array_equal(ctx.to_host(out), want)
    # what the reference's constructors reject is an argument error here as well
    from winterfell_amd._lib import WfError
    for bad in (A(0, 0, fld.new(1), stride=3), A(0, 4, fld.new(1), stride=4), A(0, 0, fld.new(1), stride=4, values=[fld.new(1)] * 3),
                A(0, 0, fld.new(1), stride=2 * n)):
        ev.assertions[-1] = bad
        with pytest.raises((WfError, AssertionError)):
            ev.evaluate(lde, domain)


def test_argument_checks(wf, oracle):
    ctx, prover, fields, air_mod, crypto = wf
    from winterfell_amd._lib import WfError
    fld, ofld, trace, air = _setup(oracle, fields, air_mod, "f64", 0, 16, 8)
    out, cc, ev, lde, polys, domain = _gpu_eval(ctx, prover, crypto, fld, trace, air, 1, 8, 1)
    air._ce_blowup = 4                                      # not the AIR's ce_blowup_factor
    with pytest.raises(WfError):
        ev.evaluate(lde, domain)
    air._ce_blowup = 2
    air.AIR_ID = 1                                          # Rescue constants only exist over f128
    with pytest.raises(WfError):
        ev.evaluate(lde, domain)
    air.AIR_ID = 0
    ev.assertions[0].column = 5                             # column out of range
    with pytest.raises(WfError):
        ev.evaluate(lde, domain)
    with pytest.raises(AssertionError, match="blowup factor too small"):
        air_mod.RescueAir(64, [1, 2], [3, 4], blowup_factor=2)
    with pytest.raises(AssertionError, match="composition coefficient"):
        prover.DefaultConstraintEvaluator(air_mod.FibSmall(16, 1), prover.ConstraintCompositionCoefficients(np.zeros(3, np.uint64), np.zeros(3, np.uint64)))


@pytest.mark.parametrize("fname,air_id,log_n,D", [("f64", 0, 20, 2), ("f128", 1, 16, 2), ("f128", 1, 20, 1), ("f128", 4, 16, 2), ("f128", 6, 14, 1),
                                                  ("f64", 3, 18, 3), ("f128", 2, 16, 1)])
def test_full_size_pipeline_properties(wf, oracle, fname, air_id, log_n, D):
    """BASELINE configs[2] shape (examples::rescue, 2^20 rows, blowup 8) and fib_small at 2^20: trace commitment ->
    constraint evaluation -> composition polynomial -> constraint commitment, everything device resident.  Checked:
    exact divisibility (the composition polynomial fits num_constraint_composition_columns * n coefficients and its top
    column is non-trivial), and the verifier's consistency equation at a random out-of-domain point z
    (verifier/src/evaluator.rs:16-89): constraints evaluated on the OOD frame == sum_i z^(i n) H_i(z)."""
    ctx, prover, fields, air_mod, crypto = wf
    from verifier_util import Ext, ood_constraint_equation_holds
    from winterfell_amd.math import fft
    n, blowup = 1 << log_n, 8
    fld, ofld, trace, air = _setup(oracle, fields, air_mod, fname, air_id, n, blowup)
    out, cc, ev, lde, polys, domain = _gpu_eval(ctx, prover, crypto, fld, trace, air, D, blowup, 99)
    ncols, ce = air.num_constraint_composition_columns(), air.ce_domain_size()
    coeffs = fft.interpolate_poly_with_offset(out.clone(), None, domain.offset, ext_degree=D, ctx=ctx, field=fld)
    co = ctx.to_host(coeffs).reshape(ce, D * fld.W)
    assert not co[ncols * n:].any() and co[(ncols - 1) * n:ncols * n].any()
    commitment, cpoly = prover.build_constraint_commitment(crypto.Blake3_256, out, ncols, domain, ext_degree=D, field=fld, ctx=ctx)
    assert cpoly.num_columns() == ncols and commitment.evaluations.num_rows() == n * blowup
    # ---- verifier-side check at z
    E = Ext(ofld, D, fld.new(1))
    rng = np.random.default_rng(11)
    z = [int(rng.integers(1, 2**62)) % fld.M for _ in range(D)]
    zw = fld.pack(z)
    table = prover.TracePolyTable(polys)
    cur, nxt = table.get_ood_frame(zw, D)
    qcur, _ = prover.composition_poly_ood_frame(cpoly, zw, D)
    one = fld.new(1)
    g = fld.new(fld.get_root_of_unity(log_n))
    zn = E.pow(z, n)
    H, zi = [0] * D, E.lift(one)
    for i in range(ncols):                                  # sum_i z^(i n) H_i(z), verifier/src/lib.rs ood check
        H = E.add(H, E.mul(zi, fld.unpack(qcur[i])))
        zi = E.mul(zi, zn)
    per = np.zeros(0, dtype=np.uint64)
    if air_id == 1:
        per = ofld.evaluate_columns_at(ofld.air_periodic_polys(1), 9, fld.pack(E.pow(z, n // 16)), D, 1).reshape(-1)
    tev = fld.unpack(ofld.air_evaluate_transition(air_id, D, cur.reshape(-1), nxt.reshape(-1), per))
    nt = air.num_transition_constraints()
    assert ood_constraint_equation_holds(E, one, g, n, z, H, [tev[k * D:(k + 1) * D] for k in range(nt)],
                                         [fld.unpack(c) for c in cc.transition], [fld.unpack(r) for r in cur],
                                         [(a.column, a.first_step, a.value) for a in ev.assertions], [fld.unpack(c) for c in cc.boundary],
                                         num_exemptions=air.num_transition_exemptions())
