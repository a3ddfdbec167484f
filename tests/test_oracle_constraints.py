"""Oracle self-check for the constraint-evaluation restatement (prover/src/constraints/evaluator + evaluation_table).
The reference ships no golden vectors for it, so the restated PROVER path (evaluate over the constraint-evaluation domain,
divide by the divisors' evaluations, combine) is pinned against the VERIFIER's formula (verifier/src/evaluator.rs:16-89,
verifier/src/lib.rs ood consistency check) evaluated at a random out-of-domain point with independent python code:
    H(z) == sum_k cc_k C_k(z) / Z_t(z)  +  sum_groups sum_a cc_a (T_col(z) - value) / (z - g^step)
plus the structural facts the prover asserts: the combined evaluations interpolate to a polynomial of degree
< num_composition_columns * n (valid trace => exact divisibility), and a corrupted trace breaks that."""
import numpy as np
import pytest

from verifier_util import Ext, ood_constraint_equation_holds


def _setup(oracle, fname):
    fld = {"f64": oracle.f64t, "f128": oracle.f128, "f62": oracle.f62}[fname]
    gen = {"f64": 7, "f128": 3, "f62": 3}[fname]
    new = {"f64": oracle.f64_new, "f128": lambda v: v, "f62": oracle.f62_new}[fname]
    return fld, new(gen), new


def _rand_e(fld, count, D, seed):
    rng = np.random.default_rng(seed)
    return [[int(rng.integers(1, 2**62)) * int(rng.integers(1, 2**62)) % fld.M for _ in range(D)] for _ in range(count)]


def _case(oracle, fname, air, n, D, seed, corrupt=False):
    fld, offset, new = _setup(oracle, fname)
    W = fld.W
    if air == fld.AIR_FIB_SMALL:
        trace = fld.fib_small_build_trace(n)
        ce_blowup, ncols = 2, 1
        one = new(1)
        res = fld.unpack(trace[1])[n - 1]
        assertions = [(0, 0, one), (1, 0, one), (1, n - 1, res)]
    elif air == fld.AIR_RESCUE:
        trace = fld.rescue_build_trace([42, 43], n // 16)
        ce_blowup, ncols = 4, 3
        t0, t1 = fld.unpack(trace[0]), fld.unpack(trace[1])
        assertions = [(0, 0, t0[0]), (1, 0, t1[0]), (0, n - 1, t0[n - 1]), (1, n - 1, t1[n - 1])]
    elif air == fld.AIR_FIB8:                        # fib8/air.rs:67-76
        trace = fld.fib8_build_trace(n)
        ce_blowup, ncols = 2, 1
        assertions = [(0, 0, new(13)), (1, 0, new(21)), (1, n - 1, fld.unpack(trace[1])[n - 1])]
    elif air == fld.AIR_MULFIB2:                     # mulfib2/air.rs:62-71: degree-2 constraints
        trace = fld.mulfib2_build_trace(n)
        ce_blowup, ncols = 2, 1
        assertions = [(0, 0, new(1)), (1, 0, new(2)), (0, n - 1, fld.unpack(trace[0])[n - 1])]
    elif air == fld.AIR_MULFIB8:                     # mulfib8/air.rs:84-93
        trace = fld.mulfib8_build_trace(n)
        ce_blowup, ncols = 2, 1
        assertions = [(0, 0, new(1)), (1, 0, new(2)), (6, n - 1, fld.unpack(trace[6])[n - 1])]
    else:                                            # vdf/regular/air.rs:63-66, vdf/exempt/air.rs: degree 3, one column
        ex = air == fld.AIR_VDF_EXEMPT
        trace = fld.vdf_build_trace(1234567, n, exempt=ex)
        ce_blowup, ncols = 2, 2                      # degree 3: ce_blowup = npo2(3 - 1) = 2 (degree.rs min_blowup_factor)
        last = n - 2 if ex else n - 1
        assertions = [(0, 0, 1234567), (0, last, fld.unpack(trace[0])[last])]
    if corrupt:
        trace = trace.copy()
        trace[min(1, trace.shape[0] - 1), 5 * W] ^= np.uint64(1)
    lde_blowup = 8
    polys, lde, _, _ = fld.build_trace_commitment(0, trace, lde_blowup, offset)
    width, nt, npc, cyc = fld.AIR_SHAPES[air]
    cc_t, cc_b = _rand_e(fld, nt, D, seed), _rand_e(fld, len(assertions), D, seed + 1)
    out = fld.evaluate_constraints(air, lde, lde.shape[1] // W, n, lde_blowup, ce_blowup, offset, D,
                                   fld.pack(sum(cc_t, [])), [(c, s, fld.pack([v])) for c, s, v in assertions], fld.pack(sum(cc_b, [])))
    return dict(nex=2 if air == getattr(fld, "AIR_VDF_EXEMPT", -1) else 1, fld=fld, offset=offset, new=new, polys=polys, out=out, n=n, D=D, ce_blowup=ce_blowup, ncols=ncols, cc_t=cc_t, cc_b=cc_b,
                assertions=assertions, air=air, width=width, nt=nt, npc=npc, cyc=cyc)


def _composition_coeffs(c):
    fld, D = c["fld"], c["D"]
    co = fld.interpolate_poly_with_offset(c["out"], c["offset"], D)          # CompositionPoly::new, composition_poly.rs:72-73
    return np.asarray(co).reshape(c["n"] * c["ce_blowup"], D * fld.W)


CASES = [("f64", 0, 16, 1), ("f64", 0, 64, 2), ("f64", 0, 32, 3), ("f62", 0, 16, 2), ("f128", 0, 16, 1),
         ("f128", 1, 32, 1), ("f128", 1, 64, 2),
         ("f128", 2, 32, 1), ("f64", 2, 16, 2), ("f128", 3, 16, 2), ("f64", 3, 32, 3), ("f62", 3, 16, 1), ("f128", 4, 32, 1), ("f64", 4, 16, 2),
         ("f128", 5, 32, 1), ("f128", 5, 16, 2), ("f128", 6, 32, 2), ("f128", 6, 64, 1)]


@pytest.mark.parametrize("fname,air,n,D", CASES)
def test_prover_evaluation_matches_verifier_formula(oracle, fname, air, n, D):
    c = _case(oracle, fname, air, n, D, 17 * n + D)
    fld, new, E = c["fld"], c["new"], Ext(c["fld"], D)
    co = _composition_coeffs(c)
    # degree: num_constraint_composition_columns * n coefficients at most (context.rs:265-285; composition_poly.rs:75-78)
    assert not co[c["ncols"] * n:].any()
    assert co[(c["ncols"] - 1) * n:c["ncols"] * n].any()
    rng = np.random.default_rng(5)
    z = [int(rng.integers(1, 2**62)) % fld.M for _ in range(D)]
    g = fld.root_of_unity(n.bit_length() - 1)
    zg = E.mul(z, E.lift(g))
    zw, zgw = fld.pack(z), fld.pack(zg)
    cur = fld.evaluate_columns_at(c["polys"], c["width"], zw, D, 1)           # ood main frame (verifier reads it from the proof)
    nxt = fld.evaluate_columns_at(c["polys"], c["width"], zgw, D, 1)
    # periodic values at z: poly(z^(n / cycle)), verifier/src/evaluator.rs:27-35
    per = np.zeros(0, dtype=np.uint64)
    if c["npc"]:
        zc = E.lift(new(1))
        for _ in range(n // c["cyc"]):
            zc = E.mul(zc, z)
        per = fld.evaluate_columns_at(fld.air_periodic_polys(c["air"]), c["npc"], fld.pack(zc), D, 1).reshape(-1)
    tev = fld.unpack(fld.air_evaluate_transition(c["air"], D, cur.reshape(-1), nxt.reshape(-1), per))
    E.one = new(1)
    curl = fld.unpack(cur.reshape(-1))
    H = E.horner([fld.unpack(row) for row in co], z)
    assert ood_constraint_equation_holds(E, new(1), g, n, z, H, [tev[k * D:(k + 1) * D] for k in range(c["nt"])], c["cc_t"],
                                         [curl[k * D:(k + 1) * D] for k in range(c["width"])], c["assertions"], c["cc_b"], num_exemptions=c["nex"])


@pytest.mark.parametrize("fname,air,n", [("f64", 0, 32), ("f128", 1, 32), ("f64", 3, 32), ("f128", 4, 16)])
def test_invalid_trace_breaks_divisibility(oracle, fname, air, n):
    """A trace that violates a transition constraint is not divisible by the divisor: the interpolated composition
    polynomial spills over the degree bound (what the prover's debug degree validation catches)."""
    c = _case(oracle, fname, air, n, 1, 3, corrupt=True)
    co = _composition_coeffs(c)
    assert co[c["ncols"] * n:].any()


def test_periodic_value_table_reference_example(oracle):
    """prover/src/constraints/evaluator/periodic_table.rs:100-146: periodic values at ce step i equal the column's
    polynomial at x_i^(n / cycle) — checked here for the Rescue AIR's 9 periodic columns through the evaluator: with all
    coefficients on constraint k only, no assertions weight, the flag column must reproduce CYCLE_MASK on the trace domain."""
    fld = oracle.f128
    polys = fld.air_periodic_polys(fld.AIR_RESCUE)
    g16 = fld.root_of_unity(4)
    for i in range(16):
        x = fld.exp(g16, i)
        vals = fld.unpack(fld.evaluate_columns_at(polys, 9, fld.pack([x]), 1, 1).reshape(-1))
        assert vals[0] == (1 if i < 14 else 0)                                   # CYCLE_MASK, examples/src/rescue/air.rs:18-35
    # x -> x^(n/16) maps the trace domain onto the 16-cycle: hash_flag(g^i) = mask[i % 16]
    n = 64
    g = fld.root_of_unity(6)
    for i in (0, 13, 14, 15, 16, 30, 47, 63):
        x = fld.exp(fld.exp(g, i), n // 16)
        assert fld.unpack(fld.evaluate_columns_at(polys[:1], 1, fld.pack([x]), 1, 1).reshape(-1))[0] == (1 if i % 16 < 14 else 0)


# ---- RescueRapsAir: a trace with an auxiliary segment (examples/src/rescue_raps) ------------------------------------------------
def _raps_case(oracle, n, D, seed, corrupt=None):
    fld, offset = oracle.f128, 3
    chain = n // 16
    seeds = [[1000 + 2 * i, 77 * i + 5] for i in range(chain)]
    permuted = seeds[2:] + seeds[:2]                                          # rescue_raps/mod.rs:83-85
    trace = fld.rescue_raps_build_trace(seeds, permuted)
    rand = _rand_e(fld, 3, D, seed + 2)
    aux = fld.rescue_raps_build_aux(trace, D, fld.pack(sum(rand, [])))
    if corrupt == "main":
        trace = trace.copy()
        trace[1, 5 * fld.W] ^= np.uint64(1)
    if corrupt == "aux":
        aux = aux.copy()
        aux[2, 7 * D * fld.W] ^= np.uint64(1)
    lde_blowup, ce_blowup, ncols = 8, 4, 3
    polys, lde, _, _ = fld.build_trace_commitment(0, trace, lde_blowup, offset)
    apolys, alde, _, _ = fld.build_trace_commitment(0, aux, lde_blowup, offset, D=D)
    t = [fld.unpack(col) for col in trace]
    last = n - 1
    assertions = [(2, 0, 0), (3, 0, 0), (6, 0, 0), (7, 0, 0), (0, last, t[0][last]), (1, last, t[1][last]), (4, last, t[4][last]), (5, last, t[5][last])]
    one_e = [1] + [0] * (D - 1)
    aux_assertions = [(2, 0, one_e), (2, last, one_e)]                        # get_aux_assertions, air.rs:236-239
    cc_t, cc_b, cc_x = _rand_e(fld, 8 + 3, D, seed), _rand_e(fld, 8, D, seed + 1), _rand_e(fld, 2, D, seed + 3)
    out = fld.evaluate_constraints_full(fld.AIR_RESCUE_RAPS, lde, lde.shape[1] // fld.W, alde, alde.shape[1] // fld.W, n, lde_blowup, ce_blowup,
                                        offset, D, fld.pack(sum(cc_t, [])), [(c, s, fld.pack([v])) for c, s, v in assertions], fld.pack(sum(cc_b, [])),
                                        [(c, s, fld.pack(v)) for c, s, v in aux_assertions], fld.pack(sum(cc_x, [])), fld.pack(sum(rand, [])))
    return dict(fld=fld, offset=offset, n=n, D=D, ce_blowup=ce_blowup, ncols=ncols, out=out, polys=polys, apolys=apolys, rand=rand, cc_t=cc_t,
                cc_b=cc_b, cc_x=cc_x, assertions=assertions, aux_assertions=aux_assertions, aux=aux)


@pytest.mark.parametrize("n,D", [(64, 1), (64, 2), (128, 2)])
def test_aux_segment_evaluation_matches_verifier_formula(oracle, n, D):
    """evaluate_fragment_full (main + auxiliary transition constraints under one divisor, main + aux assertions per boundary
    group) against the verifier's out-of-domain equation with the aux frame, for the RAP example."""
    c = _raps_case(oracle, n, D, 31 * n + D)
    fld, E = c["fld"], Ext(c["fld"], D)
    E.one = 1
    assert fld.unpack(c["aux"][2])[-D:] == [1] + [0] * (D - 1)                # the permutation argument closes: last cell = 1
    co = _composition_coeffs(c)
    assert not co[c["ncols"] * n:].any() and co[(c["ncols"] - 1) * n:c["ncols"] * n].any()
    rng = np.random.default_rng(9)
    z = [int(rng.integers(1, 2**62)) for _ in range(D)]
    g = fld.root_of_unity(n.bit_length() - 1)
    zw, zgw = fld.pack(z), fld.pack(E.mul(z, E.lift(g)))
    cur, nxt = fld.evaluate_columns_at(c["polys"], 8, zw, D, 1), fld.evaluate_columns_at(c["polys"], 8, zgw, D, 1)
    acur, anxt = fld.evaluate_columns_at(c["apolys"], 3, zw, D, D), fld.evaluate_columns_at(c["apolys"], 3, zgw, D, D)
    zc = E.pow(z, n // 16)
    per = fld.evaluate_columns_at(fld.air_periodic_polys(fld.AIR_RESCUE_RAPS), 10, fld.pack(zc), D, 1).reshape(-1)
    tev = fld.unpack(fld.air_evaluate_transition(fld.AIR_RESCUE_RAPS, D, cur.reshape(-1), nxt.reshape(-1), per))
    aev = fld.unpack(fld.air_evaluate_aux_transition(fld.AIR_RESCUE_RAPS, D, D, cur.reshape(-1), nxt.reshape(-1), acur.reshape(-1), anxt.reshape(-1),
                                                     per, fld.pack(sum(c["rand"], []))))
    evals = [tev[k * D:(k + 1) * D] for k in range(8)] + [aev[k * D:(k + 1) * D] for k in range(3)]
    row = fld.unpack(cur.reshape(-1)) + fld.unpack(acur.reshape(-1))
    ood_cur = [row[k * D:(k + 1) * D] for k in range(11)]
    # aux assertions address column 8 + c of the joint row; their values are E elements: subtract them up front
    assertions = list(c["assertions"])
    for col, step, val in c["aux_assertions"]:
        ood_cur.append(E.sub(ood_cur[8 + col], val))
        assertions.append((len(ood_cur) - 1, step, 0))
    H = E.horner([fld.unpack(r) for r in co], z)
    assert ood_constraint_equation_holds(E, 1, g, n, z, H, evals, c["cc_t"], ood_cur, assertions, c["cc_b"] + c["cc_x"])


@pytest.mark.parametrize("where", ["main", "aux"])
def test_aux_segment_invalid_trace_breaks_divisibility(oracle, where):
    c = _raps_case(oracle, 64, 2, 4, corrupt=where)
    co = _composition_coeffs(c)
    assert co[c["ncols"] * 64:].any()


def _value_poly(vals, M, wk):
    """coefficients of the polynomial with b(wk^j) = vals[j] (the inverse DFT, written out with python integers)"""
    k = len(vals)
    kinv = pow(k, M - 2, M)
    return [kinv * sum(v * pow(wk, (-j * m) % k, M) for j, v in enumerate(vals)) % M for m in range(k)]


@pytest.mark.parametrize("fname", ["f128", "f64"])
def test_multi_value_assertions_follow_the_reference_definition(oracle, fname):
    """Assertion::periodic and Assertion::sequence (air/src/air/assertions/mod.rs:84-120) in the restated evaluator, against the
    reference's definition written out with python integers: the boundary constraint of an assertion is f(x) - b(x) over the divisor
    x^k - g^(first_step k), k = the number of asserted steps (air/src/air/divisor.rs:64-97); b is the value, or for a sequence the
    polynomial interpolated from the values and evaluated at x g^(-first_step) (air/src/air/boundary/constraint.rs:60-147).  With the
    transition coefficients zeroed, every point of the constraint-evaluation domain must equal the sum of the groups' quotients."""
    fld, offset, new = _setup(oracle, fname)
    W, M, n, lde_blowup, ce_blowup = fld.W, fld.M, 64, 8, 2
    canon = (lambda v: int(oracle.f64_as_int(v))) if fname == "f64" else int
    trace = fld.fib_small_build_trace(n)
    cols = [[canon(v) for v in fld.unpack(trace[c])] for c in range(2)]
    polys, lde, _, _ = fld.build_trace_commitment(0, trace, lde_blowup, offset)
    # (column, first_step, stride, canonical values): single, periodic (not true of this trace: the evaluator does not care), two sequences
    A = [(0, 0, 0, [1]), (1, 3, 16, [123456789]), (1, 1, 8, [cols[1][1 + 8 * j] for j in range(8)]), (0, 0, 4, [cols[0][4 * j] for j in range(16)]),
         (0, 1, 8, [cols[0][1 + 8 * j] for j in range(8)])]
    D = 2
    cc_b = _rand_e(fld, len(A), D, 91)
    zero_t = fld.pack([0] * (2 * D))
    wire = [(c, f, s, fld.pack([new(v) for v in vals])) for c, f, s, vals in A]
    out = fld.evaluate_constraints_multi(fld.AIR_FIB_SMALL, lde, lde.shape[1] // W, n, lde_blowup, ce_blowup, offset, D, zero_t, wire,
                                         fld.pack(sum(cc_b, [])))
    out = [canon(v) for v in fld.unpack(out)]
    g, g_ce, off = canon(fld.root_of_unity(6)), canon(fld.root_of_unity(7)), canon(offset)
    ccb = [[canon(v) for v in e] for e in cc_b]
    ce = n * ce_blowup
    lde_rows = [[canon(v) for v in fld.unpack(lde[i * (lde_blowup // ce_blowup)][:2 * W])] for i in range(ce)]
    bpolys = [None if len(vals) == 1 else _value_poly(vals, M, pow(g, s, M)) for _, _, s, vals in A]
    for i in range(ce):
        x = off * pow(g_ce, i, M) % M
        groups = {}
        for a, (c, first, stride, vals) in enumerate(A):
            k = n // stride if stride else 1
            if bpolys[a] is None:
                b = vals[0]
            else:
                y = x * pow(g, (-first) % n, M) % M
                b = sum(cf * pow(y, m, M) for m, cf in enumerate(bpolys[a])) % M
            ev = (lde_rows[i][c] - b) % M
            num = groups.setdefault((stride, first), [[0] * D, (pow(x, k, M) - pow(g, first * k % n, M)) % M])
            num[0] = [(num[0][d] + ccb[a][d] * ev) % M for d in range(D)]          # coefficient (E) times base value
        want = [sum(nm[d] * pow(z, M - 2, M) for nm, z in groups.values()) % M for d in range(D)]
        assert out[i * D:(i + 1) * D] == want, i
    # the all-single case through the new entry point is the old entry point
    S = [(0, 0, 0, [1]), (1, 0, 0, [1]), (1, n - 1, 0, [cols[1][n - 1]])]
    cc_t, cc_s = _rand_e(fld, 2, D, 5), _rand_e(fld, 3, D, 6)
    a1 = fld.evaluate_constraints_multi(fld.AIR_FIB_SMALL, lde, lde.shape[1] // W, n, lde_blowup, ce_blowup, offset, D, fld.pack(sum(cc_t, [])),
                                        [(c, f, s, fld.pack([new(v) for v in vals])) for c, f, s, vals in S], fld.pack(sum(cc_s, [])))
    a2 = fld.evaluate_constraints(fld.AIR_FIB_SMALL, lde, lde.shape[1] // W, n, lde_blowup, ce_blowup, offset, D, fld.pack(sum(cc_t, [])),
                                  [(c, f, fld.pack([new(vals[0])])) for c, f, s, vals in S], fld.pack(sum(cc_s, [])))
    assert np.array_equal(a1, a2)
    # a VALID set (single + sequences that hold on the trace): the combined evaluations are a polynomial of degree < n (exact divisibility)
    V = [A[0], A[2], A[3], A[4]]
    cc_v = _rand_e(fld, len(V), D, 8)
    ok = fld.evaluate_constraints_multi(fld.AIR_FIB_SMALL, lde, lde.shape[1] // W, n, lde_blowup, ce_blowup, offset, D, fld.pack(sum(cc_t, [])),
                                        [(c, f, s, fld.pack([new(v) for v in vals])) for c, f, s, vals in V], fld.pack(sum(cc_v, [])))
    co = np.asarray(fld.interpolate_poly_with_offset(ok, offset, D)).reshape(ce, D * W)
    assert not co[n:].any() and co[:n].any()
    bad = [V[0], (1, 1, 8, [v + (j == 3) for j, v in enumerate(V[1][3])]), V[2], V[3]]      # one wrong value in a sequence
    nok = fld.evaluate_constraints_multi(fld.AIR_FIB_SMALL, lde, lde.shape[1] // W, n, lde_blowup, ce_blowup, offset, D, fld.pack(sum(cc_t, [])),
                                         [(c, f, s, fld.pack([new(v % M) for v in vals])) for c, f, s, vals in bad], fld.pack(sum(cc_v, [])))
    assert np.asarray(fld.interpolate_poly_with_offset(nok, offset, D)).reshape(ce, D * W)[n:].any()


def test_boundary_constraint_groups_of_the_reference_fixture(oracle):
    """The reference's own boundary-constraint fixture (air/src/air/tests.rs:64-188, `get_boundary_constraints`): eight assertions on a
    two-column trace of length 16 — three single, two sequences of four values, a sequence of two, a sequence of two with an offset, one
    periodic — and what the reference ASSERTS about them: five groups with the divisors x - 1, x - g^9, x^4 - g^8, x^2 - 1, x^2 - g^6, the
    constraints of every group with their value polynomials (`build_sequence_poly`, tests.rs:303-309) and polynomial offsets (k, g^-k),
    and the order in which the composition coefficients are handed out (the assertions sorted by stride, first step, column:
    tests.rs:88-100).  The expected evaluations below are computed from THAT structure, written out as the reference states it — not from
    the restated grouping logic — and must equal the restated evaluator's output at every point of the constraint-evaluation domain
    (transition coefficients zero).  f64, as in the reference's test."""
    fld, offset, new = _setup(oracle, "f64")
    W, M, n, lde_blowup, ce_blowup, D = fld.W, fld.M, 16, 8, 2, 1
    canon = lambda v: int(oracle.f64_as_int(v))
    rng = np.random.default_rng(1664)
    trace = fld.pack([new(int(v)) for v in rng.integers(0, 2**62, 2 * n)]).reshape(2, n * W)       # any trace: the assertions need not hold
    polys, lde, _, _ = fld.build_trace_commitment(0, trace, lde_blowup, offset)
    values = [1, 2, 3, 4]
    # the assertions in the reference's order (tests.rs:72-81): (column, first_step, stride, values); stride 0 = Assertion::single
    A = [(0, 0, 0, [3]), (0, 9, 0, [5]), (1, 9, 0, [9]), (0, 2, 4, values), (1, 2, 4, values), (1, 0, 8, values[:2]), (0, 3, 8, values[:2]), (1, 3, 8, [7])]
    cc_words = [c[0] for c in _rand_e(fld, len(A), D, 77)]           # internal words; the k-th coefficient goes to the k-th assertion of the sorted list
    cc = [canon(c) for c in cc_words]
    wire = [(c, f, s, fld.pack([new(v) for v in vals])) for c, f, s, vals in A]
    out = fld.evaluate_constraints_multi(fld.AIR_FIB_SMALL, lde, lde.shape[1] // W, n, lde_blowup, ce_blowup, offset, D, fld.pack([0, 0]), wire,
                                         fld.pack(cc_words))
    out = [canon(v) for v in fld.unpack(out)]
    g = canon(fld.root_of_unity(4))                                  # trace domain generator (tests.rs:87)
    ginv = pow(g, M - 2, M)

    def seq_poly(vals):                                              # build_sequence_poly: interpolate over the subgroup of len(vals) points
        k = len(vals)
        return _value_poly(vals, M, pow(g, n // k, M))

    no_off = (0, 1)
    # (divisor (k, g^e) meaning x^k - g^e, [(column, value polynomial, poly_offset (k, g^-k), coefficient)]) exactly as asserted in
    # tests.rs:112-188 (cc labels 0, 1, 2, 6, 7, 3, 4, 5 there = positions 0 .. 7 of the sorted assertion list here)
    groups = [
        ((1, pow(g, 0, M)), [(0, [3], no_off, cc[0])]),
        ((1, pow(g, 9, M)), [(0, [5], no_off, cc[1]), (1, [9], no_off, cc[2])]),
        ((4, pow(g, 4 * 2, M)), [(0, seq_poly(values), (2, pow(ginv, 2, M)), cc[3]), (1, seq_poly(values), (2, pow(ginv, 2, M)), cc[4])]),
        ((2, pow(g, 0, M)), [(1, seq_poly(values[:2]), no_off, cc[5])]),
        ((2, pow(g, 2 * 3, M)), [(0, seq_poly(values[:2]), (3, pow(ginv, 3, M)), cc[6]), (1, [7], no_off, cc[7])]),
    ]
    ce = n * ce_blowup
    g_ce, off = canon(fld.root_of_unity(5)), canon(offset)
    for i in range(ce):
        x = off * pow(g_ce, i, M) % M
        row = [canon(v) for v in fld.unpack(lde[i * (lde_blowup // ce_blowup)][:2 * W])]
        want = 0
        for (k, ge), constraints in groups:
            num = 0
            for col, poly, (ok, omul), coeff in constraints:
                y = x * omul % M if ok else x                        # BoundaryConstraint::evaluate_at (boundary/constraint.rs:131-144)
                b = sum(cf * pow(y, m, M) for m, cf in enumerate(poly)) % M
                num = (num + coeff * (row[col] - b)) % M
            want = (want + num * pow((pow(x, k, M) - ge) % M, M - 2, M)) % M
        assert out[i] == want, i


def test_composition_poly_columns_follow_the_reference_segment_fixture(oracle):
    """prover/src/constraints/composition_poly.rs:153-166 (`segment`): sixteen coefficients 0 .. 15 in four columns of four are the four
    CONTIGUOUS runs [0..3], [4..7], [8..11], [12..15].  The restated prover cuts the interpolated composition polynomial the same way
    (oracle/prover.py: `coeffs[:ncols * n * ew].reshape(ncols, n * ew)`); checked here on the reference's fixture through the same
    expression, over f128 as in the reference's test."""
    fld = oracle.f128
    n, ncols, ew = 4, 4, fld.W
    coeffs = fld.pack(list(range(16)))
    cpoly = coeffs[:ncols * n * ew].reshape(ncols, n * ew)
    expected = [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    assert [[int(v) for v in fld.unpack(cpoly[k])] for k in range(ncols)] == expected
