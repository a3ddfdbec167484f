"""wf_evaluate_constraints_aux: constraint evaluation for a trace with an auxiliary segment (RescueRapsAir) on the device
against the oracle's evaluate_fragment_full restatement, value for value, plus the argument checks."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(oracle, n, D, seed, blowup=8):
    import winterfell_amd
    from winterfell_amd import air as wair, crypto, prover
    from winterfell_amd.math import fields
    ctx, fld, f = winterfell_amd.default_context(), oracle.f128, fields.f128
    chain = n // 16
    seeds = [[31 * i + 7, 5 * i + 1] for i in range(chain)]
    permuted = seeds[3:] + seeds[:3]
    trace = fld.rescue_raps_build_trace(seeds, permuted)
    rng = np.random.default_rng(seed)
    rand_e = lambda k: fld.pack([int(rng.integers(1, 2**62)) * int(rng.integers(1, 2**62)) % fld.M for _ in range(k * D)])
    rand = rand_e(3)
    aux = fld.rescue_raps_build_aux(trace, D, rand)
    t = [fld.unpack(c) for c in trace]
    air = wair.RescueRapsAir(n, [[t[0][-1], t[1][-1]], [t[4][-1], t[5][-1]]], blowup)
    domain = prover.StarkDomain(n, blowup, field=f)
    tl, _ = prover.DefaultTraceLde.new(crypto.Blake3_256, prover.ColMatrix(trace, 1, ctx, f), domain)
    tl.set_aux_trace(prover.ColMatrix(aux, D, ctx, f), domain)
    cc = prover.constraints.ConstraintCompositionCoefficients(rand_e(11).reshape(11, -1), rand_e(10).reshape(10, -1))
    return ctx, fld, f, air, domain, tl, cc, rand, trace, aux


@pytest.mark.parametrize("n,D,blowup", [(64, 1, 8), (64, 2, 8), (256, 2, 4), (1 << 12, 2, 16)])
def test_aux_evaluation_equals_the_oracle(oracle, n, D, blowup):
    from winterfell_amd import prover
    ctx, fld, f, air, domain, tl, cc, rand, trace, aux = _setup(oracle, n, D, 100 + n + D, blowup)
    ev = prover.DefaultConstraintEvaluator(air, cc, D, aux_rand_elements=rand.reshape(3, -1))
    got = ctx.to_host(ev.evaluate(tl, domain))
    # the oracle, from its own LDEs of the two segments
    _, lde, _, _ = fld.build_trace_commitment(0, trace, blowup, 3)
    _, alde, _, _ = fld.build_trace_commitment(0, aux, blowup, 3, D=D)
    assert np.array_equal(tl.main_segment_lde.to_host(), lde) and np.array_equal(tl.aux_segment_lde.to_host(), alde)
    ew = D * 2
    want = fld.evaluate_constraints_full(7, lde, lde.shape[1] // 2, alde, alde.shape[1] // 2, n, blowup, 4, 3, D, cc.transition.reshape(-1),
                                         [(a.column, a.first_step, fld.pack([a.value])) for a in ev.assertions], cc.boundary.reshape(-1)[:8 * ew],
                                         [(a.column, a.first_step, fld.pack(list(a.value))) for a in ev.aux_assertions], cc.boundary.reshape(-1)[8 * ew:],
                                         rand)
    assert np.array_equal(got, want)


def test_aux_entry_point_checks_its_arguments(oracle):
    from winterfell_amd import prover
    from winterfell_amd._lib import ptr
    ctx, fld, f, air, domain, tl, cc, rand, trace, aux = _setup(oracle, 64, 2, 5)
    # a multi-segment AIR without its random elements / a single-segment evaluator call on it
    with pytest.raises(AssertionError):
        prover.DefaultConstraintEvaluator(air, cc, 2)
    lib, out = ctx.lib, ctx.empty_u64(64 * 4 * 4)
    cols, steps = np.zeros(1, dtype=np.uint32), np.zeros(1, dtype=np.uint64)
    vals = np.zeros(4, dtype=np.uint64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    off = f.element_words(f.new(3))
    main, auxm = tl.main_segment_lde, tl.aux_segment_lde
    args = lambda air_id, aux_rw: (ctx.handle, air_id, 1, 2, ptr(main.data), main.row_width, ptr(auxm.data), aux_rw, 6, 3, 2, vp(off), vp(cc.transition), 1,
                                   vp(cols), vp(steps), vp(vals), vp(cc.boundary), 1, vp(cols), vp(steps), vp(vals), vp(cc.boundary), vp(rand), ptr(out))
    assert lib.wf_evaluate_constraints_aux(*args(7, auxm.row_width)) == 0
    assert lib.wf_evaluate_constraints_aux(*args(1, auxm.row_width)) != 0        # RescueAir has no auxiliary segment
    assert lib.wf_evaluate_constraints_aux(*args(7, 4)) != 0                      # rows too narrow for 3 columns of degree 2
    # the single-segment entry point refuses the multi-segment AIR
    assert lib.wf_evaluate_constraints(ctx.handle, 7, 1, 2, ptr(main.data), main.row_width, 6, 3, 2, vp(off), vp(cc.transition), 1, vp(cols), vp(steps),
                                       vp(vals), vp(cc.boundary), ptr(out)) != 0
