"""Pin the CPU oracle (oracle/) against the reference's fixed vectors and independent derivations."""
import numpy as np

import pytest

from conftest import P, rand_field, splitmix64


def test_field_edge_cases(oracle, golden):
    # math/src/field/f64/tests.rs:64-73
    new, as_int, mul = oracle.f64_new, oracle.f64_as_int, oracle.f64_mul
    m1 = new(P - 1)
    assert as_int(mul(m1, m1)) == golden["reference"]["f64_edge"]["m_minus_1_squared"]
    assert as_int(mul(m1, new(2))) == golden["reference"]["f64_edge"]["m_minus_1_times_2"]
    assert as_int(mul(new((P + 1) // 2), new(2))) == golden["reference"]["f64_edge"]["half_times_2"]
    assert as_int(new(P)) == 0 and as_int(new(P + 5)) == 5


def test_field_vs_bigint(oracle):
    vals = [int(v) for v in splitmix64(0x5EED0001, 200)] + [0, 1, 2, P - 1, P - 2, (P + 1) // 2, 2**32, 2**32 - 1]
    R = pow(2, 64, P)
    for a, b in zip(vals, vals[3:] + vals[:3]):
        am, bm = oracle.f64_new(a), oracle.f64_new(b)
        assert am == a * R % P and am < P           # internal form is the canonical Montgomery residue
        assert oracle.f64_as_int(oracle.f64_mul(am, bm)) == a * b % P
        assert oracle.f64_as_int(oracle.f64_add(am, bm)) == (a + b) % P
        assert oracle.f64_as_int(oracle.f64_sub(am, bm)) == (a - b) % P
        assert oracle.f64_mul(am, bm) < P and oracle.f64_add(am, bm) < P and oracle.f64_sub(am, bm) < P
        if a:
            assert oracle.f64_as_int(oracle.f64_inv(am)) == pow(a, P - 2, P)
    assert oracle.f64_as_int(oracle.f64_exp(oracle.f64_new(7), 12345678901234567)) == pow(7, 12345678901234567, P)


def test_root_of_unity(oracle):
    # f64/mod.rs:255-267: omega_64 = 8, omega_8 = 2^24
    assert oracle.f64_as_int(oracle.f64_root_of_unity(6)) == 8
    assert oracle.f64_as_int(oracle.f64_root_of_unity(3)) == 2**24
    g = oracle.f64_as_int(oracle.f64_root_of_unity(32))
    assert g == 7277203076849721926 and pow(g, 2**32, P) == 1 and pow(g, 2**31, P) != 1


def test_ext_mul_reference_vectors(oracle, golden):
    for case in golden["reference"]["f64_quad_mul"]:
        out = oracle.f64_ext_mul(2, oracle.f64_from_int(case["a"]), oracle.f64_from_int(case["b"]))
        assert list(oracle.f64_to_int(out)) == case["out"]
    for case in golden["reference"]["f64_cube_mul"]:
        out = oracle.f64_ext_mul(3, oracle.f64_from_int(case["a"]), oracle.f64_from_int(case["b"]))
        assert list(oracle.f64_to_int(out)) == case["out"]


def test_twiddles(oracle):
    # math/src/fft/tests.rs:64-73: get_twiddles == permuted power series
    for n in (4, 16, 64, 1024):
        w = pow(7277203076849721926, 2**32 // n, P)
        series = [pow(w, i, P) for i in range(n // 2)]
        expect = [series[oracle.permute_index(n // 2, i)] for i in range(n // 2)]
        assert list(oracle.f64_to_int(oracle.get_twiddles(n))) == expect
        winv = pow(w, n - 1, P)
        series = [pow(winv, i, P) for i in range(n // 2)]
        expect = [series[oracle.permute_index(n // 2, i)] for i in range(n // 2)]
        assert list(oracle.f64_to_int(oracle.get_inv_twiddles(n))) == expect


def test_ntt_golden(oracle, golden):
    d = golden["derived"]
    out = oracle.evaluate_poly(oracle.f64_from_int(list(range(1, 9))))
    assert list(oracle.f64_to_int(out)) == d["f64_ntt8_1_to_8"]
    out = oracle.evaluate_poly(oracle.f64_from_int(d["f64_ntt16_in"]))
    assert list(oracle.f64_to_int(out)) == d["f64_ntt16_out"]
    out = oracle.evaluate_poly_with_offset(oracle.f64_from_int([1, 2, 3, 4]), oracle.f64_new(7), 2)
    assert list(oracle.f64_to_int(out)) == d["f64_lde_1234_b2_o7"]
    out = oracle.evaluate_poly_with_offset(oracle.f64_from_int(d["f64_ntt16_in"]), oracle.f64_new(7), 8)
    assert list(oracle.f64_to_int(out)) == d["f64_lde16_b8_o7"]


def test_ntt_matches_eval_many(oracle):
    # math/src/fft/tests.rs:20-61 restated for f64: fft == polynomial evaluation over the domain
    for n in (4, 8, 16, 1024):
        p = oracle.f64_from_int(splitmix64(0x5EED0001 + n, n))
        ev = oracle.evaluate_poly(p)
        w = oracle.f64_root_of_unity(n.bit_length() - 1)
        x = oracle.f64_new(1)
        for k in range(0, n, max(1, n // 64)):
            x = oracle.f64_exp(w, k)
            assert ev[k] == oracle.poly_eval(p, x)
        back = oracle.interpolate_poly(ev)
        assert np.array_equal(back, p)


def test_offset_roundtrip_and_concurrent_equals_serial(oracle):
    for n, D in ((2048, 1), (4096, 1), (2048, 2), (1024, 3), (8192, 1)):
        p = oracle.f64_from_int(splitmix64(0xABC + n + D, n * D))
        off = oracle.f64_new(7)
        ev = oracle.evaluate_poly_with_offset(p, off, 1, D=D)
        assert np.array_equal(oracle.interpolate_poly_with_offset(ev, off, D=D), p)
        assert np.array_equal(oracle.evaluate_poly(p, D=D, par=True), oracle.evaluate_poly(p, D=D))
        assert np.array_equal(oracle.interpolate_poly(p, D=D, par=True), oracle.interpolate_poly(p, D=D))
        assert np.array_equal(oracle.evaluate_poly_with_offset(p, off, 8, D=D, par=True),
                              oracle.evaluate_poly_with_offset(p, off, 8, D=D))


def test_extension_ntt_is_componentwise(oracle):
    n = 256
    p = oracle.f64_from_int(splitmix64(77, n * 3)).reshape(n, 3)
    ev = oracle.evaluate_poly(p.reshape(-1), D=3).reshape(n, 3)
    for d in range(3):
        assert np.array_equal(ev[:, d], oracle.evaluate_poly(np.ascontiguousarray(p[:, d])))


@pytest.mark.parametrize("hasher,D,log_len,N,rem_deg", [(0, 2, 14, 4, 31), (0, 1, 12, 2, 7), (1, 3, 11, 8, 3), (0, 2, 10, 16, 7)])
def test_parallel_fri_driver_equals_the_serial_functions(oracle, hasher, D, log_len, N, rem_deg):
    """or_fri_build_layers_par (the all-cores CPU side of bench.py's FRI number) against the serial restatement layer by layer
    with an independent DefaultProverChannel: every root, every alpha, the remainder commitment."""
    blowup, n = 8, 1 << log_len
    poly = oracle.f64_from_int(rand_field(50 + log_len + N, (n // blowup) * D))
    ev = oracle.evaluate_poly_with_offset(poly, oracle.f64_new(7), blowup, D=D, par=True)
    roots, alphas = oracle.fri_build_layers_par(hasher, ev, N, blowup, rem_deg, oracle.f64_new(7), D)
    chan = oracle.ProverChannel(hasher, D)
    cur, length = ev.copy(), n
    nl = int(oracle.fri_num_layers(n, N, blowup, rem_deg))
    assert roots.shape[0] == nl + 1 and alphas.shape[0] == nl
    for k in range(nl):
        tr = oracle.transpose_slice(cur, N, D)
        _, nodes = oracle.fri_layer_commit(hasher, tr, N, D)
        assert np.array_equal(nodes[1], roots[k]), "layer %d root" % k
        chan.commit_fri_layer(nodes[1])
        alpha = chan.draw_fri_alpha()
        assert np.array_equal(alpha, alphas[k]), "layer %d alpha" % k
        cur = oracle.apply_drp(tr, N, oracle.f64_new(7), alpha, D)
        length //= N
    _, com = oracle.fri_remainder(hasher, cur, oracle.f64_new(7), blowup, D)
    assert np.array_equal(com, roots[nl])
