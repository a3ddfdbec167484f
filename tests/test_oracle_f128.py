"""Pin the oracle's f128 restatement (math/src/field/f128/mod.rs) and the field-generic template."""
import numpy as np
import pytest

from conftest import P, splitmix64

M128 = 2**128 - 45 * 2**40 + 1


def _rand128(seed, n):
    rng = np.random.default_rng(seed)
    return [(int(a) << 64 | int(b)) % M128 for a, b in zip(rng.integers(0, 2**64, n, dtype=np.uint64),
                                                              rng.integers(0, 2**64, n, dtype=np.uint64))]


def test_f128_field_vs_bigint(oracle):
    f = oracle.f128
    vals = _rand128(1, 200) + [0, 1, 2, M128 - 1, M128 - 2, (M128 + 1) // 2, 2**64, 2**64 - 1, 2**127, 45 << 40]
    for a, b in zip(vals, vals[5:] + vals[:5]):
        assert f.mul(a, b) == a * b % M128
        assert f.add(a, b) == (a + b) % M128
        assert f.sub(a, b) == (a - b) % M128
        if a:
            assert f.mul(f.inv(a), a) == 1
    # math/src/field/f128/tests.rs:56-75 edge cases
    assert f.mul(M128 - 1, M128 - 1) == 1 and f.mul(M128 - 1, 2) == M128 - 2 and f.mul((M128 + 1) // 2, 2) == 1
    g = f.root_of_unity(40)
    assert g == 23953097886125630542083529559205016746 and pow(g, 2**40, M128) == 1 and pow(g, 2**39, M128) != 1
    assert f.exp(3, 12345678901234567) == pow(3, 12345678901234567, M128)
    # quadratic extension x^2 - x - 1 (f128/mod.rs:267-272): (a0 + a1 x)(b0 + b1 x)
    a, b = _rand128(2, 2), _rand128(3, 2)
    o = f.ext_mul(2, a, b)
    assert o[0] == (a[0] * b[0] + a[1] * b[1]) % M128
    assert o[1] == (a[0] * b[1] + a[1] * b[0] + a[1] * b[1]) % M128


def test_f128_fft_matches_definition(oracle):
    # math/src/fft/tests.rs:20-61 (the reference runs exactly this over f128): fft == eval_many
    f = oracle.f128
    for n in (4, 8, 16, 1024):
        pc = _rand128(n, n)
        p = f.pack(pc)
        ev = f.unpack(f.evaluate_poly(p))
        w = f.root_of_unity(n.bit_length() - 1)
        for k in range(0, n, max(1, n // 16)):
            x = pow(w, k, M128)
            assert ev[k] == sum(c * pow(x, i, M128) for i, c in enumerate(pc)) % M128
        assert np.array_equal(f.interpolate_poly(f.evaluate_poly(p)), p)
        ev8 = f.unpack(f.evaluate_poly_with_offset(p, 3, 8))
        g = f.root_of_unity((8 * n).bit_length() - 1)
        for k in (0, 1, 9, 8 * n - 1):
            x = 3 * pow(g, k, M128) % M128
            assert ev8[k] == sum(c * pow(x, i, M128) for i, c in enumerate(pc)) % M128
        assert np.array_equal(f.interpolate_poly_with_offset(f.evaluate_poly_with_offset(p, 3, 1), 3), p)
    tw = f.unpack(f.get_twiddles(16))
    w = f.root_of_unity(4)
    assert tw == [pow(w, oracle.permute_index(8, i), M128) for i in range(8)]


def test_template_instantiated_for_f64_equals_handwritten_f64(oracle):
    t = oracle.f64t
    for n, D in ((256, 1), (512, 2), (128, 3)):
        p = oracle.f64_from_int(splitmix64(n + D, n * D))
        off = oracle.f64_new(7)
        assert np.array_equal(t.evaluate_poly(p, D), oracle.evaluate_poly(p, D))
        assert np.array_equal(t.interpolate_poly(p, D), oracle.interpolate_poly(p, D))
        assert np.array_equal(t.evaluate_poly_with_offset(p, off, 8, D), oracle.evaluate_poly_with_offset(p, off, 8, D))
        assert np.array_equal(t.interpolate_poly_with_offset(p, off, D), oracle.interpolate_poly_with_offset(p, off, D))
    trace = oracle.f64_from_int(splitmix64(3, 5 * 256)).reshape(5, 256)
    for hasher in (0, 1):
        a = t.build_trace_commitment(hasher, trace, 4, oracle.f64_new(7), num_partitions=2, hash_rate=2)
        b = oracle.build_trace_commitment(hasher, trace, 4, oracle.f64_new(7), num_partitions=2, hash_rate=2)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    ev = oracle.f64_from_int(splitmix64(4, 1024 * 2))
    alpha = oracle.f64_from_int(splitmix64(5, 2))
    tr = t.transpose_slice(ev, 4, 2)
    assert np.array_equal(tr, oracle.transpose_slice(ev, 4, 2))
    assert np.array_equal(t.apply_drp(tr, 4, oracle.f64_new(7), alpha, 2), oracle.apply_drp(tr, 4, oracle.f64_new(7), alpha, 2))
    for x, y in zip(t.fri_layer_commit(1, tr, 4, 2), oracle.fri_layer_commit(1, tr, 4, 2)):
        assert np.array_equal(x, y)


def test_f128_trace_commitment_structure(oracle):
    f = oracle.f128
    n, c, b = 64, 3, 4
    trace = f.pack(_rand128(9, n * c)).reshape(c, n * 2)
    polys, lde, leaves, nodes = f.build_trace_commitment(0, trace, b, 3)
    assert lde.shape == (n * b, 8 * 2) and not lde[:, 6:].any()
    g = f.root_of_unity((n * b).bit_length() - 1)
    for r in (0, 5, n * b - 1):
        x = 3 * pow(g, r, M128) % M128
        assert f.unpack(lde[r, :2])[0] == f.poly_eval(polys[0], x)
        assert leaves[r].tobytes() == oracle.blake3(lde[r, :6].tobytes())     # raw bytes: IS_CANONICAL (blake/mod.rs:53-57)
    assert np.array_equal(nodes, oracle.merkle_build(0, leaves))


@pytest.mark.parametrize("N", [2, 4, 8, 16])
def test_apply_drp_equals_folding_in_coefficient_form(oracle, N):
    """The reference's DEFINITION of the degree-respecting projection (fri/src/folding/mod.rs:46-85, the doc example of apply_drp,
    there for N = 2 over f128): folding the polynomial in coefficient form — f'(x) = sum_k alpha^k f_k(x), f_k = the coefficients
    = k (mod N) — and evaluating f' over the folded domain {offset^N w^i} gives what apply_drp computes from the transposed
    evaluations.  Python integers on this side, the oracle's transpose_slice + apply_drp on the other: pins the oracle's FRI fold to
    the reference's own statement instead of to the in-repo verifier."""
    import random
    f, M = oracle.f128, oracle.F128_M
    rng = random.Random(100 + N)
    n, deg1 = 8 * N, 2 * N                     # domain of n points, a polynomial with deg1 coefficients (blowup 4)
    alpha, offset = rng.randrange(M), 3        # BaseElement::GENERATOR of f128 is 3
    poly = [rng.randrange(M) for _ in range(deg1)]
    folded = [sum(pow(alpha, k, M) * poly[N * i + k] for k in range(N)) % M for i in range(deg1 // N)]
    g = int(f.root_of_unity(n.bit_length() - 1))
    domain = [offset * pow(g, i, M) % M for i in range(n)]
    gf = pow(g, N, M)
    fdomain = [pow(offset, N, M) * pow(gf, i, M) % M for i in range(n // N)]
    ev = [sum(c * pow(x, j, M) for j, c in enumerate(poly)) % M for x in domain]
    fev = [sum(c * pow(x, j, M) for j, c in enumerate(folded)) % M for x in fdomain]
    got = f.apply_drp(f.transpose_slice(f.pack(ev), N), N, offset, f.pack([alpha]))
    assert [int(v) for v in f.unpack(got)] == fev


@pytest.mark.parametrize("N", [2, 4, 8, 16])
def test_apply_drp_definition_over_f64(oracle, N):
    """the same definitional check for the hand-written f64 fold (oracle/fri.c), Montgomery words in and out"""
    import random
    rng = random.Random(200 + N)
    n, deg1 = 8 * N, 2 * N
    alpha, offset = rng.randrange(P), 7
    poly = [rng.randrange(P) for _ in range(deg1)]
    folded = [sum(pow(alpha, k, P) * poly[N * i + k] for k in range(N)) % P for i in range(deg1 // N)]
    g = int(oracle.f64_as_int(oracle.f64_root_of_unity(n.bit_length() - 1)))
    domain = [offset * pow(g, i, P) % P for i in range(n)]
    fdomain = [pow(offset, N, P) * pow(pow(g, N, P), i, P) % P for i in range(n // N)]
    ev = [sum(c * pow(x, j, P) for j, c in enumerate(poly)) % P for x in domain]
    fev = [sum(c * pow(x, j, P) for j, c in enumerate(folded)) % P for x in fdomain]
    evm = oracle.f64_from_int(np.array(ev, dtype=np.uint64))
    got = oracle.apply_drp(oracle.transpose_slice(evm, N), N, oracle.f64_new(offset), np.array([oracle.f64_new(alpha)], dtype=np.uint64))
    assert [int(v) for v in oracle.f64_to_int(got)] == fev
