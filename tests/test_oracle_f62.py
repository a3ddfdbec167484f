"""Pin the oracle's f62 restatement (math/src/field/f62/mod.rs) against big-int arithmetic and reference edge cases."""
import random

M = 4611624995532046337


def test_f62_field_and_extensions(oracle):
    f = oracle.f62
    R = pow(2, 64, M)
    random.seed(3)
    for _ in range(500):
        a, b = random.randrange(M), random.randrange(M)
        am, bm = oracle.f62_new(a), oracle.f62_new(b)
        assert am == a * R % M
        assert oracle.f62_as_int(f.mul(am, bm)) == a * b % M
        assert oracle.f62_as_int(f.add(am, bm)) == (a + b) % M
        assert oracle.f62_as_int(f.sub(am, bm)) == (a - b) % M
    g = oracle.f62_as_int(f.root_of_unity(39))
    assert g == 4421547261963328785 and pow(g, 2**39, M) == 1 and pow(g, 2**38, M) != 1
    m1 = oracle.f62_new(M - 1)                                     # f62/tests.rs:46-66
    assert oracle.f62_as_int(f.mul(m1, m1)) == 1 and oracle.f62_as_int(f.mul(m1, oracle.f62_new(2))) == M - 2
    a = [random.randrange(M) for _ in range(3)]
    b = [random.randrange(M) for _ in range(3)]
    o = f.ext_mul(3, [oracle.f62_new(x) for x in a], [oracle.f62_new(x) for x in b])   # x^3 + 2x + 2 (mod.rs:339-371)
    c = [0] * 5
    for i in range(3):
        for j in range(3):
            c[i + j] = (c[i + j] + a[i] * b[j]) % M
    r = [(c[0] - 2 * c[3]) % M, (c[1] - 2 * c[3] - 2 * c[4]) % M, (c[2] - 2 * c[4]) % M]
    assert [oracle.f62_as_int(x) for x in o] == r
    o = f.ext_mul(2, [oracle.f62_new(x) for x in a[:2]], [oracle.f62_new(x) for x in b[:2]])   # x^2 - x - 1
    assert [oracle.f62_as_int(x) for x in o] == [(a[0] * b[0] + a[1] * b[1]) % M, (a[0] * b[1] + a[1] * b[0] + a[1] * b[1]) % M]


def test_f62_fft_definition(oracle):
    f = oracle.f62
    n = 64
    random.seed(4)
    pc = [random.randrange(M) for _ in range(n)]
    p = f.pack([oracle.f62_new(x) for x in pc])
    ev = [oracle.f62_as_int(v) for v in f.unpack(f.evaluate_poly(p))]
    w = oracle.f62_as_int(f.root_of_unity(6))
    for k in (0, 1, 5, 63):
        assert ev[k] == sum(c * pow(w, k * i, M) for i, c in enumerate(pc)) % M


def test_f62_cube_mul_reference_vectors(oracle, golden):
    """the reference's own fixed vectors for the cubic extension of f62 (math/src/field/f62/tests.rs:128-187: one product within
    bounds, two "with overflow"), parsed out of the reference's test source by tests/golden/make_golden.py"""
    f = oracle.f62
    cases = golden["reference"]["f62_cube_mul"]
    assert len(cases) == 3 and cases[0]["a"] == [15, 22, 8]
    for c in cases:
        o = f.ext_mul(3, [oracle.f62_new(x) for x in c["a"]], [oracle.f62_new(x) for x in c["b"]])
        assert [oracle.f62_as_int(x) for x in o] == c["out"], c
        o = f.ext_mul(3, [oracle.f62_new(x) for x in c["b"]], [oracle.f62_new(x) for x in c["a"]])      # commutes
        assert [oracle.f62_as_int(x) for x in o] == c["out"], c
