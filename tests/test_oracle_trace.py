"""Pin the oracle's matrix / trace-commitment restatement (prover/src/matrix, trace_lde/default)."""
import numpy as np

from conftest import P, splitmix64


def test_fib_trace_lde_fixture(oracle, golden):
    # prover/src/trace/trace_lde/default/tests.rs:22-106 with build_fib_trace(16) (prover/src/tests/mod.rs:19-31)
    c0, c1 = golden["reference"]["fib_trace_col0"], golden["reference"]["fib_trace_col1"]
    trace = np.stack([oracle.f64_from_int(c0), oracle.f64_from_int(c1)])
    blowup, offset = 8, oracle.f64_new(7)
    polys, lde, leaves, nodes = oracle.build_trace_commitment(oracle.H_BLAKE3_F64, trace, blowup, offset)
    n, N = 8, 64
    # trace polynomials evaluate back to the trace over the trace domain
    w = oracle.f64_root_of_unity(3)
    for col, expect in ((0, c0), (1, c1)):
        for i in range(n):
            assert oracle.f64_as_int(oracle.poly_eval(polys[col], oracle.f64_exp(w, i))) == expect[i]
    # LDE == evaluations of the polynomials over the shifted LDE domain, row-major, padded to 8 columns
    assert lde.shape == (N, 8) and not lde[:, 2:].any()
    g = oracle.f64_root_of_unity(6)
    for r in range(N):
        x = oracle.f64_mul(offset, oracle.f64_exp(g, r))
        assert lde[r, 0] == oracle.poly_eval(polys[0], x) and lde[r, 1] == oracle.poly_eval(polys[1], x)
    # every blowup-th row is the original trace
    assert list(oracle.f64_to_int(oracle.interpolate_poly_with_offset(np.ascontiguousarray(lde[:, 0]), offset)[:8])) \
        == list(oracle.f64_to_int(polys[0]))
    # commitment == MerkleTree over Blake3::hash_elements(row of the 2 real columns)
    for r in (0, 1, 17, 63):
        row_bytes = b"".join(int(oracle.f64_as_int(int(v))).to_bytes(8, "little") for v in lde[r, :2])
        assert leaves[r].tobytes() == oracle.blake3(row_bytes)
    assert np.array_equal(nodes, oracle.merkle_build(oracle.H_BLAKE3_F64, leaves))


def test_lde_matches_eval_many_64cols(oracle):
    # prover/src/matrix/tests.rs test_eval_poly_with_offset_matrix: 64 f64 polys, n=256, blowup 8
    n, c, b = 256, 64, 8
    polys = oracle.f64_from_int(splitmix64(0x5EED0100 + c, n * c)).reshape(c, n)
    off = oracle.f64_new(7)
    lde = oracle.evaluate_polys_over(polys, b, off)
    assert lde.shape == (n * b, 64)
    g = oracle.f64_root_of_unity((n * b).bit_length() - 1)
    for r in (0, 1, 7, 8, 1000, n * b - 1):
        x = oracle.f64_mul(off, oracle.f64_exp(g, r))
        for col in (0, 1, 31, 63):
            assert lde[r, col] == oracle.poly_eval(polys[col], x)


def test_partitioned_row_hash(oracle):
    # row_matrix.rs:204-223 + air/src/options.rs:428-444
    assert oracle.partition_size(1, 8, 1, 10) == 10
    assert oracle.partition_size(4, 8, 1, 64) == 16
    assert oracle.partition_size(4, 8, 1, 10) == 8        # min partition size = hash_rate / D
    assert oracle.partition_size(4, 8, 2, 10) == 4
    N, c = 32, 20
    data = oracle.f64_from_int(splitmix64(4, N * 24)).reshape(N, 24)
    for hasher in (0, 1):
        leaves = oracle.hash_rows(hasher, data, c, num_partitions=4, hash_rate=4)
        ps = oracle.partition_size(4, 4, 1, c)
        assert ps == 5
        for r in (0, 13, 31):
            parts = [oracle.hash_elements(hasher, data[r, k * ps:(k + 1) * ps]) for k in range(4)]
            assert np.array_equal(leaves[r], oracle.merge_many(hasher, np.stack(parts)))
        plain = oracle.hash_rows(hasher, data, c)
        assert np.array_equal(plain[5], oracle.hash_elements(hasher, data[5, :c]))
        assert not np.array_equal(plain, leaves)


def test_concurrent_pipeline_equals_serial(oracle):
    n, c = 2048, 5
    trace = oracle.f64_from_int(splitmix64(11, n * c)).reshape(c, n)
    a = oracle.build_trace_commitment(oracle.H_RP64, trace, 2, oracle.f64_new(7))
    b = oracle.build_trace_commitment(oracle.H_RP64, trace, 2, oracle.f64_new(7), par=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
