"""Pin the oracle's BLAKE3 / Rp64_256 / Merkle restatements."""
import ctypes

import numpy as np
import pytest

from conftest import P, splitmix64


def _llvm_blake3():
    for cand in ("/usr/lib/x86_64-linux-gnu/libLLVM-15.so.1", "/opt/rocm/lib/llvm/lib/libclang-cpp.so"):
        try:
            l = ctypes.CDLL(cand)
            l.llvm_blake3_hasher_init
            return l
        except (OSError, AttributeError):
            continue
    return None


def test_blake3_golden(oracle, golden):
    for case in golden["derived"]["blake3"]:
        assert oracle.blake3(bytes.fromhex(case["in_hex"])).hex() == case["out_hex"]


def test_blake3_vs_upstream_c(oracle):
    """Independent oracle: the upstream BLAKE3 C implementation bundled in LLVM."""
    l = _llvm_blake3()
    if l is None:
        pytest.skip("no LLVM-bundled BLAKE3 in this image")
    rng = np.random.default_rng(3)
    for ln in list(range(0, 130)) + [1023, 1024, 1025, 2047, 2048, 2049, 3072, 4095, 4096, 4097, 5000, 8192, 9000]:
        data = rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes()
        st = ctypes.create_string_buffer(4096)
        l.llvm_blake3_hasher_init(st)
        l.llvm_blake3_hasher_update(st, data, ctypes.c_size_t(ln))
        out = ctypes.create_string_buffer(32)
        l.llvm_blake3_hasher_finalize(st, out, ctypes.c_size_t(32))
        assert oracle.blake3(data) == out.raw, ln


def test_blake3_hasher_surface(oracle, golden):
    # blake/tests.rs: merge == merge_many for two digests; hash_elements serialises canonical LE bytes
    d = golden["derived"]
    h = oracle.hash_elements(oracle.H_BLAKE3_F64, oracle.f64_from_int([1, 2]))
    assert h.tobytes().hex() == d["blake3_f64_hash_elements_1_2"]
    lv = np.array(golden["reference"]["LEAVES4"], dtype=np.uint8)
    assert np.array_equal(oracle.merge(oracle.H_BLAKE3_F64, lv[:2]), oracle.merge_many(oracle.H_BLAKE3_F64, lv[:2]))
    seed = lv[0]
    expect = oracle.blake3(seed.tobytes() + (123456789).to_bytes(8, "little"))
    assert oracle.merge_with_int(oracle.H_BLAKE3_F64, seed, 123456789).tobytes() == expect


def test_rp64_permutation_kat(oracle, golden):
    # crypto/src/hash/rescue/rp64_256/tests.rs:70-105
    st = oracle.f64_from_int(golden["reference"]["rp64_256_permutation_in"])
    out = oracle.rp64_apply_permutation(st)
    assert list(oracle.f64_to_int(out)) == golden["reference"]["rp64_256_permutation_out"]


def test_rp64_mds_freq_equals_naive(oracle):
    # tests.rs proptest: frequency-domain MDS == dense MDS (compare as field elements)
    for seed in range(20):
        st = oracle.f64_from_int(splitmix64(seed, 12))
        a = oracle.f64_to_int(oracle.rp64_mds(st))
        b = oracle.f64_to_int(oracle.rp64_mds(st, naive=True))
        assert np.array_equal(a, b)
    edge = oracle.f64_from_int([P - 1] * 12)
    assert np.array_equal(oracle.f64_to_int(oracle.rp64_mds(edge)), oracle.f64_to_int(oracle.rp64_mds(edge, naive=True)))


def test_rp64_sponge(oracle, golden):
    d = golden["derived"]
    z = np.zeros(8, dtype=np.uint64)
    assert list(oracle.f64_to_int(oracle.merge(oracle.H_RP64, z).view(np.uint64))) == d["rp64_merge_zero"]
    h = oracle.hash_elements(oracle.H_RP64, oracle.f64_from_int([1, 2, 3, 4])).view(np.uint64)
    assert list(oracle.f64_to_int(h)) == d["rp64_hash_elements_1_2_3_4"]
    h = oracle.hash_elements(oracle.H_RP64, oracle.f64_from_int(list(range(19)))).view(np.uint64)
    assert list(oracle.f64_to_int(h)) == d["rp64_hash_elements_0_to_18"]
    # tests.rs hash_elements_vs_merge: merge == hash_elements of the 8 elements
    e = oracle.f64_from_int(splitmix64(5, 8))
    assert np.array_equal(oracle.merge(oracle.H_RP64, e), oracle.hash_elements(oracle.H_RP64, e))
    # tests.rs merge_vs_merge_many
    assert np.array_equal(oracle.merge(oracle.H_RP64, e), oracle.merge_many(oracle.H_RP64, e))
    # tests.rs hash_elements_vs_merge_with_int
    seed = oracle.f64_from_int(splitmix64(6, 4))
    for val in (7, P - 1):
        exp = oracle.hash_elements(oracle.H_RP64, np.concatenate([seed, oracle.f64_from_int([val])]))
        assert np.array_equal(oracle.merge_with_int(oracle.H_RP64, seed, val), exp)
    val = P + 2
    exp = oracle.hash_elements(oracle.H_RP64, np.concatenate([seed, oracle.f64_from_int([val % P, val // P])]))
    assert np.array_equal(oracle.merge_with_int(oracle.H_RP64, seed, val), exp)


def test_merkle_fixed_leaves(oracle, golden):
    # crypto/src/merkle/tests.rs:68-86 (new_tree) with roots pinned by the independent BLAKE3
    for name in ("LEAVES4", "LEAVES8"):
        lv = np.array(golden["reference"][name], dtype=np.uint8)
        nodes = oracle.merkle_build(oracle.H_BLAKE3_F64, lv)
        assert nodes[1].tobytes().hex() == golden["derived"]["blake3_root_" + name]
        assert not nodes[0].any()
        n = lv.shape[0]
        for i in range(n // 2):
            assert nodes[n // 2 + i].tobytes() == oracle.blake3(lv[2 * i].tobytes() + lv[2 * i + 1].tobytes())
        for i in range(1, n // 2):
            assert nodes[i].tobytes() == oracle.blake3(nodes[2 * i].tobytes() + nodes[2 * i + 1].tobytes())


def test_merkle_errors_and_concurrent(oracle):
    rng = np.random.default_rng(1)
    with pytest.raises(ValueError, match="TooFewLeaves"):
        oracle.merkle_build(0, rng.integers(0, 256, (1, 32), dtype=np.uint8))
    with pytest.raises(ValueError, match="NotPowerOfTwo"):
        oracle.merkle_build(0, rng.integers(0, 256, (6, 32), dtype=np.uint8))
    # merkle/concurrent.rs:87-95 proptest: concurrent == serial
    for hasher, n in ((0, 4096), (1, 2048)):
        lv = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        if hasher == 1:
            lv = oracle.f64_from_int(splitmix64(9, n * 4)).view(np.uint8).reshape(n, 32)
        assert np.array_equal(oracle.merkle_build(hasher, lv, par=True), oracle.merkle_build(hasher, lv))


def test_sha3_256_pinned_to_hashlib(oracle):
    """The reference's Sha3_256 hasher wraps the `sha3` crate (crypto/src/hash/sha/mod.rs:21-66, not vendored); the
    oracle restates FIPS 202 and is pinned here against Python's independent implementation, through the raw byte hash
    and through every Hasher entry point of hasher id 2."""
    import ctypes
    import hashlib
    lib = oracle.lib()

    def sha3(b):
        out = np.empty(32, dtype=np.uint8)
        a = np.frombuffer(b, dtype=np.uint8).copy() if len(b) else np.zeros(1, dtype=np.uint8)
        lib.or_sha3_256(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(b)), out.ctypes.data_as(ctypes.c_void_p))
        return out.tobytes()

    for n in (0, 1, 7, 8, 31, 32, 40, 64, 135, 136, 137, 271, 272, 273, 1000, 4097):
        b = bytes((i * 7 + 3) & 255 for i in range(n))
        assert sha3(b) == hashlib.sha3_256(b).digest(), n
    assert sha3(b"") == bytes.fromhex("a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a")   # FIPS 202 KAT
    assert sha3(b"abc") == bytes.fromhex("3a985da74fe225b2045c172d6bd390bd855f086e3e9d525b46bfe24511431532")
    ints = np.arange(1, 40, dtype=np.uint64)
    el = oracle.f64_from_int(ints)
    assert oracle.hash_elements(2, el).tobytes() == hashlib.sha3_256(ints.tobytes()).digest()      # canonical LE bytes
    two = np.arange(64, dtype=np.uint8).reshape(2, 32)
    assert oracle.merge(2, two).tobytes() == hashlib.sha3_256(two.tobytes()).digest()
    assert oracle.merge_with_int(2, two[0], 0x0102030405060708).tobytes() == \
        hashlib.sha3_256(two[0].tobytes() + (0x0102030405060708).to_bytes(8, "little")).digest()
    many = np.arange(96, dtype=np.uint8).reshape(3, 32)
    assert oracle.merge_many(2, many).tobytes() == hashlib.sha3_256(many.tobytes()).digest()


def test_rpjive64_permutation_kat_and_properties(oracle):
    """crypto/src/hash/rescue/rp64_256_jive/tests.rs:69-97 (apply_permutation known answer from the sage reference
    implementation) and the structural tests :99-178 (Jive merge differs from the sponge, padding)."""
    import ctypes
    lib = oracle.lib()
    st = np.array([oracle.f64_new(i) for i in range(8)], dtype=np.uint64)
    lib.or_rpjive_apply_permutation(st.ctypes.data_as(ctypes.c_void_p))
    assert [int(oracle.f64_as_int(int(v))) for v in st] == [
        16940713730596720799, 16218555904323712189, 11042680722444601138, 5370396747047489939,
        6349480890410006944, 1551053614279730715, 3995941143622927528, 9350074312471431779]
    H = 3
    el = oracle.f64_from_int(np.arange(11, 19, dtype=np.uint64))
    two = el.view(np.uint8).reshape(2, 32)
    assert not np.array_equal(oracle.merge(H, two), oracle.hash_elements(H, el))                  # tests.rs:99-114
    seed = two[0]
    assert not np.array_equal(oracle.merge_with_int(H, seed, 77), oracle.hash_elements(H, np.append(el[:4], np.uint64(oracle.f64_new(77)))))
    e2 = oracle.f64_from_int(np.array([5, 6], dtype=np.uint64))
    assert not np.array_equal(oracle.hash_elements(H, e2), oracle.hash_elements(H, np.append(e2, np.uint64(0))))   # tests.rs:169-178
    # merge = Jive: initial halves + permuted halves
    st = el.copy()
    lib.or_rpjive_apply_permutation(st.ctypes.data_as(ctypes.c_void_p))
    P_ = 0xFFFFFFFF00000001
    want = [(int(el[i]) + int(el[4 + i]) + int(st[i]) + int(st[4 + i])) % P_ for i in range(4)]   # Montgomery form is linear
    assert [int(v) for v in oracle.merge(H, two).view(np.uint64)] == want
    # merge_many = hash_elements over the digests' elements (mod.rs:219-221)
    many = oracle.f64_from_int(np.arange(1, 13, dtype=np.uint64))
    assert np.array_equal(oracle.merge_many(H, many.view(np.uint8).reshape(3, 32)), oracle.hash_elements(H, many))


def test_rp62_248_permutation_kat_and_properties(oracle):
    """crypto/src/hash/rescue/rp62_248/tests.rs:34-70 (apply_permutation known answer) and :72-125 (hash_elements of 8
    elements == merge of the two digests, merge == merge_many, merge_with_int == hash_elements of seed + value)."""
    import ctypes
    lib = oracle.lib()
    st = np.array([oracle.f62_new(i) for i in range(12)], dtype=np.uint64)
    lib.or_rp62_apply_permutation(st.ctypes.data_as(ctypes.c_void_p))
    assert [int(oracle.f62_as_int(int(v))) for v in st] == [
        2176593392043442589, 3663362000910009411, 2446978550600442325, 4214718471639678996, 4179776369445579812,
        2274316532403536457, 2336761070419368662, 3192888412646553651, 4092565229845701133, 753437048204208885,
        4067414342325289862, 3516613610105678931]
    H, f = 4, oracle.f62
    el = np.array([oracle.f62_new(100 + i) for i in range(8)], dtype=np.uint64)
    two = el.view(np.uint8).reshape(2, 32)
    h8 = np.empty(32, dtype=np.uint8)
    lib.or_rp62_hash_elements(el.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(8), h8.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(oracle.merge(H, two), h8)                                               # tests.rs:72-84
    assert np.array_equal(oracle.merge_many(H, two), h8)                                          # tests.rs:86-98
    seed = two[0]
    e5 = np.append(el[:4], np.uint64(oracle.f62_new(12345)))
    h5 = np.empty(32, dtype=np.uint8)
    lib.or_rp62_hash_elements(e5.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(5), h5.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(oracle.merge_with_int(H, seed, 12345), h5)                              # tests.rs:100-112
    big = oracle.F62_M + 2
    e6 = np.append(el[:4], [np.uint64(oracle.f62_new(big)), np.uint64(oracle.f62_new(1))])
    h6 = np.empty(32, dtype=np.uint8)
    lib.or_rp62_hash_elements(e6.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(6), h6.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(oracle.merge_with_int(H, seed, big), h6)                                # tests.rs:114-124


def test_blake3_192_is_truncated_blake3(oracle):
    """crypto/src/hash/blake/mod.rs:68-125: every Blake3_192 entry point is BLAKE3 of the reference's byte string,
    truncated to 24 bytes (digests occupy 32-byte slots with a zero tail in the oracle and in the library)."""
    import ctypes
    lib = oracle.lib()

    def b3(b):
        out = np.empty(32, dtype=np.uint8)
        a = np.frombuffer(b, dtype=np.uint8).copy()
        lib.or_blake3_hash(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(b)), out.ctypes.data_as(ctypes.c_void_p))
        return out.tobytes()

    H = 5
    ints = np.arange(3, 30, dtype=np.uint64)
    d = oracle.hash_elements(H, oracle.f64_from_int(ints))
    assert d.tobytes() == b3(ints.tobytes())[:24] + bytes(8)
    two = np.arange(64, dtype=np.uint8).reshape(2, 32)
    assert oracle.merge(H, two).tobytes() == b3(two[0, :24].tobytes() + two[1, :24].tobytes())[:24] + bytes(8)
    assert oracle.merge_with_int(H, two[0], 0x1122334455667788).tobytes() == \
        b3(two[0, :24].tobytes() + (0x1122334455667788).to_bytes(8, "little"))[:24] + bytes(8)
    many = np.arange(96, dtype=np.uint8).reshape(3, 32)
    assert oracle.merge_many(H, many).tobytes() == b3(b"".join(many[i, :24].tobytes() for i in range(3)))[:24] + bytes(8)
