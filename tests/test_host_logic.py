"""CPU tests of the host-side mirror: representation helpers, PartitionOptions, Merkle openings (index logic)."""
import os

import numpy as np
import pytest

from conftest import P, ROOT, splitmix64


def test_field_representation_helpers(oracle):
    from winterfell_amd.math import fields
    vals = [int(v) for v in splitmix64(1, 50)] + [0, 1, P - 1]
    for v in vals:
        assert fields.new(v) == oracle.f64_new(v) and fields.as_int(fields.new(v)) == v
    arr = np.array(vals, dtype=np.uint64)
    assert np.array_equal(fields.from_ints(arr), oracle.f64_from_int(arr))
    assert np.array_equal(fields.to_ints(fields.from_ints(arr)), arr)


def test_partition_options(oracle):
    from winterfell_amd.prover import PartitionOptions
    for parts, rate, D, cols in ((1, 8, 1, 10), (4, 8, 1, 64), (4, 8, 1, 10), (4, 8, 2, 10), (16, 4, 3, 255), (8, 8, 1, 64)):
        po = PartitionOptions(parts, rate)
        assert po.partition_size(cols, D) == oracle.partition_size(parts, rate, D, cols)
    # the reference's own vectors (air/src/options.rs:555-586, `correct_partition_sizes`): (partitions, hash rate, extension degree,
    # columns) -> (partition_size, num_partitions), for the host mirror and for the oracle's restatement
    for parts, rate, D, cols, size, num in ((4, 8, 1, 7, 8, 1), (4, 8, 1, 70, 18, 4), (2, 8, 3, 7, 4, 2), (4, 8, 3, 7, 2, 4), (4, 8, 3, 3, 2, 2)):
        po = PartitionOptions(parts, rate)
        assert po.partition_size(cols, D) == size and po.num_partitions_for(cols, D) == num, (parts, rate, D, cols)
        assert oracle.partition_size(parts, rate, D, cols) == size
    with pytest.raises(AssertionError):
        PartitionOptions(17, 1)          # air/src/options.rs:414
    with pytest.raises(AssertionError):
        PartitionOptions(0, 1)


def test_merkle_openings_on_host_nodes(oracle, golden):
    """prove / prove_batch are index walks over the reference heap layout (crypto/src/merkle/tests.rs:88-186);
    run them over oracle-built nodes (no GPU needed)."""
    from winterfell_amd.crypto import MerkleTree, MerkleTreeError, Blake3_256
    lv = np.array(golden["reference"]["LEAVES8"], dtype=np.uint8)
    tree = MerkleTree(Blake3_256, None, None, None)
    tree._leaves, tree._nodes = lv, oracle.merkle_build(0, lv)
    h2 = lambda a, b: oracle.merge(0, np.stack([a, b]))
    leaf, proof = tree.prove(1)
    want = [lv[0], h2(lv[2], lv[3]), h2(h2(lv[4], lv[5]), h2(lv[6], lv[7]))]
    assert np.array_equal(leaf, lv[1]) and all(np.array_equal(x, y) for x, y in zip(proof, want))
    leaves, bp = tree.prove_batch([1, 2])
    assert [len(x) for x in bp.nodes] == [2, 1] and bp.depth == 3
    assert np.array_equal(bp.nodes[0][1], h2(h2(lv[4], lv[5]), h2(lv[6], lv[7])))
    leaves, bp = tree.prove_batch([1, 6])
    assert np.array_equal(bp.nodes[0][0], lv[0]) and np.array_equal(bp.nodes[1][0], lv[7])
    leaves, bp = tree.prove_batch(list(range(8)))
    assert all(len(x) == 0 for x in bp.nodes) and np.array_equal(np.stack(leaves), lv)
    with pytest.raises(MerkleTreeError, match="OutOfBounds"):
        tree.prove(8)
    with pytest.raises(MerkleTreeError, match="Duplicate"):
        tree.prove_batch([1, 1])
    with pytest.raises(MerkleTreeError, match="TooFewLeafIndexes"):
        tree.prove_batch([])


class _OracleHasher:
    """Hasher shim over the oracle so the host-side proof walks can run without a GPU."""
    def __init__(self, oracle, hid):
        self.o, self.hid = oracle, hid

    def merge(self, values, ctx=None):
        v = np.ascontiguousarray(values).view(np.uint8).reshape(-1, 2, 32)
        out = np.stack([self.o.merge(self.hid, pair) for pair in v])
        return out[0] if np.asarray(values).size == 64 else out


@pytest.mark.parametrize("hid", [0, 1])
def test_batch_proof_get_root_and_verify_batch(oracle, hid):
    """crypto/src/merkle/tests.rs:188-254 (verify_batch over many index sets) + proofs.rs get_root error paths."""
    from winterfell_amd.crypto import MerkleTree, MerkleTreeError
    from conftest import splitmix64
    h = _OracleHasher(oracle, hid)
    for log_n, seed in ((1, 1), (3, 2), (5, 3), (7, 4)):
        n = 1 << log_n
        lv = (splitmix64(seed, n * 4) >> np.uint64(2)).view(np.uint8).reshape(n, 32)   # valid f64 words for Rp64
        tree = MerkleTree(h, None, None, None)
        tree._leaves, tree._nodes = lv, oracle.merkle_build(hid, lv)
        root = tree.root()
        rng = np.random.default_rng(seed)
        sets = [[0], [n - 1], list(range(n)), [0, n - 1]] + [sorted(rng.choice(n, size=rng.integers(1, n + 1), replace=False).tolist())
                                                              for _ in range(12)]
        sets.append(list(reversed(sets[-1])))                                       # unsorted index lists are allowed
        for idx in sets:
            leaves, proof = tree.prove_batch(idx)
            assert MerkleTree.verify_batch(h, root, idx, leaves, proof) is None
            assert np.array_equal(proof.get_root(h, idx, leaves), root)
            if n > 2:
                bad = [l.copy() for l in leaves]
                bad[0][0] ^= 1
                with pytest.raises(MerkleTreeError, match="InvalidProof"):
                    MerkleTree.verify_batch(h, root, idx, bad, proof)
        leaves, proof = tree.prove_batch([0])
        with pytest.raises(MerkleTreeError, match="TooFewLeafIndexes"):
            proof.get_root(h, [], leaves)
        with pytest.raises(MerkleTreeError, match="Duplicate"):
            proof.get_root(h, [0, 0], leaves + leaves)
        if n > 2:
            with pytest.raises(MerkleTreeError, match="InvalidProof"):
                proof.get_root(h, [0, 2], leaves + leaves)                           # wrong number of proof chains


def test_fri_options(oracle):
    """fri/src/options.rs:85-93 num_fri_layers incl. SURVEY D4 (folding 4/2 land on 2^8, folding 8 on 2^6)."""
    from winterfell_amd.fri import FriOptions
    for fold, rem, size in ((4, 31, 1 << 24), (2, 31, 1 << 24), (8, 31, 1 << 24), (4, 255, 1 << 16), (16, 7, 1 << 12), (2, 0, 8)):
        o = FriOptions(8, fold, rem)
        assert o.num_fri_layers(size) == oracle.fri_num_layers(size, fold, 8, rem)
    assert FriOptions(8, 4, 31).num_fri_layers(1 << 24) == 8 and FriOptions(8, 8, 31).num_fri_layers(1 << 24) == 6
    with pytest.raises(AssertionError):
        FriOptions(8, 3, 31)          # fri/src/options.rs:37-44
    from winterfell_amd.math import fields
    assert FriOptions(8, 4, 31).domain_offset() == fields.new(7)
    assert FriOptions(8, 4, 31, field=fields.f128).domain_offset() == 3


def test_field_descriptors():
    from winterfell_amd.math import fields
    assert fields.f64.M == 2**64 - 2**32 + 1 and fields.f128.M == 2**128 - 45 * 2**40 + 1 and fields.f62.M == 2**62 - 111 * 2**39 + 1
    for f in (fields.f64, fields.f128, fields.f62):
        assert (f.M - 1) % (1 << f.TWO_ADICITY) == 0 and (f.M - 1) // (1 << f.TWO_ADICITY) % 2 == 1
        for v in (0, 1, 12345, f.M - 1):
            assert f.as_int(f.new(v)) == v
        assert f.unpack(f.pack([f.new(5), f.new(f.M - 2)])) == [f.new(5), f.new(f.M - 2)]
    assert fields.f128.W == 2 and fields.f128.new(7) == 7          # canonical representation
    assert fields.f62.new(1) == (1 << 64) % fields.f62.M            # Montgomery form


def test_roots_of_unity_match_reference_constants(oracle):
    """get_root_of_unity follows the reference's TWO_ADIC_ROOT_OF_UNITY constants (f64/mod.rs:267 — note that this is
    NOT 7^((M-1)/2^32) —, f128/mod.rs:43, f62/mod.rs:54); the oracle restates the same constants independently."""
    from winterfell_amd.math import fields
    for f, o in ((fields.f64, oracle.f64t), (fields.f128, oracle.f128), (fields.f62, oracle.f62)):
        for n in (1, 2, 5, 16, f.TWO_ADICITY):
            w = f.get_root_of_unity(n)
            assert f.new(w) == o.root_of_unity(n)
            assert pow(w, 1 << n, f.M) == 1 and pow(w, 1 << (n - 1), f.M) == f.M - 1
    assert fields.f64.get_root_of_unity(6) == 8                        # omega_64 = 8: the shift-twiddle fact the NTT uses
    with pytest.raises(AssertionError):
        fields.f64.get_root_of_unity(33)


@pytest.mark.parametrize("hid", [0, 1])
def test_batch_proof_from_single_proofs_and_into_openings(oracle, hid):
    """crypto/src/merkle/tests.rs:239-313 (from_proofs, batch_proof_from_proofs, verify_into_openings, into_openings):
    aggregating the single openings gives exactly prove_batch's proof, and a batch proof decompresses into exactly the
    single openings — over many random index sets, plus from_raw_parts (mod.rs:148-160)."""
    from winterfell_amd.crypto import BatchMerkleProof, MerkleTree, MerkleTreeError
    from conftest import splitmix64
    h = _OracleHasher(oracle, hid)
    rng = np.random.default_rng(11 + hid)
    for log_n in (1, 2, 5, 7):
        n = 1 << log_n
        lv = (splitmix64(100 + log_n, n * 4) >> np.uint64(2)).view(np.uint8).reshape(n, 32)
        tree = MerkleTree.from_raw_parts(h, oracle.merkle_build(hid, lv), lv)
        assert tree.depth() == log_n and np.array_equal(tree.root(), oracle.merkle_build(hid, lv)[1])
        for _ in range(12):
            idx = sorted(set(int(v) for v in rng.integers(0, n, rng.integers(1, min(n, 20) + 1))))
            leaves, bp = tree.prove_batch(idx)
            singles = [tree.prove(i) for i in idx]
            assert BatchMerkleProof.from_single_proofs(singles, idx) == bp, (log_n, idx)
            # unsorted input order gives the same aggregate (the reference sorts through a BTreeMap)
            perm = list(rng.permutation(len(idx)))
            assert BatchMerkleProof.from_single_proofs([singles[k] for k in perm], [idx[k] for k in perm]) == bp
            opened = bp.into_openings(h, leaves, idx)
            for (leaf, path), (want_leaf, want_path) in zip(opened, singles):
                assert np.array_equal(leaf, want_leaf) and len(path) == len(want_path)
                assert all(np.array_equal(a, b) for a, b in zip(path, want_path))
            for i, (leaf, path) in zip(idx, opened):
                assert MerkleTree.verify(h, tree.root(), i, leaf, path) is None
    with pytest.raises(AssertionError):
        BatchMerkleProof.from_single_proofs([], [])
    with pytest.raises(AssertionError):
        BatchMerkleProof.from_single_proofs(singles, idx[:-1] if len(idx) > 1 else [])
    with pytest.raises(MerkleTreeError, match="TooFewLeafIndexes"):
        bp.into_openings(h, [], [])
    with pytest.raises(MerkleTreeError, match="InvalidProof"):
        bp.into_openings(h, leaves, idx + [0] if 0 not in idx else idx[:-1] + [])
    with pytest.raises(MerkleTreeError, match="TooFewLeaves"):
        MerkleTree.from_raw_parts(h, lv[:1], lv[:1])
    with pytest.raises(MerkleTreeError, match="NotPowerOfTwo"):
        MerkleTree.from_raw_parts(h, lv[:3], lv[:3])


def test_vector_commitment_names(oracle):
    """VectorCommitment for MerkleTree (crypto/src/merkle/mod.rs:401-458): same results under the trait's names."""
    from winterfell_amd.crypto import MerkleTree
    from conftest import splitmix64
    h = _OracleHasher(oracle, 0)
    lv = splitmix64(9, 16 * 4).view(np.uint8).reshape(16, 32)
    tree = MerkleTree.from_raw_parts(h, oracle.merkle_build(0, lv), lv)
    assert np.array_equal(tree.commitment(), tree.root()) and tree.domain_len() == 16
    item, proof = tree.open(5)
    assert MerkleTree.get_proof_domain_len(proof) == 16 and MerkleTree.verify(h, tree.commitment(), 5, item, proof) is None
    items, mp = tree.open_many([2, 3, 9])
    assert MerkleTree.get_multiproof_domain_len(mp) == 16
    assert MerkleTree.verify_many(h, tree.commitment(), [2, 3, 9], items, mp) is None


def test_limb_arithmetic_of_the_f64_ntt_passes_on_the_host(tmp_path):
    """winterfell_amd/csrc/l24.cuh (the carry-free 24-bit-limb DFTs of the f64 NTT passes, their bias vector, the
    multiply-accumulate exit and its fold) compiled with g++ and checked against big-integer DFTs over p — the planner's
    sign / rotation / known-zero bookkeeping and every carry path of the fold, without a GPU.  (On the GPU the same header
    is covered bit for bit by tests/test_gpu_fft.py; the device-only inline-assembly fold only exists there.)"""
    import subprocess
    exe = str(tmp_path / "l24_host_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "l24_host_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("ok")


def test_three_step_ntt_passes_emulated_on_the_host(tmp_path):
    """winterfell_amd/csrc/ntt_big.cuh (the two-pass f64 NTT plans: three-step passes of radix 2^10 / 2^11) compiled as plain C++
    — tests/cpp/stub/hip/hip_runtime.h stands in for the HIP runtime header — and run lane by lane, workgroup by workgroup: forward
    and inverse transforms, the coset-scaled inverse, and the batched row-major LDE with ragged column groups against a textbook
    radix-2 transform on canonical integers.  The same step functions are what the GPU kernel calls between its barriers
    (tests/test_gpu_ntt_two_pass.py is the device-side parity test)."""
    import shutil
    import subprocess
    clang = "/opt/rocm/lib/llvm/bin/clang++"          # __builtin_addc / __builtin_subc of the field headers are clang builtins
    if not os.path.exists(clang):
        clang = shutil.which("clang++")
    if not clang:
        pytest.skip("no clang++")
    exe = str(tmp_path / "ntt_big_host_test")
    subprocess.check_call([clang, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "tests", "cpp", "stub"),
                           os.path.join(ROOT, "tests", "cpp", "ntt_big_host_test.cpp"), "-o", exe])
    out = subprocess.run([exe, "21"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("all ok")


def test_f128_table_product_on_the_host(tmp_path):
    """winterfell_amd/csrc/f128.cuh compiled for the host: mul (two folds of the 256-bit product) and mul_tab (round 5: a table
    twiddle as the pair (w, w 2^64), a_lo w + a_hi (w 2^64), one fold) against an independent double-and-add modulo
    p = 2^128 - 45 * 2^40 + 1, 200 000 random products and the edge values (math/src/field/f128/mod.rs:429-466)."""
    import shutil
    import subprocess
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        clang = shutil.which("clang++")
    if not clang:
        pytest.skip("no clang++")
    exe = str(tmp_path / "f128_host_test")
    subprocess.check_call([clang, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "tests", "cpp", "stub"),
                           os.path.join(ROOT, "tests", "cpp", "f128_host_test.cpp"), "-o", exe])
    out = subprocess.run([exe, "200000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mul and mul_tab ok" in out.stdout


def test_transcript_encodings_on_the_reference_vectors():
    """The host mirror's TraceInfo::to_elements and ProofOptions::to_elements (winterfell_amd/prover/channel.py) on the reference's own
    test vectors: air/src/air/trace_info.rs:345-389 (main width 20, 64 rows; one auxiliary segment of width 9, 12 random elements, four
    metadata bytes) and air/src/options.rs:521-552 (no extension, folding 8, remainder degree 127, grinding 20, blowup 8, 30 queries)."""
    from winterfell_amd.prover.channel import ProofOptions, proof_options_to_elements, trace_info_to_elements
    assert trace_info_to_elements(20, 64, 8) == [int.from_bytes(bytes([0, 20, 0, 0]), "little"), 64]
    assert trace_info_to_elements(20, 64, 8, aux_width=9, num_aux_rands=12, meta=bytes([1, 2, 3, 4])) == \
        [int.from_bytes(bytes([12, 9, 1, 20]), "little"), 64, int.from_bytes(bytes([1, 2, 3, 4, 0, 0, 0, 0]), "little")]
    assert proof_options_to_elements(ProofOptions(30, 8, 20, 1, 8, 127)) == [int.from_bytes(bytes([8, 127, 8, 1]), "little"), 20, 30]

    # Context::to_elements (air/src/proof/context.rs:197-255) through the mirror's entry point, with a stand-in for the AIR's shape
    from winterfell_amd.math import fields
    from winterfell_amd.prover.channel import context_to_elements

    class Shape:
        FIELD, TRACE_WIDTH, AUX_TRACE_WIDTH, NUM_AUX_RANDS = fields.f64, 20, 9, 12
        trace_length = staticmethod(lambda: 4096)
        num_assertions = staticmethod(lambda: 100)
        num_transition_constraints = staticmethod(lambda: 28)

    assert context_to_elements(Shape, ProofOptions(30, 8, 20, 1, 8, 127)) == \
        [int.from_bytes(bytes([12, 9, 1, 20]), "little"), 4096, 1, 0xFFFFFFFF, 128, int.from_bytes(bytes([8, 127, 8, 1]), "little"), 20, 30]
