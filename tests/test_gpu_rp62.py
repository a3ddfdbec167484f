"""GPU parity for Rp62_248 (crypto/src/hash/rescue/rp62_248, f62 only) against the CPU oracle (pinned to the reference's
permutation known-answer test): sponge, merge, merge_with_int, Merkle, trace commitment over f62, proof-of-work."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HID = 4


@pytest.fixture(scope="module")
def wf():
    import winterfell_amd
    from winterfell_amd import crypto, prover
    from winterfell_amd.math import fields
    return winterfell_amd.default_context(), crypto, prover, fields


def _rand62(fields, n, seed):
    rng = np.random.default_rng(seed)
    return fields.f62.pack([fields.f62.new(int(v)) for v in rng.integers(0, fields.f62.M, size=n, dtype=np.uint64)])


def _o_hash(oracle, el):
    out = np.empty(32, dtype=np.uint8)
    e = np.ascontiguousarray(el, dtype=np.uint64)
    buf = e if e.size else np.zeros(1, dtype=np.uint64)
    oracle.lib().or_rp62_hash_elements(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(e.size), out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_sponge_merge_int_pow(wf, oracle):
    ctx, crypto, prover, fields = wf
    H = crypto.Rp62_248
    for n in (0, 1, 7, 8, 9, 16, 17, 40):
        el = _rand62(fields, n, n + 1)
        assert np.array_equal(H.hash_elements(el), _o_hash(oracle, el)), n
    lazy = _rand62(fields, 5, 77) + np.uint64(fields.f62.M)            # the reference's lazy [0, 2M) words are accepted
    assert np.array_equal(H.hash_elements(lazy), _o_hash(oracle, lazy))
    pairs = _rand62(fields, 6 * 8, 3).view(np.uint8).reshape(6, 2, 32)
    got = H.merge(pairs)
    assert all(np.array_equal(got[i], oracle.merge(HID, pairs[i])) for i in range(6))
    seed = pairs[1][0]
    for first, count in ((0, 3), (fields.f62.M - 2, 5), ((1 << 64) - 2, 2)):
        d = H.merge_with_int(seed, first, count)
        assert all(np.array_equal(d[i], oracle.merge_with_int(HID, seed, first + i)) for i in range(count))
    # digest bytes / proof of work: check_leading_zeros reads the packed 62-bit representation
    ob = np.empty(32, dtype=np.uint8)
    oracle.lib().or_rp62_digest_as_bytes(seed.ctypes.data_as(ctypes.c_void_p), ob.ctypes.data_as(ctypes.c_void_p))
    assert H.digest_as_bytes(seed) == ob.tobytes()
    for factor in (0, 5, 10):
        nonce = crypto.grind_query_seed(H, seed, factor)
        tz = crypto.check_leading_zeros(H, seed, 1, nonce)
        assert tz[-1] >= factor and (tz[:-1] < factor).all()
        d = oracle.merge_with_int(HID, seed, nonce)
        oracle.lib().or_rp62_digest_as_bytes(d.ctypes.data_as(ctypes.c_void_p), ob.ctypes.data_as(ctypes.c_void_p))
        head = int.from_bytes(ob.tobytes()[:8], "little")
        assert head % (1 << factor) == 0
    from winterfell_amd._lib import WfError
    with pytest.raises(WfError):
        H.hash_elements(np.arange(4, dtype=np.uint64), field=fields.f64)


def test_merkle_and_trace_commitment(wf, oracle):
    ctx, crypto, prover, fields = wf
    H, fld, ofld = crypto.Rp62_248, fields.f62, oracle.f62
    for log_n in (1, 3, 8, 11):
        leaves = _rand62(fields, 4 << log_n, log_n).view(np.uint8).reshape(-1, 32)
        tree = crypto.MerkleTree.new(H, leaves)
        assert np.array_equal(tree.nodes, oracle.merkle_build(HID, leaves, par=True))
    for c, log_n, blowup, parts, D in ((5, 6, 8, 1, 1), (12, 5, 4, 3, 1), (3, 7, 8, 1, 2)):
        n = 1 << log_n
        trace = _rand62(fields, c * n * D, c).reshape(c, n * D)
        domain = prover.StarkDomain(n, blowup, field=fld)
        lde, tree, polys = prover.build_trace_commitment(H, prover.ColMatrix(trace, D, ctx, fld), domain, prover.PartitionOptions(parts, 1))
        o = ofld.build_trace_commitment(HID, trace, blowup, int(domain.offset), D=D, num_partitions=parts, hash_rate=1)
        assert np.array_equal(ctx.to_host(lde.data), o[1])
        assert np.array_equal(tree.leaves, o[2]) and np.array_equal(tree.nodes, o[3])
