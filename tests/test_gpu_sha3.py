"""GPU parity for the Sha3_256 hasher (crypto/src/hash/sha/mod.rs) on the same kernels as Blake3_256: element hashing
for all three fields, merges, Merkle trees, (partitioned) trace commitments, FRI layers and proof-of-work, against the
CPU oracle (whose SHA3-256 is pinned to hashlib in tests/test_oracle_hash.py) and directly against hashlib."""
import hashlib

import numpy as np
import pytest

from conftest import P, rand_field

pytestmark = pytest.mark.gpu
HID = 2


@pytest.fixture(scope="module")
def wf():
    import winterfell_amd
    from winterfell_amd import crypto, fri, prover
    from winterfell_amd.math import fields
    return winterfell_amd.default_context(), crypto, prover, fields, fri


def test_hash_elements_all_fields_vs_hashlib(wf, oracle):
    ctx, crypto, prover, fields, fri = wf
    H = crypto.Sha3_256
    for n in (0, 1, 2, 16, 17, 18, 33, 34, 35, 51, 100, 255):      # 17 words = one 136-byte rate block
        ints = rand_field(n + 1, n)
        el = fields.from_ints(ints)
        want = hashlib.sha3_256(np.asarray(ints, dtype=np.uint64).tobytes()).digest()
        assert H.hash_elements(el).tobytes() == want, n
    # f128: IS_CANONICAL => raw element bytes; f62: canonical little-endian of as_int()
    v128 = [(i * 0x9E3779B97F4A7C15F39CC0605CEDC835 + 7) % fields.f128.M for i in range(23)]
    w = fields.f128.pack(v128)
    assert H.hash_elements(w, field=fields.f128).tobytes() == hashlib.sha3_256(w.tobytes()).digest()
    v62 = [(i * 0x2545F4914F6CDD1D + 3) % fields.f62.M for i in range(40)]
    w62 = fields.f62.pack([fields.f62.new(v) for v in v62])
    assert H.hash_elements(w62, field=fields.f62).tobytes() == hashlib.sha3_256(np.array(v62, dtype=np.uint64).tobytes()).digest()
    rows = fields.from_ints(rand_field(5, 7 * 19)).reshape(7, 19)
    got = H.hash_elements(rows)
    assert all(np.array_equal(got[i], oracle.hash_elements(HID, rows[i])) for i in range(7))


def test_merge_merkle_and_pow(wf, oracle):
    ctx, crypto, prover, fields, fri = wf
    H = crypto.Sha3_256
    rng = np.random.default_rng(3)
    pairs = rng.integers(0, 256, (9, 2, 32), dtype=np.uint8)
    got = H.merge(pairs)
    for i in range(9):
        assert got[i].tobytes() == hashlib.sha3_256(pairs[i].tobytes()).digest()
    for log_n in (1, 5, 11, 16):
        leaves = rng.integers(0, 256, (1 << log_n, 32), dtype=np.uint8)
        tree = crypto.MerkleTree.new(H, leaves)
        assert np.array_equal(tree.nodes, oracle.merkle_build(HID, leaves, par=True))
    idx = [3, 17, 40, 41]
    lv, proof = tree.prove_batch(idx)
    assert crypto.MerkleTree.verify_batch(H, tree.root(), idx, lv, proof) is None
    seed = rng.integers(0, 256, 32, dtype=np.uint8)
    d = H.merge_with_int(seed, (1 << 32) - 2, 5)
    for i in range(5):
        assert d[i].tobytes() == hashlib.sha3_256(seed.tobytes() + int((1 << 32) - 2 + i).to_bytes(8, "little")).digest()
    coin = oracle.RandomCoin(HID, [oracle.f64_new(9)])
    for factor in (0, 6, 12, 16):
        assert crypto.grind_query_seed(H, coin.seed(), factor) == coin.grind(factor)


@pytest.mark.parametrize("fname,c,log_n,blowup,parts,D", [("f64", 5, 8, 8, 1, 1), ("f64", 20, 6, 4, 4, 1), ("f64", 3, 10, 8, 1, 2),
                                                           ("f128", 4, 7, 8, 1, 1), ("f62", 6, 6, 8, 2, 1)])
def test_trace_commitment_vs_oracle(wf, oracle, fname, c, log_n, blowup, parts, D):
    ctx, crypto, prover, fields, fri = wf
    fld, ofld = {"f64": (fields.f64, oracle.f64t), "f128": (fields.f128, oracle.f128), "f62": (fields.f62, oracle.f62)}[fname]
    n = 1 << log_n
    rng = np.random.default_rng(c + log_n)
    vals = [int(a) * int(b) % fld.M for a, b in zip(rng.integers(1, 2**62, c * n * D), rng.integers(1, 2**62, c * n * D))]
    trace = fld.pack([fld.new(v) for v in vals]).reshape(c, -1)
    domain = prover.StarkDomain(n, blowup, field=fld)
    po = prover.PartitionOptions(parts, 1)
    lde, tree, polys = prover.build_trace_commitment(crypto.Sha3_256, prover.ColMatrix(trace, D, ctx, fld), domain, po)
    o_polys, o_lde, o_leaves, o_nodes = ofld.build_trace_commitment(HID, trace, blowup, int(domain.offset), D=D, num_partitions=parts, hash_rate=1)
    assert np.array_equal(ctx.to_host(lde.data), o_lde)
    assert np.array_equal(tree.leaves, o_leaves) and np.array_equal(tree.nodes, o_nodes)


def test_fri_build_layers_vs_oracle(wf, oracle):
    ctx, crypto, prover, fields, fri = wf
    D, log_len, N, blowup = 2, 12, 4, 8
    p = oracle.f64_from_int(rand_field(6, ((1 << log_len) // blowup) * D))
    ev = oracle.evaluate_poly_with_offset(p, oracle.f64_new(7), blowup, D=D, par=True)
    opts = fri.FriOptions(blowup, N, 7)
    chan, ochan = oracle.ProverChannel(HID, D), oracle.ProverChannel(HID, D)
    prover_ = fri.FriProver(opts, crypto.Sha3_256, ext_degree=D)
    prover_.build_layers(chan, ev.copy())
    cur = ev.copy()
    for k in range(prover_.num_layers()):
        tr = oracle.transpose_slice(cur, N, D)
        leaves, nodes = oracle.fri_layer_commit(HID, tr, N, D)
        ochan.commit_fri_layer(nodes[1])
        cur = oracle.apply_drp(tr, N, fields.new(7), ochan.draw_fri_alpha(), D)
        assert np.array_equal(prover_.layers[k].commitment.nodes, nodes)
    rem, com = oracle.fri_remainder(HID, cur, fields.new(7), blowup, D)
    assert np.array_equal(prover_.remainder_poly.reshape(-1), rem) and np.array_equal(chan.commitments[-1], com)
