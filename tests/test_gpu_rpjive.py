"""GPU parity for RpJive64_256 (crypto/src/hash/rescue/rp64_256_jive) against the CPU oracle (pinned to the reference's
permutation known-answer test): sponge, Jive merge, merge_with_int, Merkle, (partitioned) trace commitment, FRI, PoW."""
import numpy as np
import pytest

from conftest import P, rand_field

pytestmark = pytest.mark.gpu
HID = 3


@pytest.fixture(scope="module")
def wf():
    import winterfell_amd
    from winterfell_amd import crypto, fri, prover
    from winterfell_amd.math import fields
    return winterfell_amd.default_context(), crypto, prover, fields, fri


def test_sponge_merge_and_int(wf, oracle):
    ctx, crypto, prover, fields, fri = wf
    H = crypto.RpJive64_256
    for n in (0, 1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 31, 64, 65):
        el = fields.from_ints(rand_field(n + 3, n))
        assert np.array_equal(H.hash_elements(el), oracle.hash_elements(HID, el)), n
    rows = fields.from_ints(rand_field(9, 6 * 11)).reshape(6, 11)
    got = H.hash_elements(rows)
    assert all(np.array_equal(got[i], oracle.hash_elements(HID, rows[i])) for i in range(6))
    pairs = fields.from_ints(rand_field(4, 5 * 8)).view(np.uint8).reshape(5, 2, 32)
    got = H.merge(pairs)
    assert all(np.array_equal(got[i], oracle.merge(HID, pairs[i])) for i in range(5))
    seed = pairs[0][0]
    for first, count in ((0, 3), ((1 << 32) - 2, 4), (P - 2, 5), ((1 << 64) - 3, 2)):
        d = H.merge_with_int(seed, first, count)
        assert all(np.array_equal(d[i], oracle.merge_with_int(HID, seed, first + i)) for i in range(count))
    coin = oracle.RandomCoin(HID, [oracle.f64_new(3)])
    for factor in (0, 4, 9, 12):
        assert crypto.grind_query_seed(H, coin.seed(), factor) == coin.grind(factor)
    from winterfell_amd._lib import WfError
    with pytest.raises(WfError):
        H.hash_elements(fields.f128.pack([1, 2]), field=fields.f128)      # Rescue hashers are defined over f64 only


def test_merkle_and_commitments(wf, oracle):
    ctx, crypto, prover, fields, fri = wf
    H = crypto.RpJive64_256
    for log_n in (1, 4, 9, 12):
        leaves = fields.from_ints(rand_field(log_n, 4 << log_n)).view(np.uint8).reshape(-1, 32)
        tree = crypto.MerkleTree.new(H, leaves)
        assert np.array_equal(tree.nodes, oracle.merkle_build(HID, leaves, par=True))
    idx = [0, 5, 6, 4000]
    lv, proof = tree.prove_batch(idx)
    assert crypto.MerkleTree.verify_batch(H, tree.root(), idx, lv, proof) is None
    for c, log_n, blowup, parts, D in ((5, 7, 8, 1, 1), (12, 6, 4, 4, 1), (3, 8, 8, 2, 2)):
        n = 1 << log_n
        trace = fields.from_ints(rand_field(c + log_n, c * n * D)).reshape(c, n * D)
        domain = prover.StarkDomain(n, blowup)
        lde, tree, polys = prover.build_trace_commitment(H, prover.ColMatrix(trace, D, ctx), domain, prover.PartitionOptions(parts, 4))
        o = oracle.build_trace_commitment(HID, trace, blowup, fields.new(7), D=D, num_partitions=parts, hash_rate=4)
        assert np.array_equal(tree.leaves, o[2]) and np.array_equal(tree.nodes, o[3])


def test_fri_build_layers_vs_oracle(wf, oracle):
    ctx, crypto, prover, fields, fri = wf
    D, log_len, N, blowup = 2, 11, 4, 8
    p = oracle.f64_from_int(rand_field(16, ((1 << log_len) // blowup) * D))
    ev = oracle.evaluate_poly_with_offset(p, oracle.f64_new(7), blowup, D=D, par=True)
    opts = fri.FriOptions(blowup, N, 7)
    chan, ochan = oracle.ProverChannel(HID, D), oracle.ProverChannel(HID, D)
    pr = fri.FriProver(opts, crypto.RpJive64_256, ext_degree=D)
    pr.build_layers(chan, ev.copy())
    cur = ev.copy()
    for k in range(pr.num_layers()):
        tr = oracle.transpose_slice(cur, N, D)
        leaves, nodes = oracle.fri_layer_commit(HID, tr, N, D)
        ochan.commit_fri_layer(nodes[1])
        cur = oracle.apply_drp(tr, N, fields.new(7), ochan.draw_fri_alpha(), D)
        assert np.array_equal(pr.layers[k].commitment.nodes, nodes)
    rem, com = oracle.fri_remainder(HID, cur, fields.new(7), blowup, D)
    assert np.array_equal(pr.remainder_poly.reshape(-1), rem) and np.array_equal(chan.commitments[-1], com)
