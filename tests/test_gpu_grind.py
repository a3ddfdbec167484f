"""GPU parity: RandomCoin proof-of-work (merge_with_int over nonce ranges, grind_query_seed) and the query phase fed by
the coin's positions, against the CPU oracle's DefaultRandomCoin restatement."""
import numpy as np
import pytest

from conftest import P, rand_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wf():
    import winterfell_amd
    from winterfell_amd import crypto
    return winterfell_amd.default_context(), crypto


def _coin(oracle, hid, seed):
    return oracle.RandomCoin(hid, [oracle.f64_new(int(v)) for v in rand_field(seed, 4)])


@pytest.mark.parametrize("hname,hid", [("Blake3_256", 0), ("Rp64_256", 1)])
def test_merge_with_int_range_vs_oracle(wf, oracle, hname, hid):
    ctx, crypto = wf
    hasher = getattr(crypto, hname)
    seed = _coin(oracle, hid, 3).seed()
    # small counters, a range that crosses 2^32, and (Rp64_256: two-element path) values at / above the modulus
    for first, count in ((0, 1), (1, 300), ((1 << 32) - 7, 20), (P - 3, 8), ((1 << 64) - 5, 4)):
        got = hasher.merge_with_int(seed, first, count)
        for i in range(count):
            assert np.array_equal(got[i], oracle.merge_with_int(hid, seed, first + i)), (first, i)
    assert np.array_equal(hasher.merge_with_int(seed, 77), oracle.merge_with_int(hid, seed, 77))
    with pytest.raises(Exception):
        hasher.merge_with_int(seed, (1 << 64) - 2, 5)      # range wraps past u64::MAX


@pytest.mark.parametrize("hname,hid,factors", [("Blake3_256", 0, (0, 1, 8, 12, 16, 20)), ("Rp64_256", 1, (0, 5, 10, 14))])
def test_grind_query_seed_vs_oracle(wf, oracle, hname, hid, factors):
    """prover/src/channel.rs:169-175 serial path: the first (= smallest) nonce >= 1 that passes."""
    ctx, crypto = wf
    hasher = getattr(crypto, hname)
    for k, factor in enumerate(factors):
        coin = _coin(oracle, hid, 100 + k)
        want = coin.grind(factor)
        got = crypto.grind_query_seed(hasher, coin.seed(), factor)
        assert got == want and want >= 1
        assert crypto.check_leading_zeros(hasher, coin.seed(), got) == coin.check_leading_zeros(got) >= factor
        # every smaller nonce fails (minimality), spot-checked on the GPU's own batch evaluation
        if 1 < got <= 1 << 16:
            tz = crypto.check_leading_zeros(hasher, coin.seed(), 1, got - 1)
            assert (tz < factor).all()


def test_grind_range_and_errors(wf, oracle):
    ctx, crypto = wf
    coin = _coin(oracle, 0, 7)
    n1 = crypto.grind_query_seed(crypto.Blake3_256, coin.seed(), 10)
    # resuming after the first hit finds the next one; the oracle agrees on it being a hit and on the gap being empty
    n2 = crypto.grind_query_seed(crypto.Blake3_256, coin.seed(), 10, first_nonce=n1 + 1)
    assert n2 > n1 and coin.check_leading_zeros(n2) >= 10
    assert all(coin.check_leading_zeros(v) < 10 for v in range(n1 + 1, n2))
    with pytest.raises(RuntimeError, match="nonce not found"):
        crypto.grind_query_seed(crypto.Blake3_256, coin.seed(), 10, first_nonce=1, max_nonce=n1 - 1)
    assert crypto.grind_query_seed(crypto.Blake3_256, coin.seed(), 10, first_nonce=1, max_nonce=n1) == n1
    with pytest.raises(Exception):
        crypto.grind_query_seed(crypto.Blake3_256, coin.seed(), 65)


@pytest.mark.parametrize("hname,hid", [("Blake3_256", 0), ("Rp64_256", 1)])
def test_query_phase_from_device_resident_lde(wf, oracle, hname, hid):
    """prover/src/lib.rs:444-470: grind, draw the query positions, open the trace commitment at them — rows and
    Merkle paths come from the device-resident LDE / tree and verify against the committed root."""
    ctx, crypto = wf
    from winterfell_amd.prover import ColMatrix, StarkDomain, DefaultTraceLde
    hasher = getattr(crypto, hname)
    log_n, cols, blowup = 8, 5, 8
    trace = oracle.f64_from_int(rand_field(9, cols << log_n)).reshape(cols, 1 << log_n)
    domain = StarkDomain(1 << log_n, blowup)
    lde, polys = DefaultTraceLde.new(hasher, ColMatrix(ctx.to_device(trace)), domain)
    root = lde.main_segment_oracles.root()
    coin = oracle.RandomCoin(hid, [oracle.f64_new(5)])
    coin.reseed(root)
    nonce = crypto.grind_query_seed(hasher, coin.seed(), 8)
    assert nonce == coin.grind(8)
    positions = sorted(set(int(p) for p in coin.draw_integers(20, blowup << log_n, nonce)))
    (rows, (leaves, proof)), = lde.query(positions)
    want_lde = oracle.build_trace_commitment(hid, trace.copy(), blowup, oracle.f64_new(7))
    o_lde = want_lde[1].reshape(blowup << log_n, -1)
    for k, pos in enumerate(positions):
        assert np.array_equal(np.asarray(rows[k]).reshape(-1)[:cols], o_lde[pos][:cols])
    assert np.array_equal(np.stack(leaves), want_lde[2][positions])
    assert crypto.MerkleTree.verify_batch(hasher, root, positions, leaves, proof) is None
