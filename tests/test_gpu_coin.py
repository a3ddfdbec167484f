"""The device-resident DefaultRandomCoin (include/winterfell_hip.h: wf_coin_*) and the fused FRI layer loop built on it
(wf_fri_build_layers): value for value the transcript of the host coin / the oracle's restatement of
crypto/src/random/default.rs, for every hasher and every field + extension degree a digest can hold."""
import numpy as np
import pytest

from conftest import rand_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wf():
    import winterfell_amd
    from winterfell_amd import crypto, fri
    from winterfell_amd.math import fields
    return winterfell_amd.default_context(), crypto, fri, fields


def _script(coin_draw, coin_reseed, D, digests):
    """draw 5, reseed, draw 1, draw 3, reseed, draw 2 -> the drawn words, in order"""
    out = [coin_draw(D, 5)]
    coin_reseed(digests[0])
    out.append(coin_draw(D, 1))
    out.append(coin_draw(D, 3))
    coin_reseed(digests[1])
    out.append(coin_draw(D, 2))
    return np.concatenate([np.asarray(o, dtype=np.uint64).reshape(-1) for o in out])


@pytest.mark.parametrize("hname,fname,D", [("Blake3_256", "f64", 1), ("Blake3_256", "f64", 2), ("Blake3_256", "f64", 3), ("Blake3_256", "f128", 1),
                                           ("Blake3_256", "f128", 2), ("Rp64_256", "f64", 1), ("Rp64_256", "f64", 2), ("Rp64_256", "f64", 3)])
def test_device_coin_against_the_oracle_coin(wf, oracle, hname, fname, D):
    ctx, crypto, fri, fields = wf
    from oracle import prover as oprover
    hasher, f = getattr(crypto, hname), getattr(fields, fname)
    ofld = oracle.f64t if fname == "f64" else oracle.f128
    seed_words = f.pack([f.new(v) for v in (3, 1 << 40, 77)])
    ocoin = oprover.Coin(oprover.Hasher(oprover.BLAKE3_256 if hname == "Blake3_256" else oprover.RP64_256, ofld), seed_words)
    host = crypto.DefaultRandomCoin(hasher, f, seed_words, ctx)
    assert np.array_equal(host.seed, ocoin.seed)
    digests = [hasher.hash_elements(f.pack([f.new(k + 1)]), ctx, field=f) for k in range(2)]
    want = _script(lambda d, c: np.concatenate([ocoin.draw(d) for _ in range(c)]), ocoin.reseed, D, digests)
    dev = host.to_device()
    d_digests = [ctx.to_device(d) for d in digests]
    k = [0]

    def reseed(_):
        dev.reseed(d_digests[k[0]])
        k[0] += 1
    got = _script(lambda d, c: ctx.to_host(dev.draw(d, c)), reseed, D, digests)
    assert np.array_equal(got, want)
    seed, counter = dev.read()
    assert np.array_equal(seed, ocoin.seed) and counter == ocoin.counter
    # ... and the host coin carries on from there
    host.take_back(dev)
    assert np.array_equal(host.draw(D), ocoin.draw(D))


@pytest.mark.parametrize("hname,fname,D", [("Sha3_256", "f64", 2), ("Blake3_192", "f64", 3), ("Blake3_192", "f128", 1), ("RpJive64_256", "f64", 2),
                                           ("Rp62_248", "f62", 1), ("Rp62_248", "f62", 3), ("Sha3_256", "f128", 2)])
def test_device_coin_against_the_host_coin(wf, hname, fname, D):
    """the hashers the oracle's Python coin does not wrap: against the host coin, whose every hash is the library's batch entry
    point checked against the oracle elsewhere (test_gpu_sha3 / blake3_192 / rpjive / rp62)"""
    ctx, crypto, fri, fields = wf
    hasher, f = getattr(crypto, hname), getattr(fields, fname)
    seed_words = f.pack([f.new(v) for v in (5, 6, 7, 8, 9)])
    host = crypto.DefaultRandomCoin(hasher, f, seed_words, ctx)
    host.draw(1)                                             # hand over a coin that has already drawn (counter = 1)
    dev = host.to_device()
    digests = [hasher.hash_elements(f.pack([f.new(k + 11)]), ctx, field=f) for k in range(2)]
    want = _script(lambda d, c: np.concatenate([host.draw(d) for _ in range(c)]), host.reseed, D, digests)
    d_digests = [ctx.to_device(d) for d in digests]
    k = [0]

    def reseed(_):
        dev.reseed(d_digests[k[0]])
        k[0] += 1
    got = _script(lambda d, c: ctx.to_host(dev.draw(d, c)), reseed, D, digests)
    assert np.array_equal(got, want)
    seed, counter = dev.read()
    assert np.array_equal(seed, host.seed) and counter == host.counter


@pytest.mark.parametrize("hname,fname,D,count", [("Rp62_248", "f62", 1, 150), ("Rp62_248", "f62", 2, 40), ("Rp64_256", "f64", 2, 150),
                                                 ("RpJive64_256", "f64", 3, 70)])
def test_long_draws_walk_the_counters_in_the_reference_order(wf, hname, fname, D, count):
    """one wf_coin_draw of many elements on the Rescue family: the groups speculate over COUNTERS (windows of 64), the walker hands the
    values that decode to the draws in order — for f62, where three of four 8-byte values are rejected (M ~ 2^62), this is the
    reference's retry loop taken hundreds of times; the counter must end where the host coin's does"""
    ctx, crypto, fri, fields = wf
    hasher, f = getattr(crypto, hname), getattr(fields, fname)
    seed_words = f.pack([f.new(v) for v in (1, 2, 3)])
    host = crypto.DefaultRandomCoin(hasher, f, seed_words, ctx)
    dev = host.to_device()
    got = ctx.to_host(dev.draw(D, count)).reshape(-1)
    want = np.concatenate([host.draw(D) for _ in range(count)])
    assert np.array_equal(got, want)
    seed, counter = dev.read()
    assert counter == host.counter and np.array_equal(seed, host.seed)
    if fname == "f62":
        assert counter > 2 * count                       # the retries really happened
    # a second request continues from the counter the walker left
    assert np.array_equal(ctx.to_host(dev.draw(D, 3)).reshape(-1), np.concatenate([host.draw(D) for _ in range(3)]))


def test_reseed_copies_the_digest_and_rejects_bad_arguments(wf):
    ctx, crypto, fri, fields = wf
    import ctypes
    from winterfell_amd._lib import ptr
    dev = crypto.DefaultRandomCoin(crypto.Blake3_256, fields.f64, np.zeros(0, dtype=np.uint64), ctx).to_device()
    dig = ctx.to_device(np.arange(32, dtype=np.uint8))
    copy = ctx.empty_u8(32)
    dev.reseed(dig, copy)
    assert np.array_equal(ctx.to_host(copy), np.arange(32, dtype=np.uint8))
    lib, out = ctx.lib, ctx.empty_u64(8)
    # wf_coin_init / wf_coin_read (what a non-Python host uses instead of writing the 64 state bytes itself)
    st2, seed, counter = ctx.empty_u8(64), np.empty(32, dtype=np.uint8), ctypes.c_uint64(99)
    ctx.call("wf_coin_init", ptr(st2), np.arange(32, dtype=np.uint8).ctypes.data_as(ctypes.c_void_p))
    ctx.call("wf_coin_draw", 0, 0, 2, ptr(st2), 3, ptr(out))
    ctx.call("wf_coin_read", ptr(st2), seed.ctypes.data_as(ctypes.c_void_p), ctypes.byref(counter))
    host = crypto.DefaultRandomCoin(crypto.Blake3_256, fields.f64, np.zeros(0, dtype=np.uint64), ctx)
    host.seed, host.counter = np.arange(32, dtype=np.uint8), 0
    want = np.concatenate([host.draw(2) for _ in range(3)])
    assert np.array_equal(ctx.to_host(out)[:6], want) and np.array_equal(seed, host.seed) and counter.value == host.counter
    # commit + draw in one launch
    ctx.call("wf_coin_reseed_draw", 0, 0, 2, ptr(st2), ptr(dig), None, ptr(out))
    host.reseed(np.arange(32, dtype=np.uint8))
    assert np.array_equal(ctx.to_host(out)[:2], host.draw(2))
    assert lib.wf_coin_draw(ctx.handle, 0, 1, 3, ptr(dev.state), 1, ptr(out)) != 0          # f128 has no cubic extension
    assert lib.wf_coin_draw(ctx.handle, 0, 7, 1, ptr(dev.state), 1, ptr(out)) != 0          # unknown field
    assert lib.wf_coin_draw(ctx.handle, 99, 0, 1, ptr(dev.state), 1, ptr(out)) != 0         # unknown hasher
    assert lib.wf_coin_reseed(ctx.handle, 0, None, ptr(dig), None) != 0
    assert lib.wf_coin_init(ctx.handle, ptr(dev.state), None) != 0


@pytest.mark.parametrize("hname,fname,D,log_len,N,rem_deg", [("Blake3_256", "f64", 2, 16, 4, 31), ("Blake3_256", "f64", 1, 12, 2, 7),
                                                             ("Sha3_256", "f64", 3, 13, 8, 31), ("Blake3_256", "f128", 2, 12, 4, 15),
                                                             ("Blake3_192", "f64", 2, 14, 16, 31), ("Blake3_256", "f64", 2, 8, 4, 31),
                                                             # shapes around the one-launch tail (fri_tail_kernel) and the one-launch trees:
                                                             ("Blake3_256", "f64", 1, 14, 4, 127),    # remainder = 1024 bytes: the longest one-chunk hash
                                                             ("Blake3_256", "f64", 2, 13, 2, 63),     # two layers + a 64 x 2 remainder in the tail
                                                             ("Blake3_256", "f64", 3, 12, 2, 63),     # 1536-byte remainder: outside the tail
                                                             ("Blake3_192", "f64", 2, 12, 4, 31),     # the tail's one-lane coin and remainder paths
                                                             ("Blake3_256", "f64", 1, 13, 16, 7),     # folding 16: tail layers of 512 and 32 rows
                                                             ("Blake3_256", "f64", 2, 11, 2, 0),      # everything in the tail, remainder of ONE coefficient
                                                             ("Blake3_256", "f64", 2, 20, 4, 31),     # trees of 2^18 .. 2^12 leaves (ticket), then the tail
                                                             ("Blake3_256", "f64", 1, 21, 2, 15),     # 2^20 / 2^19-leaf trees: 4096 inputs per workgroup
                                                             ("Blake3_256", "f128", 1, 18, 4, 31)])   # f128 coin inside the one-launch tree
def test_fused_layer_loop_equals_the_layer_by_layer_prover(wf, hname, fname, D, log_len, N, rem_deg):
    """FriProver.build_layers with the coin on the device (one wf_fri_build_layers call) against the same prover driven layer
    by layer through a host coin (the path test_gpu_fri.py checks against the oracle): commitments, alphas, every layer's
    evaluations and tree, the remainder, the coin afterwards."""
    ctx, crypto, fri, fields = wf
    hasher, f = getattr(crypto, hname), getattr(fields, fname)
    n = 1 << log_len
    rng = np.random.default_rng(4000 + log_len + N)
    words = rng.integers(0, 1 << 63, (n * D, 2), dtype=np.uint64)
    ev = f.from_ints([(int(a) | (int(b) << 64)) % f.M for a, b in words])      # any vector folds; FRI's low-degree input is not needed here
    opts = fri.FriOptions(8, N, rem_deg, field=f)
    runs = []
    for device_coin in (True, False):
        chan = fri.DefaultProverChannel(n, 8, hasher, ext_degree=D, field=f, ctx=ctx, device_coin=device_coin)
        assert (chan.fri_device_coin() is not None) == device_coin
        pr = fri.FriProver(opts, hasher, ext_degree=D, ctx=ctx)
        pr.build_layers(chan, ev.copy())
        runs.append((chan, pr))
    (ca, pa), (cb, pb) = runs
    assert pa.num_layers() == pb.num_layers() == opts.num_fri_layers(n)
    assert len(ca.commitments) == len(cb.commitments) == pa.num_layers() + 1
    for x, y in zip(ca.commitments, cb.commitments):
        assert np.array_equal(x, y)
    assert len(ca.alphas) == len(cb.alphas) and all(np.array_equal(x, y) for x, y in zip(ca.alphas, cb.alphas))
    for la, lb in zip(pa.layers, pb.layers):
        assert np.array_equal(ctx.to_host(la.evaluations), ctx.to_host(lb.evaluations))
        assert np.array_equal(la.commitment.nodes, lb.commitment.nodes)
    assert np.array_equal(pa.remainder_poly, pb.remainder_poly)
    assert np.array_equal(ca.public_coin.seed, cb.public_coin.seed) and ca.public_coin.counter == cb.public_coin.counter
    assert ca.draw_query_positions(3) == cb.draw_query_positions(3)


def test_fused_layer_loop_with_a_rescue_hasher(wf, monkeypatch):
    """the library call works for every hasher; the Python prover only skips it for the Rescue family because a single-lane
    permutation is slower than the round trip it saves (crypto/hash.py DEVICE_COIN) — forced on here"""
    ctx, crypto, fri, fields = wf
    f = fields.f64
    monkeypatch.setattr(crypto.Rp64_256, "DEVICE_COIN", True)
    n, D, N = 1 << 10, 2, 4
    rng = np.random.default_rng(77)
    ev = f.from_ints([int(v) % f.M for v in rng.integers(0, 1 << 63, n * D, dtype=np.uint64)])
    opts = fri.FriOptions(8, N, 7, field=f)
    runs = []
    for device_coin in (True, False):
        chan = fri.DefaultProverChannel(n, 8, crypto.Rp64_256, ext_degree=D, field=f, ctx=ctx, device_coin=device_coin)
        pr = fri.FriProver(opts, crypto.Rp64_256, ext_degree=D, ctx=ctx)
        pr.build_layers(chan, ev.copy())
        runs.append((chan, pr))
    (ca, pa), (cb, pb) = runs
    assert ca.fri_device_coin() is not None and cb.fri_device_coin() is None
    assert len(ca.commitments) == len(cb.commitments) and all(np.array_equal(x, y) for x, y in zip(ca.commitments, cb.commitments))
    assert all(np.array_equal(x, y) for x, y in zip(ca.alphas, cb.alphas)) and np.array_equal(pa.remainder_poly, pb.remainder_poly)
    assert np.array_equal(ca.public_coin.seed, cb.public_coin.seed)


@pytest.mark.parametrize("D,N,log_len", [(2, 4, 14), (1, 2, 10), (3, 8, 13), (1, 16, 12)])
def test_build_layers_hands_out_every_folded_vector(wf, oracle, D, N, log_len):
    """the C entry point's d_folded[k] (the next layer's evaluations in natural order, an output of the call even where the
    fold is fused with the next layer's commit) against the oracle's apply_drp, layer by layer with the alphas the call drew;
    the remainder against set_remainder"""
    import ctypes
    from winterfell_amd._lib import ptr
    ctx, crypto, fri, fields = wf
    f, hasher = fields.f64, crypto.Blake3_256
    n, blowup, rem_deg = 1 << log_len, 8, 7
    ev = oracle.f64_from_int(rand_field(600 + log_len + N, n * D))
    opts = fri.FriOptions(blowup, N, rem_deg, field=f)
    nl = opts.num_fri_layers(n)
    rows, tr, lv, nd, fo = n, [], [], [], []
    for _ in range(nl):
        rows //= N
        tr.append(ctx.empty_u64(rows, N * D)); lv.append(ctx.empty_u8(rows, 32)); nd.append(ctx.empty_u8(rows, 32)); fo.append(ctx.empty_u64(rows * D))
    rem_size = rows // blowup
    roots, alphas, rem = ctx.empty_u8(nl + 1, 32), ctx.empty_u64(nl, D), ctx.empty_u64(rem_size, D)
    coin = crypto.DefaultRandomCoin(hasher, f, np.zeros(0, dtype=np.uint64), ctx).to_device()
    coin.draw(1)                                                   # uploads the state
    off = f.element_words(f.new(7))
    arr = lambda ts: (ctypes.c_void_p * nl)(*[t.data_ptr() for t in ts])
    d_ev = ctx.to_device(ev)
    ctx.call("wf_fri_build_layers", hasher.HASH_ID, f.ID, D, ptr(d_ev), log_len, N, nl, off.ctypes.data_as(ctypes.c_void_p), ptr(coin.state), arr(tr),
             arr(lv), arr(nd), arr(fo), ptr(roots), ptr(alphas), blowup, ptr(rem))
    h_alphas = ctx.to_host(alphas)
    cur = ev.copy()
    for k in range(nl):
        t = oracle.transpose_slice(cur, N, D)
        assert np.array_equal(ctx.to_host(tr[k]).reshape(-1), t), "layer %d evaluations" % k
        _, nodes = oracle.fri_layer_commit(0, t, N, D)
        assert np.array_equal(ctx.to_host(nd[k]), nodes) and np.array_equal(ctx.to_host(roots)[k], nodes[1])
        cur = oracle.apply_drp(t, N, fields.new(7), h_alphas[k], D)
        if k + 1 < nl:                                             # the last one is interpolated in place by the remainder step
            assert np.array_equal(ctx.to_host(fo[k]), cur), "folded vector %d" % k
    o_rem, o_com = oracle.fri_remainder(0, cur, fields.new(7), blowup, D)
    assert np.array_equal(ctx.to_host(rem).reshape(-1), np.asarray(o_rem).reshape(-1)) and np.array_equal(ctx.to_host(roots)[nl], o_com)


def test_remainder_only_call_leaves_the_callers_evaluations_alone(wf, oracle):
    """round-2 advice: wf_fri_build_layers with no layers interpolated the caller's `const void *d_evals` in place.  Now it works on
    a private copy: the evaluations are unchanged, the remainder and its commitment are the oracle's."""
    import ctypes
    from winterfell_amd._lib import ptr
    ctx, crypto, fri, fields = wf
    f, hasher, D, blowup = fields.f64, crypto.Blake3_256, 2, 8
    n = 64
    ev = oracle.f64_from_int(rand_field(4242, n * D))
    d_ev = ctx.to_device(ev)
    roots, alphas, rem = ctx.empty_u8(1, 32), ctx.empty_u64(1, D), ctx.empty_u64(n // blowup, D)
    coin = crypto.DefaultRandomCoin(hasher, f, np.zeros(0, dtype=np.uint64), ctx).to_device()
    coin.draw(1)
    off = f.element_words(f.new(7))
    null = (ctypes.c_void_p * 1)(None)
    ctx.call("wf_fri_build_layers", hasher.HASH_ID, f.ID, D, ptr(d_ev), 6, 4, 0, off.ctypes.data_as(ctypes.c_void_p), ptr(coin.state), null, null, null, null,
             ptr(roots), ptr(alphas), blowup, ptr(rem))
    assert np.array_equal(ctx.to_host(d_ev), ev), "d_evals was overwritten"
    o_rem, o_com = oracle.fri_remainder(0, ev, fields.new(7), blowup, D)
    assert np.array_equal(ctx.to_host(rem).reshape(-1), np.asarray(o_rem).reshape(-1)) and np.array_equal(ctx.to_host(roots)[0], o_com)


def test_fused_loop_refuses_a_one_row_layer_before_touching_the_coin(wf):
    """round-2 advice: folding 4 over a domain of 4 points with blowup 2 gives a last layer of ONE row (no Merkle tree).  The
    Python prover must say so before the coin moves to the device; the channel's coin stays usable."""
    ctx, crypto, fri, fields = wf
    f = fields.f64
    opts = fri.FriOptions(2, 4, 0, field=f)
    assert opts.num_fri_layers(8) == 1 and opts.num_fri_layers(4) == 1
    chan = fri.DefaultProverChannel(8, 1, crypto.Blake3_256, ext_degree=1, field=f, ctx=ctx, device_coin=True)
    seed_before = np.array(chan.public_coin.seed, copy=True)
    pr = fri.FriProver(opts, crypto.Blake3_256, ext_degree=1, ctx=ctx)
    with pytest.raises(ValueError, match="at least two leaves"):
        pr.build_layers(chan, f.from_ints([1, 2, 3, 4]))
    assert np.array_equal(chan.public_coin.seed, seed_before) and pr.num_layers() == 0
    assert chan.draw_query_positions(1) is not None                 # the coin still answers
