"""The multi-device entry points of the C ABI (include/winterfell_hip.h: wf_comm_*): the column / partition sharded trace
commitment with its two exchange steps, driven rank for rank through the library itself.

Only one GPU is reachable here, so (a) the G-rank path runs over the LOOPBACK transport — G contexts on the one device, one
thread per rank, the collectives as peer copies around a thread barrier — and is compared node for node with the
single-device commitment under PartitionOptions::new(G, .) (air/src/options.rs:391-451, row_matrix.rs:204-223) and with the
CPU oracle; (b) the RCCL transport is executed for real with a communicator of one rank (ncclCommInitRank, ncclAllGather,
ncclAllToAll on the library's own dlopen()ed librccl)."""
import ctypes
import threading

import numpy as np
import pytest

from conftest import rand_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wf():
    import winterfell_amd
    from winterfell_amd import crypto, prover
    from winterfell_amd.math import fields
    return winterfell_amd.default_context(), crypto, prover, fields


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


def _sharded_commit(ctx0, comms, ctxs, hasher, fld, shards, log_n, log_b, offset_words, D=1):
    """one thread per rank -> list of dict(polys, lde, leaves, nodes, top, root) as numpy"""
    import torch
    G, n, N = len(comms), 1 << log_n, 1 << (log_n + log_b)
    per = N // G
    out, errs = [None] * G, []
    bufs = []
    for r in range(G):                       # allocate on the main thread (torch), compute on the rank threads (library)
        c = shards[r].shape[0]
        rw = int(ctx0.lib.wf_row_width(c, D))
        bufs.append(dict(tr=ctx0.to_device(shards[r]), lde=ctx0.empty_u64(N, rw * fld.W), leaves=ctx0.empty_u8(per, 32),
                         nodes=ctx0.empty_u8(per, 32), top=ctx0.empty_u8(G, 32), root=np.zeros(32, dtype=np.uint8), c=c))
    torch.cuda.synchronize()

    def run(r):
        try:
            lib, b = ctx0.lib, bufs[r]
            st = lib.wf_comm_sharded_commit(comms[r], hasher.HASH_ID, fld.ID, D, _vp(b["tr"]), b["c"], n * D, log_n, log_b,   # col_stride: base elements
                                            offset_words.ctypes.data_as(ctypes.c_void_p), 0, _vp(b["lde"]), _vp(b["leaves"]), _vp(b["nodes"]),
                                            _vp(b["top"]), b["root"].ctypes.data_as(ctypes.c_void_p))
            assert st == 0, "wf_comm_sharded_commit -> %d" % st
            assert lib.wf_ctx_sync(ctxs[r]) == 0
        except BaseException as e:        # noqa: BLE001 - reported by the main thread
            errs.append(e)

    ts = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    for r in range(G):
        b = bufs[r]
        out[r] = dict(polys=ctx0.to_host(b["tr"]), lde=ctx0.to_host(b["lde"]), leaves=ctx0.to_host(b["leaves"]), nodes=ctx0.to_host(b["nodes"]),
                      top=ctx0.to_host(b["top"]), root=b["root"])
    return out


def _make_loopback(ctx0, G):
    lib = ctx0.lib
    handles = []
    for _ in range(G):
        h = ctypes.c_void_p()
        assert lib.wf_ctx_create(ctx0.device.index or 0, ctypes.byref(h)) == 0
        handles.append(h)
    arr = (ctypes.c_void_p * G)(*[h.value for h in handles])
    comms = (ctypes.c_void_p * G)()
    assert lib.wf_comm_init_loopback(arr, G, comms) == 0
    return handles, [ctypes.c_void_p(c) for c in comms]


def _teardown(ctx0, handles, comms):
    for c in comms:
        ctx0.lib.wf_comm_destroy(c)
    for h in handles:
        ctx0.lib.wf_ctx_destroy(h)


@pytest.mark.parametrize("hname,G,cols_per_shard,log_n,log_b", [("Blake3_256", 2, 3, 8, 3), ("Blake3_256", 4, 2, 7, 3), ("Rp64_256", 2, 2, 6, 2),
                                                                 ("Blake3_256", 8, 1, 10, 3), ("Blake3_256", 4, 9, 6, 1)])
def test_sharded_commit_over_loopback_equals_the_partitioned_commitment(wf, oracle, hname, G, cols_per_shard, log_n, log_b):
    ctx, crypto, prover, fields = wf
    f = fields.f64
    hasher = getattr(crypto, hname)
    n, c = 1 << log_n, G * cols_per_shard
    trace = oracle.f64_from_int(rand_field(900 + G * 10 + c, n * c)).reshape(c, n)
    shards = [trace[r * cols_per_shard:(r + 1) * cols_per_shard] for r in range(G)]
    handles, comms = _make_loopback(ctx, G)
    try:
        assert [ctx.lib.wf_comm_rank(cm) for cm in comms] == list(range(G)) and ctx.lib.wf_comm_size(comms[0]) == G
        res = _sharded_commit(ctx, comms, handles, hasher, f, shards, log_n, log_b, f.element_words(f.new(7)))
    finally:
        _teardown(ctx, handles, comms)
    # the single-device commitment under PartitionOptions(G, 1): partition size = ceil(c / G) = cols_per_shard
    hid = {"Blake3_256": 0, "Rp64_256": 1}[hname]
    o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(hid, trace, 1 << log_b, f.new(7), num_partitions=G, hash_rate=1)
    N = n << log_b
    per = N // G
    for r in range(G):
        assert np.array_equal(res[r]["polys"], o_polys[r * cols_per_shard:(r + 1) * cols_per_shard]), "polys of rank %d" % r
        assert np.array_equal(res[r]["lde"][:, :cols_per_shard], o_lde[:, r * cols_per_shard:(r + 1) * cols_per_shard]), "LDE of rank %d" % r
        assert np.array_equal(res[r]["leaves"], o_leaves[r * per:(r + 1) * per]), "leaves of rank %d" % r
        assert np.array_equal(res[r]["root"], o_nodes[1]) and np.array_equal(res[r]["top"][1:], o_nodes[1:G]), "top tree on rank %d" % r
        # the rank's subtree: local heap index j sits at global index ((G + r) << depth) + offset
        for j in range(1, per):
            depth = j.bit_length() - 1
            assert np.array_equal(res[r]["nodes"][j], o_nodes[((G + r) << depth) + (j - (1 << depth))]), (r, j)
    # ... which is also what the library's single-device entry point produces
    lde, tree, _ = prover.build_trace_commitment(hasher, prover.ColMatrix(trace), prover.StarkDomain(n, 1 << log_b), prover.PartitionOptions(G, 1))
    assert np.array_equal(tree.nodes, o_nodes)


def test_collectives_over_loopback(wf):
    ctx, crypto, prover, fields = wf
    import torch
    G, nbytes = 4, 96
    handles, comms = _make_loopback(ctx, G)
    try:
        send = [ctx.to_device(np.full(G * nbytes, 10 * r, dtype=np.uint8) + np.repeat(np.arange(G, dtype=np.uint8), nbytes)) for r in range(G)]
        recv_a2a = [ctx.empty_u8(G * nbytes) for _ in range(G)]
        recv_ag = [ctx.empty_u8(G * nbytes) for _ in range(G)]
        torch.cuda.synchronize()
        errs = []

        def run(r):
            try:
                assert ctx.lib.wf_comm_all_to_all(comms[r], _vp(send[r]), _vp(recv_a2a[r]), nbytes) == 0
                assert ctx.lib.wf_comm_all_gather(comms[r], _vp(send[r]), _vp(recv_ag[r]), nbytes) == 0
                assert ctx.lib.wf_ctx_sync(handles[r]) == 0
            except BaseException as e:    # noqa: BLE001
                errs.append(e)
        ts = [threading.Thread(target=run, args=(r,)) for r in range(G)]
        [t.start() for t in ts]
        [t.join(timeout=120) for t in ts]
        assert not errs, errs
        for r in range(G):
            a2a, ag = ctx.to_host(recv_a2a[r]).reshape(G, nbytes), ctx.to_host(recv_ag[r]).reshape(G, nbytes)
            for k in range(G):
                assert (a2a[k] == 10 * k + r).all()          # block r of rank k's send buffer
                assert (ag[k] == 10 * k).all()               # block 0 of rank k's send buffer
    finally:
        _teardown(ctx, handles, comms)


def test_rccl_transport_with_one_rank(wf, oracle):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclAllToAll actually execute (a communicator of ONE rank is all a
    one-GPU box allows), and the sharded entry point over it equals the plain commitment (G = 1: the leaf is the row hash)."""
    ctx, crypto, prover, fields = wf
    import torch
    f = fields.f64
    uid = (ctypes.c_uint8 * 128)()
    st = ctx.lib.wf_comm_get_unique_id(uid)
    assert st == 0, "wf_comm_get_unique_id -> %d (librccl not loadable?)" % st
    cm = ctypes.c_void_p()
    assert ctx.lib.wf_comm_init_rank(ctx.handle, uid, 0, 1, ctypes.byref(cm)) == 0
    try:
        a = ctx.to_device(np.arange(256, dtype=np.uint8))
        b, c = ctx.empty_u8(256), ctx.empty_u8(256)
        torch.cuda.synchronize()
        assert ctx.lib.wf_comm_all_gather(cm, _vp(a), _vp(b), 256) == 0 and ctx.lib.wf_comm_all_to_all(cm, _vp(a), _vp(c), 256) == 0
        ctx.sync()
        assert np.array_equal(ctx.to_host(b), np.arange(256, dtype=np.uint8)) and np.array_equal(ctx.to_host(c), np.arange(256, dtype=np.uint8))
        n, cols, log_n, log_b = 512, 5, 9, 3
        trace = oracle.f64_from_int(rand_field(4711, n * cols)).reshape(cols, n)
        res = _sharded_commit(ctx, [cm], [ctx.handle], crypto.Blake3_256, f, [trace], log_n, log_b, f.element_words(f.new(7)))[0]
        o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(0, trace, 8, f.new(7))
        assert np.array_equal(res["polys"], o_polys) and np.array_equal(res["lde"], o_lde)
        assert np.array_equal(res["leaves"], o_leaves) and np.array_equal(res["nodes"], o_nodes) and np.array_equal(res["root"], o_nodes[1])
    finally:
        ctx.lib.wf_comm_destroy(cm)


@pytest.mark.parametrize("G,N,D,log_len,layers", [(2, 4, 2, 12, 3), (4, 4, 1, 12, 2), (8, 4, 2, 13, 2), (2, 2, 3, 10, 4), (4, 16, 1, 14, 2), (1, 4, 2, 10, 2)])
def test_sharded_fri_layers_over_loopback(wf, oracle, G, N, D, log_len, layers):
    """wf_comm_sharded_fri_layers rank for rank (loopback transport, one thread per rank, every rank its own context and its own
    copy of the device coin) against the single-device commit phase of the oracle: each rank's rows, leaves, subtree, the top
    tree, every root and alpha, each rank's piece of every folded vector.  G = 8 with folding 4 takes the all-gather re-stride,
    the others the all-to-all."""
    ctx, crypto, prover, fields = wf
    import torch
    f, hasher, lib = fields.f64, crypto.Blake3_256, ctx.lib
    n = 1 << log_len
    ev = oracle.f64_from_int(rand_field(7000 + G * 16 + N, n * D))
    handles, comms = _make_loopback(ctx, G)
    per = n // G
    bufs = []
    image = np.zeros(64, dtype=np.uint8)
    image[:32] = crypto.DefaultRandomCoin(hasher, f, np.zeros(0, dtype=np.uint64), ctx).seed
    for r in range(G):
        b = dict(piece=ctx.to_device(ev[r * per * D:(r + 1) * per * D]), coin=ctx.to_device(image), rows=[], leaves=[], nodes=[], top=[], folded=[],
                 roots=ctx.empty_u8(layers, 32), alphas=ctx.empty_u64(layers, D))
        length = n
        for _ in range(layers):
            rl = length // N // G
            b["rows"].append(ctx.empty_u64(rl, N * D)); b["leaves"].append(ctx.empty_u8(rl, 32)); b["nodes"].append(ctx.empty_u8(rl, 32))
            b["top"].append(ctx.empty_u8(G, 32)); b["folded"].append(ctx.empty_u64(rl * D))
            length //= N
        bufs.append(b)
    torch.cuda.synchronize()
    off = f.element_words(f.new(7))
    errs = []

    def run(r):
        try:
            b = bufs[r]
            arr = lambda ts: (ctypes.c_void_p * layers)(*[t.data_ptr() for t in ts])
            st = lib.wf_comm_sharded_fri_layers(comms[r], hasher.HASH_ID, f.ID, D, _vp(b["piece"]), log_len, N, layers, off.ctypes.data_as(ctypes.c_void_p),
                                                _vp(b["coin"]), arr(b["rows"]), arr(b["leaves"]), arr(b["nodes"]), arr(b["top"]), arr(b["folded"]),
                                                _vp(b["roots"]), _vp(b["alphas"]))
            assert st == 0, "wf_comm_sharded_fri_layers -> %d" % st
            assert lib.wf_ctx_sync(handles[r]) == 0
        except BaseException as e:        # noqa: BLE001
            errs.append(e)
    try:
        ts = [threading.Thread(target=run, args=(r,)) for r in range(G)]
        [t.start() for t in ts]
        [t.join(timeout=300) for t in ts]
        assert not errs, errs
    finally:
        _teardown(ctx, handles, comms)
    # the single-device commit phase
    chan = oracle.ProverChannel(0, D)
    cur, length = ev.copy(), n
    for k in range(layers):
        tr = oracle.transpose_slice(cur, N, D).reshape(length // N, N * D)
        o_leaves, o_nodes = oracle.fri_layer_commit(0, tr.reshape(-1), N, D)
        chan.commit_fri_layer(o_nodes[1])
        alpha = chan.draw_fri_alpha()
        nxt = oracle.apply_drp(tr.reshape(-1), N, fields.new(7), alpha, D).reshape(length // N, D)
        rl = length // N // G
        for r in range(G):
            b = bufs[r]
            assert np.array_equal(ctx.to_host(b["rows"][k]), tr[r * rl:(r + 1) * rl]), "rows of rank %d, layer %d" % (r, k)
            assert np.array_equal(ctx.to_host(b["leaves"][k]), o_leaves[r * rl:(r + 1) * rl])
            got_nodes = ctx.to_host(b["nodes"][k])
            for j in range(1, rl):
                depth = j.bit_length() - 1
                assert np.array_equal(got_nodes[j], o_nodes[((G + r) << depth) + (j - (1 << depth))]), (r, k, j)
            top = ctx.to_host(b["top"][k])
            if G > 1:
                assert np.array_equal(top[1:], o_nodes[1:G]), "top tree on rank %d, layer %d" % (r, k)
            else:
                assert np.array_equal(top[0], o_nodes[1])
            assert np.array_equal(ctx.to_host(b["roots"])[k], o_nodes[1]) and np.array_equal(ctx.to_host(b["alphas"])[k], alpha)
            assert np.array_equal(ctx.to_host(b["folded"][k]).reshape(rl, D), nxt[r * rl:(r + 1) * rl]), "folded piece of rank %d, layer %d" % (r, k)
        cur, length = nxt.reshape(-1), length // N


@pytest.mark.parametrize("G,min_rows,log_len", [(2, 2, 14), (4, 64, 14), (4, 2, 12), (8, 16, 15)])
def test_whole_commit_phase_over_a_communicator(wf, G, min_rows, log_len):
    """parallel.comm_sharded_fri_build_layers — sharded layers (wf_comm_sharded_fri_layers), gather, unsharded tail and remainder
    (wf_fri_build_layers) — on G loopback ranks against the single-device FriProver: the whole transcript (every layer root, the
    remainder commitment, every alpha), the remainder polynomial and the coin afterwards, on every rank."""
    ctx, crypto, prover, fields = wf
    import torch
    from winterfell_amd import fri, parallel
    from winterfell_amd._lib import Context
    f, hasher, lib, D, N = fields.f64, crypto.Blake3_256, ctx.lib, 2, 4
    n = 1 << log_len
    rng = np.random.default_rng(31 + G)
    ev = f.from_ints([int(v) % f.M for v in rng.integers(0, 1 << 63, n * D, dtype=np.uint64)])
    opts = fri.FriOptions(8, N, 7, field=f)
    # single device
    chan = fri.DefaultProverChannel(n, 8, hasher, ext_degree=D, field=f, ctx=ctx)
    single = fri.FriProver(opts, hasher, ext_degree=D, ctx=ctx)
    single.build_layers(chan, ev.copy())
    # G ranks
    rank_ctx = [Context(ctx.device.index or 0) for _ in range(G)]
    arr = (ctypes.c_void_p * G)(*[c.handle.value for c in rank_ctx])
    comms = (ctypes.c_void_p * G)()
    assert lib.wf_comm_init_loopback(arr, G, comms) == 0
    comms = [ctypes.c_void_p(c) for c in comms]
    image = np.zeros(64, dtype=np.uint8)
    image[:32] = crypto.DefaultRandomCoin(hasher, f, np.zeros(0, dtype=np.uint64), ctx).seed
    per = n // G
    pieces = [ctx.to_device(ev[r * per * D:(r + 1) * per * D]) for r in range(G)]
    coins = [ctx.to_device(image) for _ in range(G)]
    torch.cuda.synchronize()
    out, errs = [None] * G, []

    def run(r):
        try:
            out[r] = parallel.comm_sharded_fri_build_layers(lib, comms[r], rank_ctx[r], hasher, opts, pieces[r], D, coins[r], min_rows_per_rank=min_rows)
            rank_ctx[r].sync()
        except BaseException as e:        # noqa: BLE001
            errs.append(e)
    try:
        ts = [threading.Thread(target=run, args=(r,)) for r in range(G)]
        [t.start() for t in ts]
        [t.join(timeout=300) for t in ts]
        assert not errs, errs
        total = opts.num_fri_layers(n)
        assert 0 < out[0]["num_sharded"] <= total and (min_rows <= 2 or out[0]["num_sharded"] < total)
        for r in range(G):
            roots, alphas = ctx.to_host(out[r]["roots"]), ctx.to_host(out[r]["alphas"])
            assert len(chan.commitments) == total + 1
            for k in range(total + 1):
                assert np.array_equal(roots[k], chan.commitments[k]), "rank %d, commitment %d" % (r, k)
            for k in range(total):
                assert np.array_equal(alphas[k], chan.alphas[k]), "rank %d, alpha %d" % (r, k)
            assert np.array_equal(ctx.to_host(out[r]["remainder"]), single.remainder_poly)
            st = ctx.to_host(coins[r])
            assert np.array_equal(st[:32], chan.public_coin.seed)
    finally:
        for c in comms:
            lib.wf_comm_destroy(c)
        for c in rank_ctx:
            c.close()


def test_a_failing_rank_releases_its_peers(wf, oracle):
    """round-2 advice: on the loopback transport a rank that errors out between two collectives used to leave every peer blocked
    at the thread barrier forever.  Rank 1 asks for a hasher its field does not have (Rp64_256 over f128: the row hash of
    wf_comm_sharded_commit fails after the LDE, before the digest all-to-all); rank 0, with valid arguments, must come back with
    WF_ERR_COMM_ABORTED instead of hanging, and a later collective on the same communicator fails the same way."""
    ctx, crypto, prover, fields = wf
    import torch
    lib = ctx.lib
    G, log_n, log_b = 2, 6, 2
    n, N = 1 << log_n, 1 << (log_n + log_b)
    handles, comms = _make_loopback(ctx, G)
    status = [None] * G
    try:
        f64, f128 = fields.f64, fields.f128
        tr0 = ctx.to_device(oracle.f64_from_int(rand_field(1, 2 * n)).reshape(2, n))
        tr1 = ctx.to_device(np.random.default_rng(2).integers(0, 1 << 62, (2, n * 2), dtype=np.uint64))
        bufs = [dict(lde=ctx.empty_u64(N, 8 * w), leaves=ctx.empty_u8(N // G, 32), nodes=ctx.empty_u8(N // G, 32), top=ctx.empty_u8(G, 32)) for w in (1, 2)]
        off64, off128 = f64.element_words(f64.new(7)), f128.element_words(3)
        torch.cuda.synchronize()

        def run(r):
            fld, tr, off, hid = ((f64, tr0, off64, crypto.Blake3_256.HASH_ID), (f128, tr1, off128, crypto.Rp64_256.HASH_ID))[r]
            b = bufs[r]
            # col_stride is in base-field ELEMENTS (n for both fields; n * W walked off the end of rank 1's trace: found by the guard session)
            status[r] = lib.wf_comm_sharded_commit(comms[r], hid, fld.ID, 1, _vp(tr), 2, n, log_n, log_b, off.ctypes.data_as(ctypes.c_void_p), 0,
                                                   _vp(b["lde"]), _vp(b["leaves"]), _vp(b["nodes"]), _vp(b["top"]), None)

        ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(G)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=60)
        assert not any(t.is_alive() for t in ts), "a rank is still blocked in a collective"
        assert status[1] == 5                                       # WF_ERR_UNSUPPORTED: Rescue over f128
        assert status[0] == 10                                      # WF_ERR_COMM_ABORTED
        assert b"peer rank" in lib.wf_strerror(10)
        a, b2 = ctx.empty_u8(32), ctx.empty_u8(64)
        assert lib.wf_comm_all_gather(comms[0], _vp(a), _vp(b2), 32) == 10
    finally:
        _teardown(ctx, handles, comms)


def test_sharded_fri_layers_checks_every_layer_before_the_first_collective(wf):
    """round-2 advice: a bad layer k used to be found after layers 0 .. k-1 had run collectives and reseeded the coin.  Here layer 1
    carries a null pointer: the call returns WF_ERR_INVALID_ARG and the coin is untouched."""
    ctx, crypto, prover, fields = wf
    lib, f = ctx.lib, fields.f64
    handles, comms = _make_loopback(ctx, 1)
    try:
        D, N, log_len = 1, 4, 10
        piece = ctx.to_device(np.random.default_rng(3).integers(0, fields.M, 1 << log_len, dtype=np.uint64))
        coin = crypto.DefaultRandomCoin(crypto.Blake3_256, f, np.zeros(0, dtype=np.uint64), ctx).to_device()
        coin.draw(1)
        before = ctx.to_host(coin.state).copy()
        rows = [ctx.empty_u64(256, 4), ctx.empty_u64(64, 4)]
        lv = [ctx.empty_u8(r, 32) for r in (256, 64)]
        nd = [ctx.empty_u8(r, 32) for r in (256, 64)]
        tp = [ctx.empty_u8(1, 32) for _ in range(2)]
        fo = [ctx.empty_u64(r) for r in (256, 64)]
        arr = lambda ts: (ctypes.c_void_p * 2)(*[t.data_ptr() for t in ts])
        bad_rows = (ctypes.c_void_p * 2)(rows[0].data_ptr(), None)
        roots, alphas = ctx.empty_u8(2, 32), ctx.empty_u64(2, 1)
        off = f.element_words(f.new(7))
        st = lib.wf_comm_sharded_fri_layers(comms[0], crypto.Blake3_256.HASH_ID, f.ID, D, _vp(piece), log_len, N, 2, off.ctypes.data_as(ctypes.c_void_p),
                                            _vp(coin.state), bad_rows, arr(lv), arr(nd), arr(tp), arr(fo), _vp(roots), _vp(alphas))
        assert st == 1                                              # WF_ERR_INVALID_ARG
        ctx.sync()
        assert np.array_equal(ctx.to_host(coin.state), before)
    finally:
        _teardown(ctx, handles, comms)
