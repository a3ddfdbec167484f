"""GPU parity: math::fft through the C ABI vs the CPU oracle (bit-exact on the internal u64 form)."""
import numpy as np
import pytest

from conftest import P, rand_field, splitmix64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wf():
    import winterfell_amd
    from winterfell_amd.math import fft, fields
    return winterfell_amd.default_context(), fft, fields


def test_library_is_native(wf):
    ctx, _, _ = wf
    import winterfell_amd._lib as L
    assert L.load_library().wf_version() >= 100
    assert L.LIB_PATH.endswith("winterfell_amd/libwinterfell_hip.so")


def test_twiddles_match_reference_layout(wf, oracle):
    ctx, fft, _ = wf
    for n in (2, 4, 16, 1024, 1 << 13, 1 << 17):
        assert np.array_equal(ctx.to_host(fft.get_twiddles(n)), oracle.get_twiddles(n)), n
        assert np.array_equal(ctx.to_host(fft.get_inv_twiddles(n)), oracle.get_inv_twiddles(n)), n


def test_golden_vectors(wf, oracle, golden):
    ctx, fft, fields = wf
    d = golden["derived"]
    p = fields.from_ints(list(range(1, 9)))
    assert list(fields.to_ints(fft.evaluate_poly(p.copy()))) == d["f64_ntt8_1_to_8"]
    p = fields.from_ints(d["f64_ntt16_in"])
    assert list(fields.to_ints(fft.evaluate_poly(p.copy()))) == d["f64_ntt16_out"]
    out = fft.evaluate_poly_with_offset(fields.from_ints([1, 2, 3, 4]), None, fields.new(7), 2)
    assert list(fields.to_ints(out)) == d["f64_lde_1234_b2_o7"]
    out = fft.evaluate_poly_with_offset(fields.from_ints(d["f64_ntt16_in"]), None, fields.new(7), 8)
    assert list(fields.to_ints(out)) == d["f64_lde16_b8_o7"]


@pytest.mark.parametrize("log_n", list(range(1, 18)) + [19, 20])
def test_evaluate_interpolate_vs_oracle(wf, oracle, log_n):
    ctx, fft, fields = wf
    n = 1 << log_n
    p = oracle.f64_from_int(splitmix64(0x5EED0001 + log_n, n) if log_n <= 14 else rand_field(log_n, n))
    want = oracle.evaluate_poly(p, par=log_n >= 12)
    got = fft.evaluate_poly(p.copy())
    assert np.array_equal(got, want), "evaluate_poly n=2^%d" % log_n
    back = fft.interpolate_poly(got.copy())
    assert np.array_equal(back, p), "interpolate_poly n=2^%d" % log_n
    assert np.array_equal(fft.interpolate_poly(p.copy()), oracle.interpolate_poly(p, par=log_n >= 12))


def test_edge_values(wf, oracle):
    ctx, fft, fields = wf
    n = 1 << 10
    for fill in (0, 1, P - 1):
        p = oracle.f64_from_int(np.full(n, fill, dtype=np.uint64))
        assert np.array_equal(fft.evaluate_poly(p.copy()), oracle.evaluate_poly(p))
    p = oracle.f64_from_int(np.array([P - 1, 0] * (n // 2), dtype=np.uint64))
    assert np.array_equal(fft.evaluate_poly(p.copy()), oracle.evaluate_poly(p))


@pytest.mark.parametrize("D", [2, 3])
@pytest.mark.parametrize("log_n", [3, 8, 11, 13])
def test_extension_fields(wf, oracle, D, log_n):
    ctx, fft, fields = wf
    n = 1 << log_n
    p = oracle.f64_from_int(splitmix64(D * 100 + log_n, n * D))
    assert np.array_equal(fft.evaluate_poly(p.copy(), ext_degree=D), oracle.evaluate_poly(p, D=D))
    assert np.array_equal(fft.interpolate_poly(p.copy(), ext_degree=D), oracle.interpolate_poly(p, D=D))
    off = fields.new(7)
    assert np.array_equal(fft.evaluate_poly_with_offset(p, None, off, 4, ext_degree=D),
                          oracle.evaluate_poly_with_offset(p, off, 4, D=D))
    assert np.array_equal(fft.interpolate_poly_with_offset(p.copy(), None, off, ext_degree=D),
                          oracle.interpolate_poly_with_offset(p, off, D=D))


@pytest.mark.parametrize("log_n,blowup", [(1, 2), (4, 8), (9, 1), (10, 2), (12, 8), (16, 8), (17, 4)])
def test_with_offset_vs_oracle(wf, oracle, log_n, blowup):
    ctx, fft, fields = wf
    n = 1 << log_n
    p = oracle.f64_from_int(rand_field(log_n * 31 + blowup, n))
    for off_int in (7, 3, P - 1):
        off = fields.new(off_int)
        got = fft.evaluate_poly_with_offset(p, None, off, blowup)
        assert np.array_equal(got, oracle.evaluate_poly_with_offset(p, off, blowup, par=True)), (log_n, blowup, off_int)
    ev = oracle.f64_from_int(rand_field(99 + log_n, n))
    assert np.array_equal(fft.interpolate_poly_with_offset(ev.copy(), None, fields.new(7)),
                          oracle.interpolate_poly_with_offset(ev, fields.new(7)))


def test_batched_vectors(wf, oracle):
    ctx, fft, fields = wf
    n, batch = 1 << 9, 37
    p = oracle.f64_from_int(rand_field(5, n * batch)).reshape(batch, n)
    got = fft.evaluate_poly(p.copy(), batch=batch)
    for v in (0, 1, 17, 36):
        assert np.array_equal(got[v], oracle.evaluate_poly(p[v]))


def test_error_behaviour(wf):
    ctx, fft, fields = wf
    with pytest.raises(AssertionError, match="power of 2"):
        fft.evaluate_poly(np.zeros(12, dtype=np.uint64))          # fft/mod.rs:90
    with pytest.raises(AssertionError, match="twiddles"):
        fft.evaluate_poly(np.zeros(16, dtype=np.uint64), twiddles=np.zeros(4, dtype=np.uint64))   # mod.rs:91-96
    with pytest.raises(AssertionError, match="offset cannot be zero"):
        fft.evaluate_poly_with_offset(np.zeros(16, dtype=np.uint64), None, 0, 2)                  # mod.rs:185


@pytest.mark.parametrize("log_n", [22, 24])
def test_full_size_properties(wf, oracle, log_n):
    """BASELINE configs[1] sizes: round trip, linearity and spot values (Horner on the CPU oracle)."""
    ctx, fft, fields = wf
    import torch
    n = 1 << log_n
    a = oracle.f64_from_int(rand_field(1000 + log_n, n))
    b = oracle.f64_from_int(rand_field(2000 + log_n, n))
    da, db = ctx.to_device(a), ctx.to_device(b)
    ea = fft.evaluate_poly(da.clone())
    eb = fft.evaluate_poly(db.clone())
    # round trip == identity
    assert torch.equal(fft.interpolate_poly(ea.clone()), da)
    # spot values against the definition: p(w^k) by Horner
    host = ctx.to_host(ea)
    w = oracle.f64_root_of_unity(log_n)
    for k in (0, 1, 12345, n // 2 + 7, n - 1):
        assert host[k] == oracle.poly_eval(a, oracle.f64_exp(w, k)), k
    # linearity: NTT(a + b) == NTT(a) + NTT(b) (sum computed on the CPU oracle for a sample)
    s = np.array([oracle.f64_add(int(x), int(y)) for x, y in zip(a[:4096], b[:4096])], dtype=np.uint64)
    full_sum = a.copy()
    # vectorised modular add on canonical Montgomery residues
    t = a.astype(object) + b.astype(object)
    full_sum = np.where(t >= P, t - P, t).astype(np.uint64)
    assert np.array_equal(full_sum[:4096], s)
    es = ctx.to_host(fft.evaluate_poly(ctx.to_device(full_sum)))
    hb = ctx.to_host(eb)
    idx = np.random.default_rng(1).integers(0, n, 2000)
    for k in idx:
        assert es[k] == oracle.f64_add(int(host[k]), int(hb[k]))


@pytest.mark.parametrize("log_n", [22, 24])
def test_full_size_output_for_output(wf, oracle, log_n):
    """BASELINE configs[1] at its own sizes, word for word: the whole 2^22 / 2^24-point evaluate_poly and interpolate_poly
    outputs against the CPU oracle's restatement of the reference's concurrent path (math/src/fft/mod.rs:85-112,264-295,
    concurrent.rs), plus the coset variants the LDE uses (mod.rs:168-211,351-386)."""
    ctx, fft, fields = wf
    n = 1 << log_n
    a = oracle.f64_from_int(rand_field(31000 + log_n, n))
    ev = ctx.to_host(fft.evaluate_poly(ctx.to_device(a)))
    want = oracle.evaluate_poly(a, par=True)
    assert np.array_equal(ev, want), "evaluate_poly differs at %d positions" % int(np.count_nonzero(ev != want))
    ip = ctx.to_host(fft.interpolate_poly(ctx.to_device(a)))
    want = oracle.interpolate_poly(a, par=True)
    assert np.array_equal(ip, want), "interpolate_poly differs at %d positions" % int(np.count_nonzero(ip != want))
    if log_n == 22:
        # coset transforms at full size: evaluate over offset * <g_{8n}> (blowup 8 of a 2^19-point polynomial = 2^22 outputs)
        # and interpolate_poly_with_offset of the 2^22-point vector
        poly = a[: n // 8].copy()
        off = fields.new(7)
        got = fft.evaluate_poly_with_offset(poly.copy(), None, off, 8)
        assert np.array_equal(got, oracle.evaluate_poly_with_offset(poly, off, 8))
        got = fft.interpolate_poly_with_offset(a.copy(), None, off)
        assert np.array_equal(got, oracle.interpolate_poly_with_offset(a, off))


@pytest.mark.parametrize("log_n", [26, 27])
def test_beyond_baseline_sizes(wf, oracle, log_n):
    """Sizes above BASELINE's 2^24 (1 GiB vector at 2^27; four passes at 2^26+): round trip and Horner spot values."""
    ctx, fft, fields = wf
    import torch
    n = 1 << log_n
    a = np.random.default_rng(log_n).integers(0, P, n, dtype=np.uint64)      # canonical words are valid Montgomery residues
    da = ctx.to_device(a)
    ea = fft.evaluate_poly(da.clone())
    assert torch.equal(fft.interpolate_poly(ea.clone()), da)
    w = oracle.f64_root_of_unity(log_n)
    ks = [0, 1, 3 * n // 4 + 11, n - 1]
    vals = ctx.to_host(ea[torch.tensor(ks, device=ea.device)])
    for k, v in zip(ks, vals):
        assert int(v) == oracle.poly_eval(a, oracle.f64_exp(w, k)), k


def test_concurrent_threads_with_their_own_contexts(oracle):
    """SURVEY 8b threading convention: `ColMatrix::interpolate_columns` calls fft::interpolate_poly from many Rayon
    threads at once (col_matrix.rs:194-199), and TraceLde must be Sync.  The library's rule is one context per calling
    thread (include/winterfell_hip.h): four threads, each with its own context on its own stream, hammer different sizes
    concurrently (ctypes drops the GIL during the calls); every result must still be the oracle's."""
    import threading
    import torch
    from winterfell_amd import crypto, prover
    from winterfell_amd._lib import Context
    from winterfell_amd.math import fft, fields
    cases = []
    for t, log_n in enumerate((10, 13, 16, 12)):
        p = oracle.f64_from_int(rand_field(900 + t, 1 << log_n))
        cases.append((p, oracle.evaluate_poly(p, par=True), oracle.evaluate_poly_with_offset(p, oracle.f64_new(7), 8, par=True)))
    errors = []

    def worker(t):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                ctx = Context(0)
                p, want_ev, want_lde = cases[t]
                for it in range(12):
                    got = fft.evaluate_poly(p.copy(), ctx=ctx)
                    assert np.array_equal(got, want_ev), ("evaluate", t, it)
                    assert np.array_equal(fft.interpolate_poly(got.copy(), ctx=ctx), p), ("interpolate", t, it)
                    lde = fft.evaluate_poly_with_offset(p.copy(), None, fields.new(7), 8, ctx=ctx)
                    assert np.array_equal(lde, want_lde), ("lde", t, it)
                    tree = crypto.MerkleTree.new(crypto.Blake3_256, lde.view(np.uint8).reshape(-1, 32), ctx)
                    assert np.array_equal(tree.root(), oracle.merkle_build(0, lde.view(np.uint8).reshape(-1, 32))[1]), ("merkle", t, it)
                ctx.close()
        except Exception as e:      # surfaced in the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(cases))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_registered_host_buffers_round_trip(wf, oracle):
    """wf_host_register / wf_host_unregister: a page-locked caller buffer goes through the same wf_memcpy_* entry points
    and the same transform, bit for bit."""
    import ctypes
    ctx, fft, fields = wf
    n = 1 << 16
    p = oracle.f64_from_int(rand_field(4242, n))
    host, back = p.copy(), np.empty_like(p)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ctx.call("wf_host_register", vp(host), host.nbytes)
    ctx.call("wf_host_register", vp(back), back.nbytes)
    try:
        dev = ctx.empty_u64(n)
        ctx.call("wf_memcpy_h2d", ctypes.c_void_p(dev.data_ptr()), vp(host), host.nbytes)
        fft.evaluate_poly(dev, ctx=ctx)
        ctx.call("wf_memcpy_d2h", vp(back), ctypes.c_void_p(dev.data_ptr()), back.nbytes)
    finally:
        ctx.call("wf_host_unregister", vp(host))
        ctx.call("wf_host_unregister", vp(back))
    assert np.array_equal(back, oracle.evaluate_poly(p, par=True))
    from winterfell_amd._lib import WfError
    with pytest.raises(WfError):
        ctx.call("wf_host_register", None, 16)


def test_pageable_host_buffers_reallocated_at_the_same_address(wf):
    """Regression test of the round-3 abort (tools/repro_pinned_cache.py): a pageable host buffer is copied,
    unmapped, and a new buffer mapped at the same address is copied again.  The HIP runtime, handed such a range directly, pins it in
    place and finds the stale pinned object again by address — a GPU page fault at a host address that takes the process down.  The
    library's copies go through its own page-locked bounce buffers, so the pattern is harmless through wf_memcpy_* (and therefore
    through Context.to_device / to_host, which every test uses)."""
    import mmap
    import time
    import torch
    from winterfell_amd._lib import _vp
    ctx = wf[0]
    size = 4 << 20
    n = size // 8
    d = torch.arange(n, dtype=torch.int64, device=ctx.device)
    d2 = torch.empty_like(d)
    want = np.arange(n, dtype=np.int64)
    same, last = 0, None
    for it in range(24):
        mm = mmap.mmap(-1, size)
        arr = np.frombuffer(mm, dtype=np.int64)
        addr = arr.ctypes.data
        same += int(addr == last)
        last = addr
        ctx.call("wf_memcpy_d2h", _vp(addr), _vp(d.data_ptr()), size)
        assert np.array_equal(arr, want), it
        ctx.call("wf_memcpy_h2d", _vp(d2.data_ptr()), _vp(addr), size)
        assert torch.equal(d, d2), it
        del arr
        mm.close()
        time.sleep(0.005)                   # the driver's restore worker finds the range unmapped
    assert same >= 12, "the allocator did not hand the same address back (%d of 24): the test did not exercise the pattern" % same


def test_one_context_called_from_many_threads(wf, oracle):
    """TraceLde: Sync (prover/src/trace/trace_lde/mod.rs:26; read_main_trace_frame_into is called from Rayon workers,
    constraints/evaluator/default.rs:187): every entry point locks its context, so several host threads may call into ONE context —
    the calls run one after the other and every result is right."""
    import threading
    ctx, fft, fields = wf
    from winterfell_amd import crypto
    cases = []
    for t, log_n in enumerate((9, 12, 15, 11, 14, 10)):
        p = oracle.f64_from_int(rand_field(700 + t, 1 << log_n))
        cases.append((p, oracle.evaluate_poly(p, par=True)))
    errors = []

    def worker(t):
        try:
            p, want = cases[t]
            for it in range(10):
                got = fft.evaluate_poly(p.copy(), ctx=ctx)
                assert np.array_equal(got, want), ("evaluate", t, it)
                assert np.array_equal(fft.interpolate_poly(got.copy(), ctx=ctx), p), ("interpolate", t, it)
                lv = got.view(np.uint8).reshape(-1, 32)
                assert np.array_equal(crypto.MerkleTree.new(crypto.Blake3_256, lv, ctx).root(), oracle.merkle_build(0, lv)[1]), ("merkle", t, it)
        except Exception as e:
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(cases))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_serial_fft_permute_index_infer_degree(wf, oracle):
    """the rest of math::fft's public surface: serial_fft (mod.rs:405-429), permute_index (:570-578), infer_degree
    (:543-562, the doc example and a random polynomial of known degree)."""
    ctx, fft, fields = wf
    n = 1 << 9
    p = oracle.f64_from_int(rand_field(808, n))
    assert np.array_equal(fft.serial_fft(p.copy(), fft.get_twiddles(n)), oracle.evaluate_poly(p))
    with pytest.raises(AssertionError):
        fft.serial_fft(p.copy(), fft.get_twiddles(n // 2))
    assert [fft.permute_index(8, i) for i in range(8)] == [0, 4, 2, 6, 1, 5, 3, 7] and fft.permute_index(1, 0) == 0
    tw = ctx.to_host(fft.get_twiddles(16))
    root = pow(fields.f64.get_root_of_unity(4), 1, fields.M)
    assert all(int(fields.to_ints(tw)[fft.permute_index(8, i)]) == pow(root, i, fields.M) for i in range(8))      # mod.rs:464-467
    # doc example: p(x) = x^2 + 1 over the coset of size 4 has degree 2
    ev = oracle.evaluate_poly_with_offset(fields.from_ints([1, 0, 1, 0]), oracle.f64_new(7), 1)
    assert fft.infer_degree(ev, fields.new(7)) == 2
    for deg, D in ((100, 1), (37, 2), (0, 1)):
        coeffs = np.zeros(256 * D, dtype=np.uint64)
        coeffs[:(deg + 1) * D] = oracle.f64_from_int(rand_field(deg + 3, (deg + 1) * D) | np.uint64(1))
        ev = oracle.evaluate_poly_with_offset(coeffs, oracle.f64_new(7), 4, D=D)
        assert fft.infer_degree(ev, fields.new(7), ext_degree=D) == deg
    with pytest.raises(AssertionError):
        fft.infer_degree(ev, 0)


@pytest.mark.parametrize("fname", ["f64", "f128", "f62"])
def test_power_series_and_batch_inversion(wf, fname):
    """math::utils (utils/mod.rs:36-79, 169-215) against python big-int arithmetic: s * b^i for ragged lengths, inverses with
    zeros in the input staying zero."""
    ctx, fft, fields = wf
    from winterfell_amd.math import utils
    f = getattr(fields, fname)
    b, s = 3, 0x123456789ABCDEF % f.M
    for n in (0, 1, 5, 16, 17, 1000, (1 << 16) + 3):
        got = f.to_ints(ctx.to_host(utils.get_power_series_with_offset(f.new(b), f.new(s), n, field=f)))
        want, cur = [], s
        for _ in range(min(n, 1000)):
            want.append(cur)
            cur = cur * b % f.M
        assert got[:len(want)] == want, n
        if n > 1000:
            assert got[n - 1] == s * pow(b, n - 1, f.M) % f.M
    assert f.to_ints(ctx.to_host(utils.get_power_series(f.new(b), 4, field=f))) == [1, 3, 9, 27]
    rng = np.random.default_rng(5)
    vals = [int(v) % f.M for v in rng.integers(1, 1 << 62, 1003)]
    for z in (0, 15, 16, 500, 1002):
        vals[z] = 0
    vals[1] = f.M - 1
    got = f.to_ints(utils.batch_inversion(f.from_ints(vals), field=f))
    assert got == [pow(v, f.M - 2, f.M) if v else 0 for v in vals]


def test_follows_torchs_current_stream(wf, oracle):
    """The context re-binds to torch's current stream on every call: library kernels issued inside `with torch.cuda.stream(s)`
    stay ordered with the torch work (clone / copies) issued on that stream around them."""
    ctx, fft, fields = wf
    import torch
    n = 1 << 16
    a = oracle.f64_from_int(rand_field(555, n))
    want = oracle.evaluate_poly(a)
    s = torch.cuda.Stream(device=ctx.device)
    for _ in range(5):
        with torch.cuda.stream(s):
            d = ctx.to_device(a)
            e = fft.evaluate_poly(d.clone())
            back = fft.interpolate_poly(e.clone())
            got, rt = ctx.to_host(e), ctx.to_host(back)
        assert np.array_equal(got, want) and np.array_equal(rt, a)
    torch.cuda.synchronize()
    assert np.array_equal(ctx.to_host(fft.evaluate_poly(ctx.to_device(a))), want)     # and back on the default stream


@pytest.mark.parametrize("plan", ["18:6,6,6", "18:8,5,5", "18:5,8,5", "18:4,7,7", "18:8,8,2", "18:7,7,3,1", "18:3,3,3,3,3,3", "18:1,8,8,1"])
def test_every_pass_plan_gives_the_same_transform(wf, oracle, plan, monkeypatch):
    """WF_NTT_PLAN (the measurement hook of tools/time_batch_ntt.py) splits a 2^18-point transform into passes of any radix
    2^1 .. 2^8 in any position (first / middle / last use different kernels): forward, inverse, coset evaluation and a batch of
    vectors must not depend on the split.  The variable is read ONCE, by wf_ctx_create (round-2 advice: a stray variable must not
    change the pass shapes of a running process), so every plan gets a context of its own; the default context, created without it,
    keeps the built-in plan."""
    from winterfell_amd._lib import Context
    ctx, fft, fields = wf[0], wf[1], wf[2]
    n = 1 << 18
    p = oracle.f64_from_int(rand_field(321, n))
    monkeypatch.setenv("WF_NTT_PLAN", plan)
    planned = Context(ctx.device.index or 0)
    try:
        assert np.array_equal(fft.evaluate_poly(p.copy(), ctx=planned), oracle.evaluate_poly(p, par=True))
        assert np.array_equal(fft.interpolate_poly(p.copy(), ctx=planned), oracle.interpolate_poly(p, par=True))
        vecs = oracle.f64_from_int(rand_field(322, 4 * n)).reshape(4, n)
        want = np.stack([oracle.evaluate_poly(vecs[k], par=True) for k in range(4)])
        assert np.array_equal(np.asarray(fft.evaluate_poly(vecs.copy(), batch=4, ctx=planned)).reshape(4, n), want)
        # the plan is really in force: the number of pass launches of one transform is the plan's
        planned.prof_enable(True)
        fft.evaluate_poly(p.copy(), ctx=planned)
        launches = sum(c for _, (c, _ms) in planned.prof_collect().items())
        planned.prof_enable(False)
        assert launches == len(plan.split(":")[1].split(","))
    finally:
        planned.sync()
        planned.close()
