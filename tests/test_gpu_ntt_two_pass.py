"""GPU parity of the two-pass NTT plans (csrc/ntt_big.cuh: three-step passes of radix 2^10 .. 2^12) against the CPU oracle, bit
for bit: math::fft evaluate / interpolate (math/src/fft/mod.rs:85-386) at 2^20 .. 2^24 points, extension fields, coset
evaluation / interpolation, and the batched row-major LDE behind RowMatrix::evaluate_polys_over
(prover/src/matrix/row_matrix.rs:84-100) with ragged column groups.  The host-side emulation of the same code
(tests/cpp/ntt_big_host_test.cpp, run by test_host_logic.py) covers the index arithmetic without a GPU."""
import numpy as np
import pytest

from conftest import P, rand_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(request):
    """a context of its own with WF_NTT_BIG=1 (read once, by wf_ctx_create): every eligible f64 transform takes two passes"""
    import os
    import winterfell_amd
    from winterfell_amd._lib import Context
    from winterfell_amd.math import fft, fields
    old = os.environ.get("WF_NTT_BIG")
    os.environ["WF_NTT_BIG"] = "1"
    try:
        ctx = Context(winterfell_amd.default_context().device.index or 0)
    finally:
        if old is None:
            del os.environ["WF_NTT_BIG"]
        else:
            os.environ["WF_NTT_BIG"] = old
    yield ctx, fft, fields
    ctx.sync()
    ctx.close()


def _launches(ctx, fn):
    ctx.prof_enable(True)
    fn()
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    return prof


@pytest.mark.parametrize("log_n", [20, 21, 22])
def test_two_pass_transform_vs_oracle(big, oracle, log_n):
    ctx, fft, fields = big
    n = 1 << log_n
    p = oracle.f64_from_int(rand_field(1000 + log_n, n))
    want = oracle.evaluate_poly(p, par=True)
    got = fft.evaluate_poly(p.copy(), ctx=ctx)
    assert np.array_equal(got, want), "evaluate_poly n=2^%d" % log_n
    assert np.array_equal(fft.interpolate_poly(got.copy(), ctx=ctx), p), "round trip n=2^%d" % log_n
    assert np.array_equal(fft.interpolate_poly(p.copy(), ctx=ctx), oracle.interpolate_poly(p, par=True))
    # the plan is in force: two launches of the three-step kernels
    prof = _launches(ctx, lambda: fft.evaluate_poly(p.copy(), ctx=ctx))
    assert sorted(prof) == ["ntt_pass3", "ntt_pass3_last"] and all(c == 1 for c, _ in prof.values()), prof


@pytest.mark.parametrize("log_n", [23, 24])
def test_two_pass_equals_three_pass_at_full_size(big, log_n):
    """radix-4096 passes (one 1024-lane workgroup per CU): the two plans against each other, word for word, and the round trip; the
    three-pass plan is the one tests/test_gpu_fft.py holds against the oracle at these sizes"""
    import torch
    import winterfell_amd
    ctx, fft, fields = big
    base = winterfell_amd.default_context()
    n = 1 << log_n
    d = base.to_device(np.random.default_rng(log_n).integers(0, fields.M, n, dtype=np.uint64))
    want = fft.evaluate_poly(d.clone(), ctx=base)
    got = fft.evaluate_poly(d.clone(), ctx=ctx)
    base.sync()
    ctx.sync()
    assert torch.equal(got, want)
    back = fft.interpolate_poly(got, ctx=ctx)
    ctx.sync()
    assert torch.equal(back, d)
    prof = _launches(ctx, lambda: fft.evaluate_poly(d.clone(), ctx=ctx))
    assert sorted(prof) == ["ntt_pass3", "ntt_pass3_last"], prof


def test_edge_values_two_pass(big, oracle):
    ctx, fft, fields = big
    n = 1 << 20
    for fill in (0, 1, P - 1):
        p = oracle.f64_from_int(np.full(n, fill, dtype=np.uint64))
        assert np.array_equal(fft.evaluate_poly(p.copy(), ctx=ctx), oracle.evaluate_poly(p, par=True)), fill
    p = np.zeros(n, dtype=np.uint64)
    p[n - 1] = oracle.f64_from_int(np.array([P - 1], dtype=np.uint64))[0]
    assert np.array_equal(fft.evaluate_poly(p.copy(), ctx=ctx), oracle.evaluate_poly(p, par=True))


@pytest.mark.parametrize("D", [2, 3])
def test_extension_fields_two_pass(big, oracle, D):
    ctx, fft, fields = big
    n = 1 << 20
    p = oracle.f64_from_int(rand_field(D * 7 + 1, n * D))
    assert np.array_equal(fft.evaluate_poly(p.copy(), ext_degree=D, ctx=ctx), oracle.evaluate_poly(p, D=D, par=True))
    assert np.array_equal(fft.interpolate_poly(p.copy(), ext_degree=D, ctx=ctx), oracle.interpolate_poly(p, D=D, par=True))
    off = fields.new(7)
    assert np.array_equal(fft.evaluate_poly_with_offset(p, None, off, 2, ext_degree=D, ctx=ctx),
                          oracle.evaluate_poly_with_offset(p, off, 2, D=D, par=True))
    assert np.array_equal(fft.interpolate_poly_with_offset(p.copy(), None, off, ext_degree=D, ctx=ctx),
                          oracle.interpolate_poly_with_offset(p, off, D=D))


@pytest.mark.parametrize("log_n,blowup", [(20, 4), (21, 2)])
def test_with_offset_two_pass(big, oracle, log_n, blowup):
    ctx, fft, fields = big
    n = 1 << log_n
    p = oracle.f64_from_int(rand_field(log_n * 31 + blowup, n))
    for off_int in (7, P - 1):
        off = fields.new(off_int)
        got = fft.evaluate_poly_with_offset(p, None, off, blowup, ctx=ctx)
        assert np.array_equal(got, oracle.evaluate_poly_with_offset(p, off, blowup, par=True)), (log_n, blowup, off_int)
    ev = oracle.f64_from_int(rand_field(99 + log_n, n))
    assert np.array_equal(fft.interpolate_poly_with_offset(ev.copy(), None, fields.new(7), ctx=ctx),
                          oracle.interpolate_poly_with_offset(ev, fields.new(7)))


def test_batched_vectors_two_pass(big, oracle):
    ctx, fft, fields = big
    n, batch = 1 << 20, 5
    p = oracle.f64_from_int(rand_field(5, n * batch)).reshape(batch, n)
    got = np.asarray(fft.evaluate_poly(p.copy(), batch=batch, ctx=ctx)).reshape(batch, n)
    for v in range(batch):
        assert np.array_equal(got[v], oracle.evaluate_poly(p[v], par=True)), v


@pytest.fixture(scope="module")
def big_rm():
    """WF_NTT_BIG=1 and WF_ROWS_HASH_WIDE=0: every eligible transform in two passes, wide rows stored row-major by the last three-step
    pass (the leaves come from the separate row-hash kernel)"""
    import os
    import winterfell_amd
    from winterfell_amd._lib import Context
    os.environ["WF_NTT_BIG"] = "1"
    os.environ["WF_ROWS_HASH_WIDE"] = "0"
    try:
        ctx = Context(winterfell_amd.default_context().device.index or 0)
    finally:
        del os.environ["WF_NTT_BIG"]
        del os.environ["WF_ROWS_HASH_WIDE"]
    yield ctx
    ctx.sync()
    ctx.close()


SHAPES = [
    (8, 20, 2),       # one group of eight columns, radix-1024 passes (tiles of eight columns)
    (12, 20, 2),      # a group of eight + a ragged one, padding to sixteen
    (5, 21, 2),       # groups of four (radix 2^11 / 2^10), one ragged, padding to eight
    (9, 21, 1),       # 2^21-point columns, radix-2048 + radix-1024 passes, blowup 1
    (32, 18, 8),      # rows of 32 columns
    (20, 19, 4),      # padded row of 24 columns in a 32-column tile (radix-64 last pass), three-pass plan
    (16, 18, 8),      # rows of exactly sixteen columns
    (12, 15, 4),      # plan 8, 7: the radix-128 last pass (two 8-point DFTs per lane) with rows + leaves
    (32, 15, 2),      # the same with full 32-column rows
]


def _commit_and_compare(oracle, ctx, c, log_n, blowup):
    from winterfell_amd import crypto, prover
    from winterfell_amd.math import fields
    n = 1 << log_n
    trace = oracle.f64_from_int(rand_field(c * 1000 + log_n, n * c)).reshape(c, n)
    ctx.prof_enable(True)
    lde, tree, polys = prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(ctx.to_device(trace), 1, ctx), prover.StarkDomain(n, blowup))
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(0, trace, blowup, fields.new(7), par=True)
    assert np.array_equal(polys.to_host(), o_polys), "polys"
    assert np.array_equal(lde.to_host(), o_lde), "lde"
    assert np.array_equal(tree.leaves, o_leaves), "leaves"
    assert np.array_equal(tree.nodes, o_nodes), "nodes"
    return prof


@pytest.mark.parametrize("c,log_n,blowup", SHAPES)
def test_trace_commitment_default_plans_vs_oracle(oracle, c, log_n, blowup):
    """wf_build_trace_commitment on the DEFAULT context, every word against the oracle: rows of 9 .. 32 f64 columns get their
    Blake3_256 leaves from the last NTT pass (rows + leaves mode for wide rows: no row-hash launch), batches of 2^20-point vectors
    take the two-pass plan by themselves"""
    import winterfell_amd
    prof = _commit_and_compare(oracle, winterfell_amd.default_context(), c, log_n, blowup)
    if 8 < c <= 32 and not (log_n + (blowup.bit_length() - 1) >= 24 and c > 16):
        assert "ntt_pass_last_rows_hash" in prof and "hash_rows_blake3" not in prof, prof
    if log_n == 20:
        assert "ntt_pass3_last" in prof, prof          # the interpolation of >= 8 columns


@pytest.mark.parametrize("c,log_n,blowup", SHAPES[:4])
def test_trace_commitment_two_pass_row_major_vs_oracle(oracle, big_rm, c, log_n, blowup):
    """the same with the row-major store of the three-step last pass (ragged column groups, zero padding)"""
    prof = _commit_and_compare(oracle, big_rm, c, log_n, blowup)
    assert "ntt_pass3_last" in prof and (c <= 8 or "ntt_pass_last_rows_hash" not in prof), prof


def test_full_size_wide_rows_hash_properties(oracle):
    """The bench shape 2^22 rows x 32 f64 columns, blowup 8, through the rows + leaves last pass (plan 8, 8, 6 for
    32-column rows): size-independent properties instead of a full CPU run — LDE rows by Horner evaluation of the returned trace
    polynomials, leaves by the oracle's hash of those rows, a Merkle path, and the root against the same commitment built with
    the separate row-hash kernel (WF_ROWS_HASH_WIDE=0, the round-4 path that the suite holds against the oracle word for word)."""
    import os
    import torch
    import winterfell_amd
    from winterfell_amd import crypto, prover
    from winterfell_amd._lib import Context
    from winterfell_amd.math import fields
    ctx = winterfell_amd.default_context()
    n, c, b = 1 << 22, 32, 8
    trace = ctx.to_device(np.random.default_rng(2232).integers(0, fields.M, (c, n), dtype=np.uint64))
    ctx.prof_enable(True)
    lde, tree, polys = prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(trace.clone(), 1, ctx), prover.StarkDomain(n, b))
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    assert "ntt_pass_last_rows_hash" in prof and "hash_rows_blake3" not in prof, prof
    N = n * b
    hp = polys.to_host()
    g = oracle.f64_root_of_unity(25)
    pos = [0, 1, 9, 4095, N // 2 + 3, N - 1]
    rows = lde.rows(pos)
    for r, k in zip(rows, pos):
        x = oracle.f64_mul(fields.new(7), oracle.f64_exp(g, k))
        for col in (0, 7, 8, 31):
            assert r[col] == oracle.poly_eval(hp[col], x), (k, col)
    leaves = tree.leaves
    for k, r in zip(pos, rows):
        assert np.array_equal(leaves[k], oracle.hash_elements(0, r)), k
    leaf, proof = tree.prove(4095)
    crypto.MerkleTree.verify(crypto.Blake3_256, tree.root(), 4095, leaf, proof)
    root = tree.root().copy()
    del lde, tree, polys
    torch.cuda.empty_cache()
    os.environ["WF_ROWS_HASH_WIDE"] = "0"
    try:
        other = Context(ctx.device.index or 0)
    finally:
        del os.environ["WF_ROWS_HASH_WIDE"]
    try:
        _, tree2, _ = prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(trace.clone(), 1, other), prover.StarkDomain(n, b))
        assert np.array_equal(tree2.root(), root)
        del tree2
    finally:
        other.sync()
        other.close()


@pytest.mark.parametrize("c,D,log_n,blowup", [(6, 2, 13, 8), (5, 3, 14, 4), (16, 2, 12, 8)])
def test_wide_rows_hash_with_extension_columns_vs_oracle(oracle, c, D, log_n, blowup):
    """an auxiliary-segment shaped matrix (columns over the quadratic / cubic extension: 12, 15 and 32 base columns per row) through
    the rows + leaves last pass: polys, every LDE word, leaves and nodes against the oracle"""
    import winterfell_amd
    from winterfell_amd import crypto, prover
    from winterfell_amd.math import fields
    ctx = winterfell_amd.default_context()
    n = 1 << log_n
    cols = oracle.f64_from_int(rand_field(c * 100 + D, n * c * D)).reshape(c, n * D)
    ctx.prof_enable(True)
    lde, tree, polys = prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(cols, ext_degree=D), prover.StarkDomain(n, blowup))
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(0, cols, blowup, fields.new(7), D=D, par=True)
    assert np.array_equal(polys.to_host(), o_polys), "polys"
    assert np.array_equal(lde.to_host(), o_lde), "lde"
    assert np.array_equal(tree.leaves, o_leaves), "leaves"
    assert np.array_equal(tree.nodes, o_nodes), "nodes"
    assert "ntt_pass_last_rows_hash" in prof and "hash_rows_blake3" not in prof, prof


# ---- vector tiles (round 6, ntt_pass<..., VT>): the coset LDE of a wide f64 trace on column-interleaved buffers ----
VT_SHAPES = [          # (columns, log2 rows, blowup, takes vector tiles)
    (32, 12, 4, True),      # two radix-64 passes, rows + leaves last pass, blowup 4 (the smallest blowup that takes vector tiles)
    (32, 12, 16, True),     # blowup 16: no coset window in the first pass's tile order (linear order)
    (32, 13, 64, True),     # the largest blowup the mode takes; radix 128 + 64
    (64, 13, 8, True),      # 64 columns: row-major last pass (two groups of 32) + separate row hash
    (96, 15, 4, True),      # 96 columns, radix 256 + 128: three column groups, tiles of 16 and of 32 columns
    (48, 16, 8, True),      # 48 = 32 + a ragged group of 16 in the last pass; 2^16 = 256 x 256: the first pass has tiles of 16 columns
    (32, 17, 8, False),     # plan 6, 6, 5: a radix-32 last pass has no vector-tile variant — position-major tiles
    (40, 12, 8, False),     # not a multiple of the tile width
    (32, 12, 2, False),     # blowup 2: the interleaving copy costs more than the shared twiddles save
]


@pytest.mark.parametrize("c,log_n,blowup,eligible", VT_SHAPES)
def test_vector_tile_lde_vs_oracle_and_vs_position_major_tiles(oracle, c, log_n, blowup, eligible):
    """wf_build_trace_commitment through the vector-tile passes, every polynomial, LDE word, leaf and node against the oracle, and the
    same call on a context created with WF_LDE_VT=0 (position-major tiles, per-lane twiddle progressions): the same root"""
    import os
    import winterfell_amd
    from winterfell_amd import crypto, prover
    from winterfell_amd._lib import Context
    from winterfell_amd.math import fields
    ctx = winterfell_amd.default_context()
    prof = _commit_and_compare(oracle, ctx, c, log_n, blowup)
    assert ("vt_interleave" in prof) == eligible, prof
    os.environ["WF_LDE_VT"] = "0"
    try:
        other = Context(ctx.device.index or 0)
    finally:
        del os.environ["WF_LDE_VT"]
    try:
        n = 1 << log_n
        trace = oracle.f64_from_int(rand_field(c * 1000 + log_n, n * c)).reshape(c, n)
        other.prof_enable(True)
        _, tree2, _ = prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(other.to_device(trace), 1, other), prover.StarkDomain(n, blowup))
        prof2 = other.prof_collect()
        other.prof_enable(False)
        assert "vt_interleave" not in prof2
        _, tree1, _ = prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(ctx.to_device(trace), 1, ctx), prover.StarkDomain(n, blowup))
        assert np.array_equal(tree1.root(), tree2.root())
    finally:
        other.sync()
        other.close()


def test_vector_tiles_serve_evaluate_polys_over_and_other_hashers(oracle):
    """RowMatrix::evaluate_polys_over (no leaves) and a Rescue commitment of a 32-column trace: the LDE comes from the vector-tile
    passes (row-major last pass), the rows are hashed by the hasher's own kernel"""
    import winterfell_amd
    from winterfell_amd import crypto, prover
    from winterfell_amd.math import fields
    ctx = winterfell_amd.default_context()
    c, log_n, b = 32, 12, 8
    n = 1 << log_n
    trace = oracle.f64_from_int(rand_field(77, n * c)).reshape(c, n)
    ctx.prof_enable(True)
    lde, tree, polys = prover.build_trace_commitment(crypto.Rp64_256, prover.ColMatrix(ctx.to_device(trace), 1, ctx), prover.StarkDomain(n, b))
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    assert "vt_interleave" in prof, prof
    o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(1, trace, b, fields.new(7), par=True)
    assert np.array_equal(polys.to_host(), o_polys) and np.array_equal(lde.to_host(), o_lde)
    assert np.array_equal(tree.nodes, o_nodes)
    m = prover.RowMatrix.evaluate_polys_over(polys, b, fields.new(7)) if hasattr(prover.RowMatrix, "evaluate_polys_over") else None
    if m is not None:
        assert np.array_equal(m.to_host(), o_lde)


# ---- block tiles (round 6, ntt_pass BT0): single three-pass f64 transforms with a transposed first pass ----
@pytest.fixture(scope="module")
def bt_ctx():
    """a context with WF_NTT_BT=1 and WF_NTT_BIG=0: every single f64 transform of three radix 64 .. 256 passes takes block tiles"""
    import os
    import winterfell_amd
    from winterfell_amd._lib import Context
    old = {k: os.environ.get(k) for k in ("WF_NTT_BT", "WF_NTT_BIG")}
    os.environ.update({"WF_NTT_BT": "1", "WF_NTT_BIG": "0"})
    try:
        ctx = Context(winterfell_amd.default_context().device.index or 0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    yield ctx
    ctx.sync()
    ctx.close()


@pytest.mark.parametrize("log_n", [18, 19, 20, 21, 22, 23, 24])
def test_block_tile_transform_vs_oracle(oracle, bt_ctx, log_n):
    """math::fft evaluate / interpolate (math/src/fft/mod.rs:85-112, 264-295) through the block-tile plan, word for word against the
    oracle's fft_in_place: forward, inverse (with its 1/n riding on the last pass's table), and the coset interpolation's post-scale"""
    import torch
    from winterfell_amd.math import fft, fields
    n = 1 << log_n
    p = oracle.f64_from_int(rand_field(7000 + log_n, n))
    dp = bt_ctx.to_device(p)
    ev = fft.evaluate_poly(dp.clone(), ctx=bt_ctx)
    assert np.array_equal(bt_ctx.to_host(ev), oracle.evaluate_poly(p, par=True))
    assert torch.equal(fft.interpolate_poly(ev, ctx=bt_ctx), dp)
    assert np.array_equal(bt_ctx.to_host(fft.interpolate_poly(dp.clone(), ctx=bt_ctx)), oracle.interpolate_poly(p, par=True))
    if log_n <= 20:
        got = fft.interpolate_poly_with_offset(dp.clone(), None, fields.new(7), ctx=bt_ctx)
        assert np.array_equal(bt_ctx.to_host(got), oracle.interpolate_poly_with_offset(p, fields.new(7)))


def test_default_context_takes_block_tiles_at_2_23(oracle):
    """the default rule: 2^23-point single transforms (-3 .. -5 % measured); the result is the standard plan's"""
    import os
    import torch
    import winterfell_amd
    from winterfell_amd._lib import Context
    from winterfell_amd.math import fft
    ctx = winterfell_amd.default_context()
    d = ctx.to_device(oracle.f64_from_int(rand_field(23, 1 << 23)))
    os.environ["WF_NTT_BT"] = "0"
    try:
        plain = Context(ctx.device.index or 0)
    finally:
        del os.environ["WF_NTT_BT"]
    try:
        assert torch.equal(fft.evaluate_poly(d.clone(), ctx=ctx), fft.evaluate_poly(d.clone(), ctx=plain))
        assert torch.equal(fft.interpolate_poly(d.clone(), ctx=ctx), fft.interpolate_poly(d.clone(), ctx=plain))
    finally:
        plain.sync()
        plain.close()
