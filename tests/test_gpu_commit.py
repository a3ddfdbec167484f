"""GPU parity: hashing, Merkle trees and the fused trace LDE + commitment vs the CPU oracle."""
import numpy as np
import pytest

from conftest import P, rand_field, splitmix64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wf():
    import winterfell_amd
    from winterfell_amd import crypto, prover
    from winterfell_amd.math import fields
    return winterfell_amd.default_context(), crypto, prover, fields


def _hid(crypto, hasher):
    return 0 if hasher is crypto.Blake3_256 else 1


def test_blake3_hash_elements_all_lengths(wf, oracle, golden):
    ctx, crypto, _, fields = wf
    h = crypto.Blake3_256
    assert h.hash_elements(fields.from_ints([1, 2])).tobytes().hex() == golden["derived"]["blake3_f64_hash_elements_1_2"]
    # 0..300 elements: crosses the 64-byte block and the 1024-byte chunk boundary (128 elements), multi-chunk tree
    for n in list(range(0, 20)) + [63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 300, 384, 385, 512, 640, 1000]:
        e = oracle.f64_from_int(splitmix64(n + 1, n)) if n else np.zeros(0, dtype=np.uint64)
        assert np.array_equal(h.hash_elements(e), oracle.hash_elements(oracle.H_BLAKE3_F64, e)), n


def test_rp64_hash_elements_and_kat(wf, oracle, golden):
    ctx, crypto, _, fields = wf
    h = crypto.Rp64_256
    d = golden["derived"]
    got = h.hash_elements(fields.from_ints([1, 2, 3, 4])).view(np.uint64)
    assert list(fields.to_ints(got)) == d["rp64_hash_elements_1_2_3_4"]
    got = h.hash_elements(fields.from_ints(list(range(19)))).view(np.uint64)
    assert list(fields.to_ints(got)) == d["rp64_hash_elements_0_to_18"]
    z = np.zeros((2, 32), dtype=np.uint8)
    assert list(fields.to_ints(h.merge(z).view(np.uint64))) == d["rp64_merge_zero"]
    # the reference's permutation KAT (rp64_256/tests.rs:70-105) through the sponge: hash_elements of the 8 rate
    # elements [4..11] with capacity word = 8 is merge(); reproduce the KAT state via the oracle-checked identity
    for n in list(range(0, 26)) + [64, 65, 100]:
        e = oracle.f64_from_int(splitmix64(500 + n, n)) if n else np.zeros(0, dtype=np.uint64)
        assert np.array_equal(h.hash_elements(e), oracle.hash_elements(oracle.H_RP64, e)), n
    edge = oracle.f64_from_int(np.full(16, P - 1, dtype=np.uint64))
    assert np.array_equal(h.hash_elements(edge), oracle.hash_elements(oracle.H_RP64, edge))


def test_merge_batches(wf, oracle):
    ctx, crypto, _, fields = wf
    rng = np.random.default_rng(2)
    pairs = rng.integers(0, 256, (300, 2, 32), dtype=np.uint8)
    got = crypto.Blake3_256.merge(pairs)
    for i in (0, 1, 150, 299):
        assert np.array_equal(got[i], oracle.merge(0, pairs[i]))
    ep = oracle.f64_from_int(rand_field(3, 300 * 8)).view(np.uint8).reshape(300, 2, 32)
    got = crypto.Rp64_256.merge(ep)
    for i in (0, 7, 299):
        assert np.array_equal(got[i], oracle.merge(1, ep[i]))
    e = oracle.f64_from_int(rand_field(4, 8))
    assert np.array_equal(crypto.Rp64_256.merge(e.view(np.uint8).reshape(2, 32)), crypto.Rp64_256.hash_elements(e))


def test_merkle_reference_fixtures(wf, oracle, golden):
    ctx, crypto, _, _ = wf
    for name in ("LEAVES4", "LEAVES8"):
        lv = np.array(golden["reference"][name], dtype=np.uint8)
        tree = crypto.MerkleTree.new(crypto.Blake3_256, lv)
        assert tree.root().tobytes().hex() == golden["derived"]["blake3_root_" + name]
        assert np.array_equal(tree.nodes, oracle.merkle_build(0, lv))
    # crypto/src/merkle/tests.rs:88-122 prove(): exact node lists
    lv = np.array(golden["reference"]["LEAVES8"], dtype=np.uint8)
    tree = crypto.MerkleTree.new(crypto.Blake3_256, lv)
    h2 = lambda a, b: oracle.merge(0, np.stack([a, b]))
    leaf, proof = tree.prove(1)
    want = [lv[0], h2(lv[2], lv[3]), h2(h2(lv[4], lv[5]), h2(lv[6], lv[7]))]
    assert np.array_equal(leaf, lv[1]) and all(np.array_equal(x, y) for x, y in zip(proof, want))
    leaf, proof = tree.prove(6)
    want = [lv[7], h2(lv[4], lv[5]), h2(h2(lv[0], lv[1]), h2(lv[2], lv[3]))]
    assert np.array_equal(leaf, lv[6]) and all(np.array_equal(x, y) for x, y in zip(proof, want))
    crypto.MerkleTree.verify(crypto.Blake3_256, tree.root(), 6, leaf, proof)
    with pytest.raises(crypto.MerkleTreeError, match="InvalidProof"):
        crypto.MerkleTree.verify(crypto.Blake3_256, tree.root(), 5, leaf, proof)
    # tests.rs:149-186 prove_batch(): exact node lists
    leaves, bp = tree.prove_batch([1, 2])
    assert bp.depth == 3 and [len(x) for x in bp.nodes] == [2, 1]
    assert np.array_equal(bp.nodes[0][0], lv[0]) and np.array_equal(bp.nodes[0][1], h2(h2(lv[4], lv[5]), h2(lv[6], lv[7])))
    assert np.array_equal(bp.nodes[1][0], lv[3])
    leaves, bp = tree.prove_batch([1, 6])
    assert np.array_equal(bp.nodes[0][1], h2(lv[2], lv[3])) and np.array_equal(bp.nodes[1][1], h2(lv[4], lv[5]))
    leaves, bp = tree.prove_batch(list(range(8)))
    assert [len(x) for x in bp.nodes] == [0, 0, 0, 0]


@pytest.mark.parametrize("log_n", [1, 2, 5, 9, 10, 11, 12, 13, 14, 17, 18, 19, 20, 21, 22])      # 2^11..2^18: one-launch tree (2..256 workgroups + ticket); >= 2^20: wave kernel first
def test_merkle_vs_oracle(wf, oracle, log_n):
    ctx, crypto, _, _ = wf
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    lv = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    assert np.array_equal(crypto.MerkleTree.new(crypto.Blake3_256, lv).nodes, oracle.merkle_build(0, lv, par=True)), log_n
    if log_n >= 20:
        lv[:, 24:] = 0
        assert np.array_equal(crypto.MerkleTree.new(crypto.Blake3_192, lv).nodes, oracle.merkle_build(5, lv, par=True)), log_n
    if log_n <= 14:
        lv = oracle.f64_from_int(rand_field(log_n, n * 4)).view(np.uint8).reshape(n, 32)
        assert np.array_equal(crypto.MerkleTree.new(crypto.Rp64_256, lv).nodes, oracle.merkle_build(1, lv, par=True)), log_n


def test_a_corrupted_merkle_ticket_is_reported_not_swallowed(wf, oracle):
    """Failure detection of the one-launch tree (round-3 review): merkle_finish_kernel's workgroups meet on a ticket word; a word that
    is not in the state the launch expects used to make the kernel skip the top of the tree SILENTLY.  Now the ticket carries its
    epoch, the kernel flags the context's status word, and the next synchronising call returns WF_ERR_DEVICE_STATUS; the ring is reset,
    so the tree after that is right again."""
    from winterfell_amd._lib import WfError
    ctx, crypto, _, _ = wf
    lv = np.random.default_rng(77).integers(0, 256, (1 << 13, 32), dtype=np.uint8)
    want = oracle.merkle_build(0, lv, par=True)
    assert np.array_equal(crypto.MerkleTree.new(crypto.Blake3_256, lv).nodes, want)
    for garbage in (0xDEADBEEF, 3):              # a foreign epoch; the right epoch with arrivals already counted
        ctx.call("wf_debug_poke_tree_ticket", garbage)
        tree = crypto.MerkleTree.new(crypto.Blake3_256, lv)
        with pytest.raises(WfError) as ei:
            ctx.sync()
        assert ei.value.status == 11, ei.value
        assert ctx.lib.wf_last_device_status(ctx.handle) & 1
        del tree
        assert np.array_equal(crypto.MerkleTree.new(crypto.Blake3_256, lv).nodes, want)      # the ring was put back: clean again
        ctx.sync()


def test_wf_free_refuses_what_is_not_a_live_block_of_the_pool(wf):
    """wf_free used to pass an unknown pointer on to hipFree (another owner's memory pulled from under it); now it is an error."""
    import ctypes
    from winterfell_amd._lib import WfError
    ctx = wf[0]
    p = ctypes.c_void_p()
    ctx.call("wf_malloc", 4096, ctypes.byref(p))
    t = ctx.empty_u8(4096)
    with pytest.raises(WfError):
        ctx.call("wf_free", ctypes.c_void_p(t.data_ptr()))        # a torch tensor's memory
    ctx.call("wf_free", p)
    with pytest.raises(WfError):
        ctx.call("wf_free", p)                                     # double free
    assert int(t.sum()) >= 0                                       # the tensor is still there


def test_merkle_errors(wf):
    ctx, crypto, _, _ = wf
    with pytest.raises(crypto.MerkleTreeError, match="TooFewLeaves"):
        crypto.MerkleTree.new(crypto.Blake3_256, np.zeros((1, 32), dtype=np.uint8))      # merkle/mod.rs:117
    with pytest.raises(crypto.MerkleTreeError, match="NotPowerOfTwo"):
        crypto.MerkleTree.new(crypto.Blake3_256, np.zeros((6, 32), dtype=np.uint8))      # merkle/mod.rs:120


def test_fib_trace_lde_fixture(wf, oracle, golden):
    """prover/src/trace/trace_lde/default/tests.rs:22-106 through DefaultTraceLde."""
    ctx, crypto, prover, fields = wf
    c0, c1 = golden["reference"]["fib_trace_col0"], golden["reference"]["fib_trace_col1"]
    trace = np.stack([fields.from_ints(c0), fields.from_ints(c1)])
    domain = prover.StarkDomain(8, 8)
    lde, polys = prover.DefaultTraceLde.new(crypto.Blake3_256, prover.ColMatrix(trace), domain)
    o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(0, trace, 8, fields.new(7))
    assert np.array_equal(polys.to_host(), o_polys)
    assert np.array_equal(lde.main_segment_lde.to_host(), o_lde)
    assert np.array_equal(lde.main_segment_oracles.nodes, o_nodes)
    assert np.array_equal(lde.get_main_trace_commitment(), o_nodes[1])
    assert lde.trace_len() == 64 and lde.blowup() == 8
    cur, nxt = lde.read_main_trace_frame_into(60)
    assert np.array_equal(cur, o_lde[60, :2]) and np.array_equal(nxt, o_lde[4, :2])   # wraps around
    (rows, (leaves, proof)), = lde.query([3, 17, 40])
    assert np.array_equal(rows, o_lde[[3, 17, 40], :2])
    assert np.array_equal(np.stack(leaves), o_leaves[[3, 17, 40]])
    with pytest.raises(AssertionError, match="number of rows"):
        lde.set_aux_trace(prover.ColMatrix(np.zeros((1, 16), dtype=np.uint64)), prover.StarkDomain(16, 8))


@pytest.mark.parametrize("hname,c,log_n,blowup,parts", [
    ("Blake3_256", 64, 8, 8, 1),     # prover/src/matrix/tests.rs shape: 64 polys, n=256, blowup 8
    ("Blake3_256", 4, 12, 8, 1),     # rescue-like 4 columns
    ("Rp64_256", 2, 11, 8, 1),       # fib_small-like 2 columns + Rp64_256
    ("Blake3_256", 20, 10, 4, 4),    # partitioned commitment
    ("Rp64_256", 20, 9, 2, 4),
    ("Blake3_256", 96, 9, 8, 1),     # row_matrix bench width 96 (768-byte rows)
    ("Blake3_256", 150, 6, 2, 1),    # > 1 BLAKE3 chunk per row
    ("Rp64_256", 3, 13, 8, 1),
    ("Blake3_256", 9, 16, 8, 1),     # 2 column groups, 3-pass NTT on the LDE domain
    ("Blake3_256", 3, 8, 8, 4),      # partition size 4 (hash_rate floor) ABOVE the 3 columns: merge_many over one digest
    ("Rp64_256", 3, 7, 8, 4),
    ("Blake3_256", 4, 8, 8, 2),      # partition size == column count: the plain row hash (row_matrix.rs:193)
])
def test_build_trace_commitment_vs_oracle(wf, oracle, hname, c, log_n, blowup, parts):
    ctx, crypto, prover, fields = wf
    hasher = getattr(crypto, hname)
    n = 1 << log_n
    trace = oracle.f64_from_int(rand_field(c * 1000 + log_n, n * c)).reshape(c, n)
    po = prover.PartitionOptions(parts, 4)
    lde, tree, polys = prover.build_trace_commitment(hasher, prover.ColMatrix(trace), prover.StarkDomain(n, blowup), po)
    o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(_hid(crypto, hasher), trace, blowup, fields.new(7),
                                                                      num_partitions=parts, hash_rate=4, par=True)
    assert np.array_equal(polys.to_host(), o_polys), "polys"
    assert np.array_equal(lde.to_host(), o_lde), "lde"
    assert np.array_equal(tree.leaves, o_leaves), "leaves"
    assert np.array_equal(tree.nodes, o_nodes), "nodes"


@pytest.mark.parametrize("D", [2, 3])
def test_aux_segment_extension_field(wf, oracle, D):
    ctx, crypto, prover, fields = wf
    n, c, b = 1 << 9, 3, 8
    main = oracle.f64_from_int(rand_field(1, n * 2)).reshape(2, n)
    aux = oracle.f64_from_int(rand_field(D, n * c * D)).reshape(c, n * D)
    domain = prover.StarkDomain(n, b)
    lde, _ = prover.DefaultTraceLde.new(crypto.Blake3_256, prover.ColMatrix(main), domain)
    polys, root = lde.set_aux_trace(prover.ColMatrix(aux, ext_degree=D), domain)
    o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(0, aux, b, fields.new(7), D=D)
    assert np.array_equal(polys.to_host(), o_polys)
    assert np.array_equal(lde.aux_segment_lde.to_host(), o_lde)
    assert np.array_equal(root, o_nodes[1])
    with pytest.raises(AssertionError, match="already been added"):
        lde.set_aux_trace(prover.ColMatrix(aux, ext_degree=D), domain)


def test_full_size_lde_commit_properties(wf, oracle):
    """BASELINE config 3b shape (2^20 rows, blowup 8): size-independent properties instead of a full CPU run."""
    ctx, crypto, prover, fields = wf
    import torch
    n, c, b = 1 << 20, 4, 8
    trace = oracle.f64_from_int(rand_field(77, n * c)).reshape(c, n)
    lde, tree, polys = prover.build_trace_commitment(crypto.Blake3_256, prover.ColMatrix(trace), prover.StarkDomain(n, b))
    N = n * b
    # every blowup-th LDE row is... not the trace (coset); instead check rows by Horner at a few positions
    hp = polys.to_host()
    g = oracle.f64_root_of_unity(23)
    pos = [0, 1, 7, 8, 12345, N // 2 + 3, N - 1]
    rows = lde.rows(pos)
    for r, k in zip(rows, pos):
        x = oracle.f64_mul(fields.new(7), oracle.f64_exp(g, k))
        for col in range(c):
            assert r[col] == oracle.poly_eval(hp[col], x), (k, col)
    # polys interpolate the trace: spot-check evaluations over the trace domain
    w = oracle.f64_root_of_unity(20)
    for i in (0, 5, n - 1):
        for col in (0, 3):
            assert oracle.poly_eval(hp[col], oracle.f64_exp(w, i)) == trace[col, i]
    # leaves / nodes: spot-check against the oracle hasher, and the root path of one leaf
    leaves, nodes = tree.leaves, tree.nodes
    for k in pos:
        assert np.array_equal(leaves[k], oracle.hash_elements(0, lde.rows([k])[0]))
    leaf, proof = tree.prove(12345)
    crypto.MerkleTree.verify(crypto.Blake3_256, tree.root(), 12345, leaf, proof)
    assert not nodes[0].any()


@pytest.mark.parametrize("hname", ["Blake3_256", "Rp64_256"])
def test_full_size_lde_commit_output_for_output(wf, oracle, hname):
    """BASELINE configs[2] shape at full size (2^20 rows x 4 f64 columns, blowup 8), word for word: trace polynomials, the
    whole 2^23-row LDE matrix, every leaf, every Merkle node and the root of wf_build_trace_commitment against the CPU
    oracle's restatement of DefaultTraceLde::new (prover/src/trace/trace_lde/default/mod.rs:245-282, concurrent variants)."""
    ctx, crypto, prover, fields = wf
    hasher = getattr(crypto, hname)
    n, c, b = 1 << 20, 4, 8
    trace = oracle.f64_from_int(rand_field(4242, n * c)).reshape(c, n)
    lde, tree, polys = prover.build_trace_commitment(hasher, prover.ColMatrix(trace), prover.StarkDomain(n, b))
    o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(_hid(crypto, hasher), trace, b, fields.new(7), par=True)
    assert np.array_equal(polys.to_host(), o_polys), "polys"
    got = lde.to_host()
    assert got.shape == o_lde.shape and np.array_equal(got, o_lde), "lde: %d words differ" % int(np.count_nonzero(got != o_lde))
    assert np.array_equal(tree.leaves, o_leaves), "leaves"
    assert np.array_equal(tree.nodes, o_nodes), "nodes"
    assert np.array_equal(tree.root(), o_nodes[1])


@pytest.mark.parametrize("hname,world", [("Blake3_256", 4), ("Rp64_256", 2), ("Blake3_256", 8)])
def test_column_sharded_commitment_emulated_on_one_gpu(wf, oracle, hname, world):
    """G logical column shards on one GPU (same kernels, same shard math as the multi-process path)
    must reproduce the single-device commitment built with PartitionOptions(G, .) bit for bit."""
    ctx, crypto, prover, fields = wf
    from winterfell_amd import parallel
    hasher = getattr(crypto, hname)
    hid = _hid(crypto, hasher)
    n, blowup, c = 1 << 9, 8, world * 4
    trace = oracle.f64_from_int(rand_field(world, n * c)).reshape(c, n)
    parts = parallel.column_partitions(c, world, 1, 1)
    assert len(parts) == world
    shards = [prover.ColMatrix(np.ascontiguousarray(trace[c0:c1])) for c0, c1 in parts]
    res = parallel.emulated_sharded_commit(parallel.HipBackend(hasher, ctx), shards, prover.StarkDomain(n, blowup))
    o_polys, o_lde, o_leaves, o_nodes = oracle.build_trace_commitment(hid, trace, blowup, fields.new(7), num_partitions=world,
                                                                      hash_rate=1, par=True)
    N = n * blowup
    per = N // world
    for r, (leaves, nodes) in enumerate(res["per_rank"]):
        assert np.array_equal(ctx.to_host(leaves), o_leaves[r * per:(r + 1) * per])
    full = parallel.assemble_nodes(world, N, [ctx.to_host(nd) for _, nd in res["per_rank"]], ctx.to_host(res["top"]))
    assert np.array_equal(full, o_nodes)
    assert np.array_equal(ctx.to_host(res["root"]), o_nodes[1])
    # and the single-device library call with the same PartitionOptions agrees too
    _, tree, _ = prover.build_trace_commitment(hasher, prover.ColMatrix(trace), prover.StarkDomain(n, blowup),
                                               prover.PartitionOptions(world, 1))
    assert np.array_equal(tree.nodes, o_nodes)


@pytest.mark.parametrize("D,num_cols,ce_blowup", [(1, 2, 2), (2, 4, 4), (3, 8, 8), (1, 1, 2)])
def test_constraint_commitment_vs_oracle(wf, oracle, D, num_cols, ce_blowup):
    """build_constraint_commitment (prover/src/constraints/commitment/default.rs:109-150): CompositionPoly::new
    (interpolate over the ce coset, segment) + LDE + commit, vs the oracle composed from its restated pieces."""
    ctx, crypto, prover, fields = wf
    n, blowup = 1 << 9, 8
    ce_n = n * ce_blowup
    comp_trace = oracle.f64_from_int(rand_field(D * 7 + num_cols, ce_n * D))
    domain = prover.StarkDomain(n, blowup)
    cc, poly = prover.build_constraint_commitment(crypto.Blake3_256, comp_trace.copy(), num_cols, domain, ext_degree=D)
    coeffs = oracle.interpolate_poly_with_offset(comp_trace, fields.new(7), D=D)
    cols = coeffs[: num_cols * n * D].reshape(num_cols, n * D)
    assert np.array_equal(poly.data.to_host(), cols)
    # polys -> LDE + commit (no interpolation): oracle pieces
    o_lde = oracle.evaluate_polys_over(cols, blowup, fields.new(7), D=D)
    o_leaves = oracle.hash_rows(0, o_lde, num_cols * D, D=D)
    o_nodes = oracle.merkle_build(0, o_leaves)
    assert np.array_equal(cc.evaluations.to_host(), o_lde)
    assert np.array_equal(cc.vector_commitment.nodes, o_nodes) and np.array_equal(cc.commitment(), o_nodes[1])
    rows, (leaves, proof) = cc.query([5, 100])
    assert np.array_equal(rows, o_lde[[5, 100], : num_cols * D])
    with pytest.raises(AssertionError, match="trace length must be smaller"):
        prover.CompositionPoly.new(comp_trace[: n * D], domain, 1, ext_degree=D)


@pytest.mark.parametrize("world,hname,fname,c,log_n,blowup", [(2, "Blake3_256", "f64", 5, 8, 8), (8, "Blake3_256", "f64", 12, 7, 8),
                                                                (4, "Rp64_256", "f64", 3, 6, 4), (8, "Blake3_256", "f128", 4, 6, 8)])
def test_strided_sharding_emulation_equals_default_commitment(wf, oracle, world, hname, fname, c, log_n, blowup):
    """SURVEY 8e Alternative B on one device: G logical ranks each run interpolate + coset LDE (blowup b/G, offset s*g^k) +
    plain row hashes through the kernels; after the leaf exchange the tree is the default unpartitioned commitment."""
    ctx, crypto, prover, fields = wf[0], wf[1], wf[2], wf[3]
    from winterfell_amd import parallel
    fld, ofld = {"f64": (fields.f64, oracle.f64t), "f128": (fields.f128, oracle.f128)}[fname]
    hasher = getattr(crypto, hname)
    hid = 0 if hname == "Blake3_256" else 1
    n = 1 << log_n
    rng = np.random.default_rng(world + c)
    vals = [int(a) * int(b) % fld.M for a, b in zip(rng.integers(1, 2**62, c * n), rng.integers(1, 2**62, c * n))]
    trace = fld.pack([fld.new(v) for v in vals]).reshape(c, -1)
    res = parallel.emulated_strided_commit(parallel.HipStridedBackend(hasher, fld, ctx), prover.ColMatrix(trace, 1, ctx, fld), n, blowup,
                                           fld.GENERATOR, fld, world)
    o = ofld.build_trace_commitment(hid, trace, blowup, fld.new(fld.GENERATOR))
    N = n * blowup
    assert np.array_equal(ctx.to_host(res["root"]), o[3][1])
    for k in range(world):
        assert np.array_equal(ctx.to_host(res["shards"][k][1].data), o[1][k::world])
    full = parallel.assemble_nodes(world, N, [ctx.to_host(nd) for _, nd in res["per_rank"]], ctx.to_host(res["top"]))
    assert np.array_equal(full, o[3])


@pytest.mark.parametrize("hname,c,log_n,D", [("Blake3_256", 3, 9, 1), ("Rp64_256", 5, 6, 1), ("Blake3_256", 2, 7, 2), ("Blake3_256", 70, 10, 1),
                                               ("Sha3_256", 9, 1, 3), ("Blake3_256", 1, 12, 1)])
def test_colmatrix_evaluate_columns_over_and_commit_to_rows(wf, oracle, hname, c, log_n, D):
    """ColMatrix::evaluate_columns_over (col_matrix.rs:230-243) = fft::evaluate_poly_with_offset per column, and
    ColMatrix::commit_to_rows (col_matrix.rs:262-286) = tree over hash_elements(row) of a COLUMN-major matrix."""
    ctx, crypto, prover, fields = wf
    hasher = getattr(crypto, hname)
    hid = {"Blake3_256": 0, "Rp64_256": 1, "Sha3_256": 2}[hname]
    n, blowup = 1 << log_n, 4
    cols = np.stack([oracle.f64_from_int(rand_field(31 * c + k, n * D)) for k in range(c)])
    m = prover.ColMatrix(cols.copy(), D, ctx)
    dom = prover.StarkDomain(n, blowup)
    ev = m.evaluate_columns_over(dom)
    assert ev.num_cols() == c and ev.num_rows() == n * blowup and ev.ext_degree == D
    got = ev.to_host()
    for k in range(c):
        assert np.array_equal(got[k], oracle.evaluate_poly_with_offset(cols[k], oracle.f64_new(7), blowup, D=D)), k
    # row hashes of the column-major matrix: row r = [col_0[r], col_1[r], ...] (D words each)
    tree = m.commit_to_rows(hasher)
    rows = cols.reshape(c, n, D).transpose(1, 0, 2).reshape(n, c * D)
    want_leaves = np.stack([oracle.hash_elements(hid, rows[r]) for r in range(n)])
    assert np.array_equal(tree.leaves, want_leaves)
    assert np.array_equal(tree.root(), oracle.merkle_build(hid, want_leaves)[1])
    rm = prover.RowMatrix.evaluate_polys(m, blowup)                      # row_matrix.rs:57-74: offset = GENERATOR = 7
    assert np.array_equal(rm.get(c - 1, 5), got[c - 1][5 * D:6 * D]) and np.array_equal(rm.row(3)[:D], got[0][3 * D:4 * D])
    # accessors (col_matrix.rs:85-165)
    assert np.array_equal(m.get(c - 1, n - 1), cols[c - 1][(n - 1) * D:])
    assert np.array_equal(m.read_row_into(1), rows[1])
    extra = oracle.f64_from_int(rand_field(5, n * D))
    m.merge_column(extra)
    assert m.num_cols() == c + 1 and np.array_equal(m.get_column(c), extra)
    assert np.array_equal(ctx.to_host(m.remove_column(0)), cols[0]) and m.num_cols() == c


def test_colmatrix_commit_to_rows_f128(wf, oracle):
    ctx, crypto, prover, fields = wf
    f, of = fields.f128, oracle.f128
    n, c = 1 << 8, 6
    rng = np.random.default_rng(77)
    cols = np.stack([f.pack([int(v) for v in rng.integers(0, 1 << 62, n)]) for _ in range(c)])    # small canonical values
    m = prover.ColMatrix(cols.copy(), 1, ctx, f)
    tree = m.commit_to_rows(crypto.Blake3_256)
    rows = cols.reshape(c, n, 2).transpose(1, 0, 2).reshape(n, c * 2)
    want = np.stack([np.frombuffer(oracle.blake3(rows[r].tobytes()), dtype=np.uint8) for r in range(n)])
    assert np.array_equal(tree.leaves, want)
    ev = m.evaluate_columns_over(prover.StarkDomain(n, 2, field=f)).to_host()
    for k in range(c):
        assert np.array_equal(ev[k], of.evaluate_poly_with_offset(cols[k], int(f.GENERATOR), 2))


@pytest.mark.parametrize("hname,hid", [("Blake3_256", 0), ("Rp64_256", 1), ("Sha3_256", 2), ("RpJive64_256", 3), ("Blake3_192", 5)])
def test_merge_many_on_every_hasher(wf, oracle, hname, hid):
    """Hasher::merge_many (crypto/src/hash/mod.rs:39-41): single and batched calls vs the oracle; merge == merge_many for two
    digests wherever the reference's tests say so (blake/tests.rs, rescue tests merge_vs_merge_many — not the Jive hasher,
    whose merge is the compression mode)."""
    ctx, crypto, prover, fields = wf
    hasher = getattr(crypto, hname)
    rng = np.random.default_rng(hid)
    elems = oracle.f64_from_int(rand_field(60 + hid, 5 * 3 * 4)).reshape(5, 3, 4)
    digs = np.ascontiguousarray(elems).view(np.uint8).reshape(5, 3, 32).copy()
    if hname == "Blake3_192":
        digs[:, :, 24:] = 0
    got = hasher.merge_many(digs)
    for i in range(5):
        assert np.array_equal(got[i], oracle.merge_many(hid, digs[i])), i
        assert np.array_equal(hasher.merge_many(digs[i]), got[i])
    if hname != "RpJive64_256":
        assert np.array_equal(hasher.merge_many(digs[0][:2]), hasher.merge(digs[0][:2]))


def _chunk_elements(data, modulus):
    """test-side restatement of the Rescue byte -> element rule (rp64_256/mod.rs:123-160): canonical integers"""
    out = []
    n = -(-len(data) // 7)
    for k in range(n):
        ch = data[7 * k:7 * k + 7]
        v = int.from_bytes(ch, "little") + ((1 << (8 * len(ch))) if k == n - 1 else 0)
        out.append(v % modulus)
    return out


def test_hasher_hash_bytes(wf, oracle):
    """Hasher::hash(&[u8]) (crypto/src/hash/mod.rs:33-35) on every hasher.  Byte hashers: every length 0..300 (block, chunk and
    rate-block boundaries; partial words) against the LLVM-pinned BLAKE3 restatement / hashlib's SHA3; Rescue hashers: the
    sponge over the 7-byte chunks, plus the reference's hash_padding properties (rescue tests.rs:162-182)."""
    import hashlib
    ctx, crypto, prover, fields = wf
    rng = np.random.default_rng(21)
    for n in list(range(0, 80)) + [127, 128, 129, 135, 136, 137, 271, 272, 273, 300, 1023, 1024, 1025, 2048, 3000]:
        msg = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert crypto.Blake3_256.hash(msg).tobytes() == oracle.blake3(msg), n
        assert crypto.Blake3_192.hash(msg).tobytes() == oracle.blake3(msg)[:24] + bytes(8), n
        assert crypto.Sha3_256.hash(msg).tobytes() == hashlib.sha3_256(msg).digest(), n
    batch = [rng.integers(0, 256, 41, dtype=np.uint8).tobytes() for _ in range(70)]
    got = crypto.Blake3_256.hash(batch)
    assert all(got[i].tobytes() == oracle.blake3(batch[i]) for i in range(70))
    assert crypto.Blake3_256.hash(bytes([1])).tobytes() == oracle.blake3(bytes([1]))       # crypto/src/merkle/mod.rs:72 doc example
    for hasher, hid, f in ((crypto.Rp64_256, 1, fields.f64), (crypto.RpJive64_256, 3, fields.f64), (crypto.Rp62_248, 4, fields.f62)):
        for n in (0, 1, 3, 6, 7, 8, 13, 14, 28, 29, 55, 56):                                  # up to 8 chunks: one rule for all three
            msg = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            want = oracle.hash_elements(hid, f.pack([f.new(v) for v in _chunk_elements(msg, f.M)])) if hid != 4 else None
            got = hasher.hash(msg)
            if hid != 4:
                assert np.array_equal(got, want), (hid, n)
            else:
                assert np.array_equal(got, hasher.hash_elements(f.pack([f.new(v) for v in _chunk_elements(msg, f.M)])))
        pairs = [([1, 2, 3], [1, 2, 3, 0]), ([1, 2, 3, 4, 5, 6], [1, 2, 3, 4, 5, 6, 0]), ([1, 2, 3, 4, 5, 6, 7], [1, 2, 3, 4, 5, 6, 7, 0]),
                 ([1, 2, 3, 4, 5, 6, 7, 0, 0], [1, 2, 3, 4, 5, 6, 7, 0, 0, 0, 0])]
        for a, b in pairs:
            assert not np.array_equal(hasher.hash(bytes(a)), hasher.hash(bytes(b)))
    # longer strings: Rp64 / RpJive keep the chunk-index rule, Rp62_248 the position-based one (rp62_248/mod.rs:119)
    msg = rng.integers(0, 256, 100, dtype=np.uint8).tobytes()
    assert np.array_equal(crypto.Rp64_256.hash(msg), oracle.hash_elements(1, fields.f64.pack([fields.f64.new(v) for v in _chunk_elements(msg, fields.f64.M)])))
    with pytest.raises(ValueError):
        crypto.Rp62_248.hash(msg)                                                            # 100 = 14 * 7 + 2: the reference panics
    m70 = rng.integers(0, 256, 70, dtype=np.uint8).tobytes()
    f62 = fields.f62
    verbatim = [int.from_bytes(m70[7 * k:7 * k + 7], "little") for k in range(10)]
    assert np.array_equal(crypto.Rp62_248.hash(m70), crypto.Rp62_248.hash_elements(f62.pack([f62.new(v) for v in verbatim])))


@pytest.mark.parametrize("hname,hid", [("Blake3_256", 0), ("Rp64_256", 1), ("Sha3_256", 2), ("RpJive64_256", 3), ("Blake3_192", 5)])
@pytest.mark.parametrize("rows,cols,width,parts,rate,D", [(1000, 24, 24, 3, 1, 1), (37, 20, 24, 2, 8, 2), (65, 9, 16, 1, 1, 1), (4097, 40, 40, 16, 1, 1),
                                                           (1, 130, 136, 1, 1, 1), (129, 17, 17, 1, 1, 1),
                                                           # partition size ABOVE the column count (hash_rate floor): still
                                                           # merge_many over the single digest (row_matrix.rs:193; the case
                                                           # PartitionOptions::new(4, 8) with 7 columns of the reference's tests)
                                                           (300, 7, 8, 4, 8, 1), (70, 3, 8, 4, 8, 1), (33, 6, 8, 2, 16, 2),
                                                           # hash_rate 256 is stored `as u8` = 0 by the reference
                                                           (40, 24, 24, 4, 256, 1)])
def test_hash_rows_ragged_shapes_every_kernel_path(wf, oracle, hname, hid, rows, cols, width, parts, rate, D):
    """wf_hash_rows on row counts that are not multiples of the 64-row wavefront / 16-lane group, padded row widths, partitions
    with a short last chunk and a hash_rate floor — the wave-cooperative wide-row kernels (BLAKE3 / SHA3), the lane-cooperative
    Rescue kernels (few rows) and the per-lane kernels all against the oracle's hash_rows (row_matrix.rs:184-228)."""
    ctx, crypto, prover, fields = wf
    hasher = getattr(crypto, hname)
    data = np.zeros((rows, width), dtype=np.uint64)
    data[:, :cols] = oracle.f64_from_int(rand_field(rows + cols, rows * cols)).reshape(rows, cols)
    data[:, cols:] = np.uint64(0xDEAD)                     # padding must never be hashed
    if cols % D:
        pytest.skip("columns must hold whole extension elements")
    m = prover.RowMatrix(ctx.to_device(data), width, cols, D, ctx, fields.f64)
    got = ctx.to_host(m.hash_rows(hasher, prover.PartitionOptions(parts, rate)))
    want = oracle.hash_rows(hid, data, cols, D=D, num_partitions=parts, hash_rate=rate)
    assert np.array_equal(got, want)
