"""ORACLE — test infrastructure only.

ctypes loader for ``oracle/liboracle.so`` (the C restatement of the reference's CPU algorithms).
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package, and only as the checker; nothing under ``winterfell_amd/`` imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

H_BLAKE3_F64 = 0
H_RP64 = 1

M = 0xFFFFFFFF00000001


def build(force=False):
    """Compile the oracle with gcc (build, not use)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


_u64 = ctypes.c_uint64
_p = ctypes.c_void_p


def _declare(l):
    for name in ("or_f64_mul1", "or_f64_add1", "or_f64_sub1", "or_f64_exp1"):
        getattr(l, name).restype = _u64
        getattr(l, name).argtypes = [_u64, _u64]
    for name in ("or_f64_inv1", "or_f64_new1", "or_f64_as_int1"):
        getattr(l, name).restype = _u64
        getattr(l, name).argtypes = [_u64]
    l.or_f64_root_of_unity1.restype = _u64
    l.or_f64_root_of_unity1.argtypes = [ctypes.c_uint]
    l.or_permute_index.restype = _u64
    l.or_permute_index.argtypes = [_u64, _u64]
    l.or_f64_poly_eval.restype = _u64
    l.or_f64_poly_eval.argtypes = [_p, _u64, _u64]
    l.or_row_width.restype = _u64
    l.or_row_width.argtypes = [_u64]
    l.or_partition_size.restype = _u64
    l.or_partition_size.argtypes = [_u64, _u64, ctypes.c_uint, _u64]
    l.or_num_partitions.restype = _u64
    l.or_num_partitions.argtypes = [_u64, _u64, ctypes.c_uint, _u64]
    l.or_merkle_build.restype = ctypes.c_int
    l.or_merkle_build_par.restype = ctypes.c_int
    l.or_build_trace_commitment.restype = ctypes.c_int


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _u64arr(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


# ---- field helpers --------------------------------------------------------------------------------
def set_num_threads(n):
    """OpenMP team size for oracle calls made from the calling thread (test plumbing)."""
    lib().or_set_num_threads(ctypes.c_int(n))


def f64_from_int(vals):
    a = _u64arr(vals)
    out = np.empty_like(a)
    lib().or_f64_from_int(_ptr(a), _ptr(out), _u64(a.size))
    return out


def f64_to_int(vals):
    a = _u64arr(vals)
    out = np.empty_like(a)
    lib().or_f64_to_int(_ptr(a), _ptr(out), _u64(a.size))
    return out


def f64_new(v):
    return lib().or_f64_new1(_u64(v % (1 << 64)))


def f64_as_int(v):
    return lib().or_f64_as_int1(_u64(v))


def f64_mul(a, b):
    return lib().or_f64_mul1(_u64(a), _u64(b))


def f64_add(a, b):
    return lib().or_f64_add1(_u64(a), _u64(b))


def f64_sub(a, b):
    return lib().or_f64_sub1(_u64(a), _u64(b))


def f64_inv(a):
    return lib().or_f64_inv1(_u64(a))


def f64_exp(a, e):
    return lib().or_f64_exp1(_u64(a), _u64(e))


def f64_root_of_unity(log_n):
    return lib().or_f64_root_of_unity1(log_n)


def f64_ext_mul(D, a, b):
    a, b = _u64arr(a), _u64arr(b)
    out = np.empty(D, dtype=np.uint64)
    lib().or_f64_ext_mul(ctypes.c_uint(D), _ptr(a), _ptr(b), _ptr(out))
    return out


# ---- fft ------------------------------------------------------------------------------------------
def get_twiddles(n):
    out = np.empty(n // 2, dtype=np.uint64)
    lib().or_f64_get_twiddles(_ptr(out), _u64(n))
    return out


def get_inv_twiddles(n):
    out = np.empty(n // 2, dtype=np.uint64)
    lib().or_f64_get_inv_twiddles(_ptr(out), _u64(n))
    return out


def permute_index(size, index):
    return lib().or_permute_index(_u64(size), _u64(index))


def evaluate_poly(p, D=1, par=False, twiddles=None, inplace=False):
    """fft::evaluate_poly — natural-order coefficients -> natural-order evaluations (a copy unless inplace).
    `twiddles` = get_twiddles(n) computed by the caller (the reference computes them once per domain and its benches keep
    them out of the timed body, math/benches/fft.rs)."""
    v = _u64arr(p) if inplace else _u64arr(p).copy()
    n = v.size // D
    tw = get_twiddles(n) if twiddles is None else twiddles
    fn = lib().or_f64_evaluate_poly_par if par else lib().or_f64_evaluate_poly
    fn(_ptr(v), _u64(n), ctypes.c_uint(D), _ptr(tw))
    return v


def interpolate_poly(ev, D=1, par=False, twiddles=None, inplace=False):
    v = _u64arr(ev) if inplace else _u64arr(ev).copy()
    n = v.size // D
    tw = get_inv_twiddles(n) if twiddles is None else twiddles
    fn = lib().or_f64_interpolate_poly_par if par else lib().or_f64_interpolate_poly
    fn(_ptr(v), _u64(n), ctypes.c_uint(D), _ptr(tw))
    return v


def evaluate_poly_with_offset(p, domain_offset, blowup, D=1, par=False):
    v = _u64arr(p)
    n = v.size // D
    tw = get_twiddles(n)
    out = np.empty(n * blowup * D, dtype=np.uint64)
    fn = lib().or_f64_evaluate_poly_with_offset_par if par else lib().or_f64_evaluate_poly_with_offset
    fn(_ptr(v), _u64(n), ctypes.c_uint(D), _ptr(tw), _u64(domain_offset), _u64(blowup), _ptr(out))
    return out


def interpolate_poly_with_offset(ev, domain_offset, D=1):
    v = _u64arr(ev).copy()
    n = v.size // D
    tw = get_inv_twiddles(n)
    lib().or_f64_interpolate_poly_with_offset(_ptr(v), _u64(n), ctypes.c_uint(D), _ptr(tw), _u64(domain_offset))
    return v


def poly_eval(p, x):
    v = _u64arr(p)
    return lib().or_f64_poly_eval(_ptr(v), _u64(v.size), _u64(x))


# ---- hashing --------------------------------------------------------------------------------------
def blake3(data: bytes) -> bytes:
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    out = np.empty(32, dtype=np.uint8)
    lib().or_blake3_hash(_ptr(buf), _u64(len(data)), _ptr(out))
    return out.tobytes()


def rp64_apply_permutation(state):
    s = _u64arr(state).copy()
    lib().or_rp64_apply_permutation(_ptr(s))
    return s


def rp64_mds(state, naive=False):
    s = _u64arr(state).copy()
    (lib().or_rp64_mds_naive if naive else lib().or_rp64_mds_freq)(_ptr(s))
    return s


def rp64_hash_bytes(data: bytes):
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    out = np.empty(4, dtype=np.uint64)
    lib().or_rp64_hash_bytes(_ptr(buf), _u64(len(data)), _ptr(out))
    return out


def hash_elements(hasher, elems):
    e = _u64arr(elems)
    out = np.empty(32, dtype=np.uint8)
    lib().or_hash_elements(ctypes.c_int(hasher), _ptr(e) if e.size else None, _u64(e.size), _ptr(out))
    return out


def merge(hasher, two):
    t = np.ascontiguousarray(two).view(np.uint8).reshape(-1)
    assert t.size == 64
    out = np.empty(32, dtype=np.uint8)
    lib().or_hash_merge(ctypes.c_int(hasher), _ptr(t), _ptr(out))
    return out


def merge_many(hasher, digests):
    t = np.ascontiguousarray(digests).view(np.uint8).reshape(-1)
    out = np.empty(32, dtype=np.uint8)
    lib().or_hash_merge_many(ctypes.c_int(hasher), _ptr(t), _u64(t.size // 32), _ptr(out))
    return out


def merge_with_int(hasher, seed, value):
    t = np.ascontiguousarray(seed).view(np.uint8).reshape(-1)
    out = np.empty(32, dtype=np.uint8)
    lib().or_hash_merge_with_int(ctypes.c_int(hasher), _ptr(t), _u64(value), _ptr(out))
    return out


def merkle_build(hasher, leaves, par=False, out=None):
    """leaves: (n, 32) uint8.  Returns nodes (n, 32) uint8 in the reference heap layout (root at [1]).  `out`: a caller-owned
    (already touched) nodes array to write into (bench.py's cpu_baseline times the build, not the page faults of a fresh buffer)."""
    lv = np.ascontiguousarray(leaves).view(np.uint8).reshape(-1, 32)
    nodes = np.empty_like(lv) if out is None else out
    assert nodes.shape == lv.shape and nodes.dtype == np.uint8 and nodes.flags.c_contiguous
    fn = lib().or_merkle_build_par if par else lib().or_merkle_build
    rc = fn(ctypes.c_int(hasher), _ptr(lv), _u64(lv.shape[0]), _ptr(nodes))
    if rc:
        raise ValueError({1: "TooFewLeaves", 2: "NumberOfLeavesNotPowerOfTwo"}[rc])
    return nodes


# ---- matrices / trace commitment ------------------------------------------------------------------
def row_width(base_cols):
    return lib().or_row_width(_u64(base_cols))


def partition_size(num_partitions, hash_rate, D, num_columns):
    return lib().or_partition_size(_u64(num_partitions), _u64(hash_rate), ctypes.c_uint(D), _u64(num_columns))


def interpolate_columns(cols, D=1, par=False):
    """cols: (c, n*D) uint64, column-major.  Returns coefficient matrix of the same shape."""
    v = _u64arr(cols).copy()
    c, nD = v.shape
    lib().or_interpolate_columns(_ptr(v), _u64(c), _u64(nD // D), ctypes.c_uint(D), ctypes.c_int(par))
    return v


def evaluate_polys_over(polys, blowup, domain_offset, D=1, par=False):
    v = _u64arr(polys)
    c, nD = v.shape
    n = nD // D
    rw = row_width(c * D)
    out = np.empty((n * blowup, rw), dtype=np.uint64)
    lib().or_evaluate_polys_over(_ptr(v), _u64(c), _u64(n), ctypes.c_uint(D), _u64(blowup), _u64(domain_offset),
                                 _ptr(out), ctypes.c_int(par))
    return out


def hash_rows(hasher, data, elements_per_row, D=1, num_partitions=1, hash_rate=1):
    v = _u64arr(data)
    N, rw = v.shape
    leaves = np.empty((N, 32), dtype=np.uint8)
    lib().or_hash_rows(ctypes.c_int(hasher), _ptr(v), _u64(N), _u64(rw), _u64(elements_per_row), ctypes.c_uint(D),
                       _u64(num_partitions), _u64(hash_rate), _ptr(leaves))
    return leaves


def build_trace_commitment(hasher, trace, blowup, domain_offset, D=1, num_partitions=1, hash_rate=1, par=False, out=None):
    """trace: (c, n*D) uint64 column-major evaluations.  Returns (polys, lde, leaves, nodes).  `out` = (polys, lde, leaves, nodes)
    of a previous call: the result arrays are reused (already page-faulted) instead of freshly allocated."""
    c, nD = _u64arr(trace).shape
    n = nD // D
    N = n * blowup
    rw = row_width(c * D)
    if out is None:
        polys = _u64arr(trace).copy()
        lde = np.empty((N, rw), dtype=np.uint64)
        leaves = np.empty((N, 32), dtype=np.uint8)
        nodes = np.empty((N, 32), dtype=np.uint8)
    else:
        polys, lde, leaves, nodes = out
        assert polys.shape == (c, nD) and lde.shape == (N, rw) and leaves.shape == (N, 32) and nodes.shape == (N, 32)
        polys[...] = _u64arr(trace)
    rc = lib().or_build_trace_commitment(ctypes.c_int(hasher), _ptr(polys), _u64(c), _u64(n), ctypes.c_uint(D),
                                         _u64(blowup), _u64(domain_offset), _u64(num_partitions), _u64(hash_rate),
                                         _ptr(lde), _ptr(leaves), _ptr(nodes), ctypes.c_int(par))
    assert rc == 0
    return polys, lde, leaves, nodes


# ---- FRI ------------------------------------------------------------------------------------------------
def transpose_slice(src, N, D=1):
    v = _u64arr(src)
    out = np.empty_like(v)
    lib().or_transpose_slice(_ptr(v), _u64(v.size // D), ctypes.c_uint(D), _u64(N), _ptr(out))
    return out


def fri_layer_commit(hasher, transposed, N, D=1):
    v = _u64arr(transposed)
    rows = v.size // (N * D)
    leaves = np.empty((rows, 32), dtype=np.uint8)
    nodes = np.empty((rows, 32), dtype=np.uint8)
    rc = lib().or_fri_layer_commit(ctypes.c_int(hasher), _ptr(v), _u64(rows), ctypes.c_uint(D), _u64(N), _ptr(leaves), _ptr(nodes))
    assert rc == 0
    return leaves, nodes


def apply_drp(transposed, N, domain_offset, alpha, D=1):
    v = _u64arr(transposed)
    rows = v.size // (N * D)
    a = _u64arr(alpha)
    out = np.empty(rows * D, dtype=np.uint64)
    lib().or_apply_drp(_ptr(v), _u64(rows), ctypes.c_uint(D), _u64(N), _u64(domain_offset), _ptr(a), _ptr(out))
    return out


def fri_build_layers_par(hasher, evals, N, blowup, remainder_max_degree, domain_offset, D=1):
    """FriProver::build_layers on all cores (the reference's `concurrent` feature) against a DefaultProverChannel.
    Returns (roots (layers + 1, 32), alphas (layers, D)); `evals` is copied."""
    v = _u64arr(evals).copy()
    length = v.size // D
    nl = fri_num_layers(length, N, blowup, remainder_max_degree)
    roots = np.empty((nl + 1, 32), dtype=np.uint8)
    alphas = np.empty((max(nl, 1), D), dtype=np.uint64)
    lib().or_fri_build_layers_par.restype = _u64
    got = lib().or_fri_build_layers_par(ctypes.c_int(hasher), _ptr(v), _u64(length), ctypes.c_uint(D), _u64(N), _u64(blowup),
                                        _u64(remainder_max_degree), _u64(domain_offset), _ptr(roots), _ptr(alphas))
    assert int(got) == nl
    return roots, alphas[:nl]


def apply_drp_rows(transposed, N, domain_size, row_start, domain_offset, alpha, D=1):
    """apply_drp for a contiguous range of rows of a layer with `domain_size` points (multi-GPU shard)."""
    v = _u64arr(transposed)
    rows = v.size // (N * D)
    a = _u64arr(alpha)
    out = np.empty(rows * D, dtype=np.uint64)
    lib().or_apply_drp_rows(_ptr(v), _u64(rows), ctypes.c_uint(D), _u64(N), _u64(domain_size), _u64(row_start), _u64(domain_offset),
                            _ptr(a), _ptr(out))
    return out


def fri_num_layers(domain_size, folding, blowup, remainder_max_degree):
    lib().or_fri_num_layers.restype = _u64
    return lib().or_fri_num_layers(_u64(domain_size), _u64(folding), _u64(blowup), _u64(remainder_max_degree))


def fri_remainder(hasher, evals, domain_offset, blowup, D=1):
    v = _u64arr(evals).copy()
    n = v.size // D
    rem = np.empty((n // blowup) * D, dtype=np.uint64)
    com = np.empty(32, dtype=np.uint8)
    lib().or_fri_remainder(ctypes.c_int(hasher), _ptr(v), _u64(n), ctypes.c_uint(D), _u64(domain_offset), _u64(blowup), _ptr(rem), _ptr(com))
    return rem, com


class RandomCoin:
    """DefaultRandomCoin (crypto/src/random/default.rs) over the oracle's hashers."""

    def __init__(self, hasher, seed_elems=()):
        lib().or_coin_sizeof.restype = _u64
        self._buf = ctypes.create_string_buffer(int(lib().or_coin_sizeof()))
        s = _u64arr(list(seed_elems)) if len(seed_elems) else np.zeros(1, dtype=np.uint64)
        lib().or_coin_new(self._buf, ctypes.c_int(hasher), _ptr(s), _u64(len(seed_elems)))

    def reseed(self, digest):
        d = np.ascontiguousarray(digest).view(np.uint8).reshape(-1)
        lib().or_coin_reseed(self._buf, _ptr(d))

    def draw(self, D=1):
        out = np.empty(D, dtype=np.uint64)
        rc = lib().or_coin_draw(self._buf, ctypes.c_uint(D), _ptr(out))
        assert rc == 0
        return out

    def seed(self):
        out = np.empty(32, dtype=np.uint8)
        lib().or_coin_seed(self._buf, _ptr(out))
        return out

    def check_leading_zeros(self, value):
        lib().or_coin_check_leading_zeros.restype = ctypes.c_uint32
        return int(lib().or_coin_check_leading_zeros(self._buf, _u64(value)))

    def grind(self, grinding_factor, limit=1 << 40):
        """ProverChannel::grind_query_seed, serial path (prover/src/channel.rs:169-175); 0 when nothing below limit."""
        lib().or_coin_grind.restype = _u64
        return int(lib().or_coin_grind(self._buf, ctypes.c_uint32(grinding_factor), _u64(limit)))

    def draw_integers(self, num_values, domain_size, nonce):
        assert domain_size & (domain_size - 1) == 0 and num_values < domain_size
        out = np.empty(num_values, dtype=np.uint64)
        lib().or_coin_draw_integers.restype = _u64
        n = int(lib().or_coin_draw_integers(self._buf, _u64(num_values), _u64(domain_size), _u64(nonce), _ptr(out)))
        assert n == num_values, "failed to draw enough integers"   # RandomCoinError::FailedToDrawIntegers
        return out


class ProverChannel:
    """fri::DefaultProverChannel (fri/src/prover/channel.rs:60-127): coin seeded with no elements."""

    def __init__(self, hasher, D=1):
        self.coin = RandomCoin(hasher, ())
        self.D = D
        self.commitments = []

    def commit_fri_layer(self, root):
        self.commitments.append(np.array(root, copy=True))
        self.coin.reseed(root)

    def draw_fri_alpha(self):
        return self.coin.draw(self.D)


# ---- field-generic template instantiations (field_tmpl.inc): f128 and the f64 cross-check copy -----------
class GenericField:
    """Elements are W little-endian u64 words (W = 2 for f128).  Arrays are uint64 with the word axis last/flattened."""

    def __init__(self, name, words, modulus):
        self.name, self.W, self.M = name, words, modulus

    def _fn(self, n):
        return getattr(lib(), "or_%s_%s" % (self.name, n))

    # scalars as python ints <-> W-word arrays (internal representation: canonical for f128, Montgomery for f64t)
    def pack(self, vals):
        vals = np.asarray(vals, dtype=object).reshape(-1)
        out = np.empty((len(vals), self.W), dtype=np.uint64)
        for k in range(self.W):
            out[:, k] = [(int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for v in vals]
        return out.reshape(-1)

    def unpack(self, arr):
        a = _u64arr(arr).reshape(-1, self.W)
        return [sum(int(a[i, k]) << (64 * k) for k in range(self.W)) for i in range(a.shape[0])]

    def _bin(self, name, a, b):
        out = np.empty(self.W, dtype=np.uint64)
        pa, pb = self.pack([a]), self.pack([b])      # keep the temporaries alive across the call
        self._fn(name)(_ptr(pa), _ptr(pb), _ptr(out))
        return self.unpack(out)[0]

    def mul(self, a, b):
        return self._bin("mul", a, b)

    def add(self, a, b):
        return self._bin("add", a, b)

    def sub(self, a, b):
        return self._bin("sub", a, b)

    def inv(self, a):
        out = np.empty(self.W, dtype=np.uint64)
        pa = self.pack([a])
        self._fn("inv")(_ptr(pa), _ptr(out))
        return self.unpack(out)[0]

    def exp(self, a, e):
        out = np.empty(self.W, dtype=np.uint64)
        pa = self.pack([a])
        self._fn("exp")(_ptr(pa), _u64(e), _ptr(out))
        return self.unpack(out)[0]

    def root_of_unity(self, log_n):
        out = np.empty(self.W, dtype=np.uint64)
        self._fn("root_of_unity")(ctypes.c_uint(log_n), _ptr(out))
        return self.unpack(out)[0]

    def ext_mul(self, D, a, b):
        out = np.empty(D * self.W, dtype=np.uint64)
        pa, pb = self.pack(a), self.pack(b)
        self._fn("ext_mul")(ctypes.c_uint(D), _ptr(pa), _ptr(pb), _ptr(out))
        return self.unpack(out)

    def get_twiddles(self, n, inverse=False):
        out = np.empty((n // 2) * self.W, dtype=np.uint64)
        self._fn("get_twiddles")(_ptr(out), _u64(n), ctypes.c_int(int(inverse)))
        return out

    def evaluate_poly(self, p, D=1):
        v = _u64arr(p).copy()
        self._fn("evaluate_poly")(_ptr(v), _u64(v.size // (D * self.W)), ctypes.c_uint(D))
        return v

    def interpolate_poly(self, ev, D=1):
        v = _u64arr(ev).copy()
        self._fn("interpolate_poly")(_ptr(v), _u64(v.size // (D * self.W)), ctypes.c_uint(D))
        return v

    def evaluate_poly_with_offset(self, p, offset, blowup, D=1):
        v = _u64arr(p)
        n = v.size // (D * self.W)
        out = np.empty(n * blowup * D * self.W, dtype=np.uint64)
        po = self.pack([offset])
        self._fn("evaluate_poly_with_offset")(_ptr(v), _u64(n), ctypes.c_uint(D), _ptr(po), _u64(blowup), _ptr(out))
        return out

    def interpolate_poly_with_offset(self, ev, offset, D=1):
        v = _u64arr(ev).copy()
        po = self.pack([offset])
        self._fn("interpolate_poly_with_offset")(_ptr(v), _u64(v.size // (D * self.W)), ctypes.c_uint(D), _ptr(po))
        return v

    def poly_eval(self, p, x):
        v = _u64arr(p)
        out = np.empty(self.W, dtype=np.uint64)
        px = self.pack([x])
        self._fn("poly_eval")(_ptr(v), _u64(v.size // self.W), _ptr(px), _ptr(out))
        return self.unpack(out)[0]

    def build_trace_commitment(self, hasher, trace, blowup, offset, D=1, num_partitions=1, hash_rate=1):
        """trace: (c, n*D*W) uint64.  Returns (polys, lde (N, row_width*W), leaves, nodes)."""
        polys = _u64arr(trace).copy()
        c = polys.shape[0]
        n = polys.shape[1] // (D * self.W)
        N, rw = n * blowup, row_width(c * D)
        lde = np.empty((N, rw * self.W), dtype=np.uint64)
        leaves, nodes = np.empty((N, 32), dtype=np.uint8), np.empty((N, 32), dtype=np.uint8)
        fn = self._fn("build_trace_commitment")
        fn.restype = ctypes.c_int
        po = self.pack([offset])
        rc = fn(ctypes.c_int(hasher), _ptr(polys), _u64(c), _u64(n), ctypes.c_uint(D), _u64(blowup), _ptr(po),
                _u64(num_partitions), _u64(hash_rate), _ptr(lde), _ptr(leaves), _ptr(nodes))
        assert rc == 0
        return polys, lde, leaves, nodes

    def hash_rows(self, hasher, rows, num_cols, D=1, num_partitions=1, hash_rate=1):
        """row digests of a row-major matrix (N, row_width*W) whose rows hold num_cols elements of degree D: commit_to_rows'
        leaf rule, partitioned or not (prover/src/matrix/row_matrix.rs:184-228)."""
        v = _u64arr(rows)
        N, rw = v.shape[0], v.shape[1] // self.W
        leaves = np.empty((N, 32), dtype=np.uint8)
        self._fn("hash_rows")(ctypes.c_int(hasher), _ptr(v), _u64(N), _u64(rw), _u64(num_cols), ctypes.c_uint(D), _u64(num_partitions),
                              _u64(hash_rate), _ptr(leaves))
        return leaves

    def transpose_slice(self, src, N, D=1):
        v = _u64arr(src)
        out = np.empty_like(v)
        self._fn("transpose_slice")(_ptr(v), _u64(v.size // (D * self.W)), ctypes.c_uint(D), _u64(N), _ptr(out))
        return out

    def fri_layer_commit(self, hasher, transposed, N, D=1):
        v = _u64arr(transposed)
        rows = v.size // (N * D * self.W)
        leaves, nodes = np.empty((rows, 32), dtype=np.uint8), np.empty((rows, 32), dtype=np.uint8)
        fn = self._fn("fri_layer_commit")
        fn.restype = ctypes.c_int
        assert fn(ctypes.c_int(hasher), _ptr(v), _u64(rows), ctypes.c_uint(D), _u64(N), _ptr(leaves), _ptr(nodes)) == 0
        return leaves, nodes

    def apply_drp(self, transposed, N, offset, alpha, D=1):
        v = _u64arr(transposed)
        rows = v.size // (N * D * self.W)
        out = np.empty(rows * D * self.W, dtype=np.uint64)
        po, pa = self.pack([offset]), _u64arr(alpha)
        self._fn("apply_drp")(_ptr(v), _u64(rows), ctypes.c_uint(D), _u64(N), _ptr(po), _ptr(pa), _ptr(out))
        return out

    # ---- DEEP composition (arrays are internal-form words; points / coefficients are flat arrays of D*W words) ----
    def evaluate_columns_at(self, polys, c, x, D, pD=1):
        """polys: c columns of n coefficients of degree pD; x: one degree-D element.  Returns (c, D*W) words."""
        v, px = _u64arr(polys), _u64arr(x)
        n = v.size // (c * pD * self.W)
        out = np.empty((c, D * self.W), dtype=np.uint64)
        self._fn("evaluate_columns_at")(_ptr(v), _u64(c), _u64(n), ctypes.c_uint(pD), _ptr(px), ctypes.c_uint(D), _ptr(out))
        return out

    def deep_compose(self, main, c_main, aux, c_aux, quot, c_q, n, D, z, cc_trace, cc_constraints, ood_t_cur, ood_t_next,
                     ood_q_cur, ood_q_next):
        """DeepCompositionPoly::add_trace_polys (prover/src/composer/mod.rs:67-169) -> n*D*W words."""
        arrs = [_u64arr(a if a is not None else np.zeros(1, dtype=np.uint64))
                for a in (main, aux, quot, z, cc_trace, cc_constraints, ood_t_cur, ood_t_next, ood_q_cur, ood_q_next)]
        out = np.empty(n * D * self.W, dtype=np.uint64)
        m, a, q, pz, cct, ccc, otc, otn, oqc, oqn = arrs
        self._fn("deep_compose")(_ptr(m), _u64(c_main), _ptr(a), _u64(c_aux), _ptr(q), _u64(c_q), _u64(n), ctypes.c_uint(D),
                                 _ptr(pz), _ptr(cct), _ptr(ccc), _ptr(otc), _ptr(otn), _ptr(oqc), _ptr(oqn), _ptr(out))
        return out

    # ---- constraint evaluation (constraints_tmpl.inc) -----------------------------------------------------------
    AIR_FIB_SMALL, AIR_RESCUE, AIR_FIB8, AIR_MULFIB2, AIR_MULFIB8, AIR_VDF, AIR_VDF_EXEMPT, AIR_RESCUE_RAPS = 0, 1, 2, 3, 4, 5, 6, 7
    # width, transition constraints, periodic columns, cycle
    AIR_SHAPES = {0: (2, 2, 0, 0), 1: (4, 4, 9, 16), 2: (2, 2, 0, 0), 3: (2, 2, 0, 0), 4: (8, 8, 0, 0), 5: (1, 1, 0, 0), 6: (1, 1, 0, 0),
                  7: (8, 8, 10, 16)}
    # auxiliary segment: width (columns of E), transition constraints, random elements
    AIR_AUX_SHAPES = {7: (3, 3, 3)}

    def air_evaluate_transition(self, air, D, cur, nxt, periodic):
        """Air::evaluate_transition over degree-D elements; cur / nxt: width*D*W words, periodic: num_periodic*D*W."""
        w, nt, npc, _ = self.AIR_SHAPES[air]
        c, n_, pv = _u64arr(cur), _u64arr(nxt), _u64arr(periodic if len(periodic) else np.zeros(1, dtype=np.uint64))
        out = np.empty(nt * D * self.W, dtype=np.uint64)
        fn = self._fn("air_evaluate_transition")
        fn.restype = ctypes.c_int
        assert fn(ctypes.c_int(air), ctypes.c_uint(D), _ptr(c), _ptr(n_), _ptr(pv), _ptr(out)) == 0
        return out

    def air_evaluate_aux_transition(self, air, Dm, D, mcur, mnxt, acur, anxt, periodic, rand):
        """Air::evaluate_aux_transition: main frame / periodic values over F (Dm components per element), aux frame / rand over E."""
        _, nta, _ = self.AIR_AUX_SHAPES[air]
        out = np.empty(nta * D * self.W, dtype=np.uint64)
        a = [_u64arr(x) for x in (mcur, mnxt, acur, anxt, periodic, rand)]
        fn = self._fn("air_evaluate_aux_transition")
        fn.restype = ctypes.c_int
        assert fn(ctypes.c_int(air), ctypes.c_uint(Dm), ctypes.c_uint(D), *[_ptr(x) for x in a], _ptr(out)) == 0
        return out

    def air_periodic_polys(self, air):
        """Air::get_periodic_column_polys -> (num_periodic, cycle*W) words."""
        _, _, npc, cyc = self.AIR_SHAPES[air]
        out = np.empty((max(npc, 1), max(cyc, 1) * self.W), dtype=np.uint64)
        fn = self._fn("air_periodic_polys")
        fn.restype = ctypes.c_int
        assert fn(ctypes.c_int(air), _ptr(out)) == 0
        return out[:npc]

    def evaluate_constraints(self, air, lde, row_width, n, lde_blowup, ce_blowup, offset, D, cc_t, assertions, cc_b):
        """DefaultConstraintEvaluator::evaluate (single segment, single-value assertions).
        lde: (n*lde_blowup, row_width*W) words; assertions: list of (column, step, value words); cc_t / cc_b: flat words.
        Returns n*ce_blowup*D*W words (CompositionPolyTrace)."""
        l, t, b = _u64arr(lde), _u64arr(cc_t), _u64arr(cc_b)
        cols = np.array([a[0] for a in assertions], dtype=np.uint64)
        steps = np.array([a[1] for a in assertions], dtype=np.uint64)
        vals = np.concatenate([_u64arr(a[2]).reshape(-1) for a in assertions])
        po = self.pack([offset])
        out = np.empty(n * ce_blowup * D * self.W, dtype=np.uint64)
        fn = self._fn("evaluate_constraints")
        fn.restype = ctypes.c_int
        rc = fn(ctypes.c_int(air), _ptr(l), _u64(row_width), _u64(n), _u64(lde_blowup), _u64(ce_blowup), _ptr(po), ctypes.c_uint(D),
                _ptr(t), _u64(len(assertions)), _ptr(cols), _ptr(steps), _ptr(vals), _ptr(b), _ptr(out))
        assert rc == 0
        return out

    def evaluate_constraints_multi(self, air, lde, row_width, n, lde_blowup, ce_blowup, offset, D, cc_t, assertions, cc_b):
        """the same with assertions of every kind (air/src/air/assertions/mod.rs:57-120): a list of (column, first_step, stride, value
        words) — stride 0 = Assertion::single (one value), stride > 0 with ONE value = Assertion::periodic, with trace_length / stride
        values = Assertion::sequence.  cc_b: one coefficient (D elements) per assertion, in the list's order."""
        l, t, b = _u64arr(lde), _u64arr(cc_t), _u64arr(cc_b)
        cols = np.array([a[0] for a in assertions], dtype=np.uint64)
        first = np.array([a[1] for a in assertions], dtype=np.uint64)
        stride = np.array([a[2] for a in assertions], dtype=np.uint64)
        vl = [_u64arr(a[3]).reshape(-1) for a in assertions]
        nvals = np.array([len(v) // self.W for v in vl], dtype=np.uint64)
        voff = np.concatenate([[0], np.cumsum(nvals)[:-1]]).astype(np.uint64)
        vals = np.concatenate(vl)
        po = self.pack([offset])
        out = np.empty(n * ce_blowup * D * self.W, dtype=np.uint64)
        fn = self._fn("evaluate_constraints_multi")
        fn.restype = ctypes.c_int
        rc = fn(ctypes.c_int(air), _ptr(l), _u64(row_width), _u64(n), _u64(lde_blowup), _u64(ce_blowup), _ptr(po), ctypes.c_uint(D),
                _ptr(t), _u64(len(assertions)), _ptr(cols), _ptr(first), _ptr(stride), _ptr(nvals), _ptr(voff), _ptr(vals), _ptr(b), _ptr(out))
        assert rc == 0, rc
        return out

    def evaluate_constraints_full(self, air, lde, row_width, aux_lde, aux_row_width, n, lde_blowup, ce_blowup, offset, D, cc_t, assertions,
                                  cc_b, aux_assertions, cc_x, rand):
        """the same for a trace with an auxiliary segment (evaluate_fragment_full): aux_lde (n*lde_blowup, aux_row_width*W) words,
        rows of E elements; aux_assertions: (column, step, E value words); cc_t = main then aux coefficients; rand: E elements."""
        l, x, t, b, cx, rd = (_u64arr(v) for v in (lde, aux_lde, cc_t, cc_b, cc_x, rand))
        cols = np.array([a[0] for a in assertions], dtype=np.uint64)
        steps = np.array([a[1] for a in assertions], dtype=np.uint64)
        vals = np.concatenate([_u64arr(a[2]).reshape(-1) for a in assertions])
        xcols = np.array([a[0] for a in aux_assertions], dtype=np.uint64)
        xsteps = np.array([a[1] for a in aux_assertions], dtype=np.uint64)
        xvals = np.concatenate([_u64arr(a[2]).reshape(-1) for a in aux_assertions])
        po = self.pack([offset])
        out = np.empty(n * ce_blowup * D * self.W, dtype=np.uint64)
        fn = self._fn("evaluate_constraints_full")
        fn.restype = ctypes.c_int
        rc = fn(ctypes.c_int(air), _ptr(l), _u64(row_width), _ptr(x), _u64(aux_row_width), _u64(n), _u64(lde_blowup), _u64(ce_blowup), _ptr(po),
                ctypes.c_uint(D), _ptr(t), _u64(len(assertions)), _ptr(cols), _ptr(steps), _ptr(vals), _ptr(b), _u64(len(aux_assertions)),
                _ptr(xcols), _ptr(xsteps), _ptr(xvals), _ptr(cx), _ptr(rd), _ptr(out))
        assert rc == 0, rc
        return out

    def rescue_raps_build_trace(self, seeds, permuted_seeds):
        """f128 only: RescueRapsProver::build_trace; seeds / permuted_seeds: lists of [a, b] -> (8, 16*len(seeds)*W) words."""
        assert self.name == "f128" and len(seeds) == len(permuted_seeds)
        out = np.empty((8, 16 * len(seeds) * self.W), dtype=np.uint64)
        ps, pp = self.pack([v for s_ in seeds for v in s_]), self.pack([v for s_ in permuted_seeds for v in s_])
        self._fn("rescue_raps_build_trace")(_ptr(ps), _ptr(pp), _u64(len(seeds)), _ptr(out))
        return out

    def rescue_raps_build_aux(self, trace, D, rand):
        """f128 only: RescueRapsProver::build_aux_trace; trace (8, n*W) words, rand: 3*D*W words -> (3, n*D*W) words."""
        assert self.name == "f128"
        tr, rd = _u64arr(trace), _u64arr(rand)
        n = tr.size // (8 * self.W)
        out = np.empty((3, n * D * self.W), dtype=np.uint64)
        self._fn("rescue_raps_build_aux")(_ptr(tr), _u64(n), ctypes.c_uint(D), _ptr(rd), _ptr(out))
        return out

    def fib_small_build_trace(self, n):
        out = np.empty((2, n * self.W), dtype=np.uint64)
        self._fn("fib_small_build_trace")(_u64(n), _ptr(out))
        return out

    def fib8_build_trace(self, n):
        out = np.empty((2, n * self.W), dtype=np.uint64)
        self._fn("fib8_build_trace")(_u64(n), _ptr(out))
        return out

    def mulfib2_build_trace(self, n):
        out = np.empty((2, n * self.W), dtype=np.uint64)
        self._fn("mulfib2_build_trace")(_u64(n), _ptr(out))
        return out

    def mulfib8_build_trace(self, n):
        out = np.empty((8, n * self.W), dtype=np.uint64)
        self._fn("mulfib8_build_trace")(_u64(n), _ptr(out))
        return out

    def vdf_build_trace(self, seed, n, exempt=False):
        """f128 only: VdfProver::build_trace (regular / exempt) -> (1, n*W) words."""
        assert self.name == "f128"
        out = np.empty((1, n * self.W), dtype=np.uint64)
        ps = self.pack([seed])
        self._fn("vdf_build_trace")(_ptr(ps), _u64(n), ctypes.c_int(int(exempt)), _ptr(out))
        return out

    def rescue_build_trace(self, seed, iterations):
        """f128 only: RescueProver::build_trace -> (4, 16*iterations*W) words."""
        assert self.name == "f128"
        out = np.empty((4, 16 * iterations * self.W), dtype=np.uint64)
        ps = self.pack(seed)
        self._fn("rescue_build_trace")(_ptr(ps), _u64(iterations), _ptr(out))
        return out


F128_M = 2**128 - 45 * 2**40 + 1
f128 = GenericField("f128", 2, F128_M)
f64t = GenericField("f64t", 1, M)
F62_M = 4611624995532046337
f62 = GenericField("f62", 1, F62_M)


def f62_new(v):
    lib().or_f62_new1.restype = _u64
    return lib().or_f62_new1(_u64(v))


def f62_as_int(v):
    lib().or_f62_as_int1.restype = _u64
    return lib().or_f62_as_int1(_u64(v))
