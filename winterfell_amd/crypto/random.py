"""Proof-of-work side of RandomCoin (crypto/src/random/mod.rs:43, default.rs:141-146) on the GPU.

The coin itself (reseed / draw) is sequential host logic and stays with the caller (the Rust channel in a real
integration, the oracle's restatement in the tests); what is data-parallel is evaluating
`check_leading_zeros(nonce)` for millions of nonces, which is what `ProverChannel::grind_query_seed`
(prover/src/channel.rs:169-185) does.
"""
import ctypes

import numpy as np

from .._lib import WfError, default_context

WF_ERR_NOT_FOUND = 9


def check_leading_zeros(hasher, seed, value, count=None, ctx=None):
    """RandomCoin::check_leading_zeros for `value` (or value .. value+count-1): trailing zero bits of the first
    8 bytes (little-endian) of merge_with_int(seed, value).as_bytes()."""
    d = hasher.merge_with_int(seed, value, 1 if count is None else count, ctx)
    heads = np.array([int.from_bytes(hasher.digest_as_bytes(row)[:8], "little") for row in d], dtype=object)
    tz = np.array([64 if h == 0 else (h & -h).bit_length() - 1 for h in heads], dtype=np.uint32)
    return int(tz[0]) if count is None else tz


def grind_query_seed(hasher, seed, grinding_factor, first_nonce=1, max_nonce=(1 << 64) - 2, ctx=None):
    """ProverChannel::grind_query_seed: the smallest nonce >= first_nonce whose check_leading_zeros is at least
    `grinding_factor` (the value the reference's serial `find` returns).  Raises like the reference's
    `expect("nonce not found")` when the range is exhausted."""
    ctx = ctx or default_context()
    s = np.ascontiguousarray(seed).view(np.uint8).reshape(32)
    nonce = ctypes.c_uint64(0)
    try:
        ctx.call("wf_grind", hasher.HASH_ID, s.ctypes.data_as(ctypes.c_void_p), int(grinding_factor), int(first_nonce),
                 int(max_nonce), ctypes.byref(nonce))
    except WfError as e:
        if e.status == WF_ERR_NOT_FOUND:
            raise RuntimeError("nonce not found") from e
        raise
    return int(nonce.value)
