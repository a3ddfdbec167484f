"""Hasher / ElementHasher (crypto/src/hash/mod.rs:31-64) backed by the HIP kernels.

Digests are 32-byte rows: raw bytes for Blake3_256 (ByteDigest<32>), four Montgomery-form u64 for Rp64_256
(ElementDigest, crypto/src/hash/rescue/rp64_256/digest.rs:16).  The scalar methods below launch a one-element
batch on the GPU — they exist so tests can be written exactly like the reference's; bulk work goes through
RowMatrix.hash_rows / MerkleTree.
"""
import ctypes

import numpy as np

from .._lib import WF_FIELD_F64, WF_HASH_BLAKE3_256, WF_HASH_RP64_256, WF_HASH_SHA3_256, WF_HASH_RPJIVE64_256, WF_HASH_RP62_248, WF_HASH_BLAKE3_192, default_context, ptr
from ..math import fields


class _Hasher:
    HASH_ID = None
    COLLISION_RESISTANCE = 128
    # whether a chain of coin steps is worth queueing as single-lane kernels (crypto/random.py DeviceCoin): yes for the byte
    # hashers (one compression, ~2 us); a Rescue permutation on one lane costs more than the host round trip it would save
    DEVICE_COIN = True

    @classmethod
    def merge(cls, values, ctx=None):
        """Hasher::merge(&[Digest; 2]) — values: (2, 32) uint8 (or (k, 2, 32) for a batch) -> (32,) / (k, 32)."""
        ctx = ctx or default_context()
        v = np.ascontiguousarray(values).view(np.uint8)
        batch = v.reshape(-1, 64)
        d_in = ctx.to_device(batch)
        d_out = ctx.empty_u8(batch.shape[0], 32)
        ctx.call("wf_hash_merge_batch", cls.HASH_ID, ptr(d_in), batch.shape[0], ptr(d_out))
        out = ctx.to_host(d_out)
        return out[0] if v.size == 64 else out

    @classmethod
    def hash_elements(cls, elements, ctx=None, field=fields.f64):
        """ElementHasher::hash_elements — elements: base-field words in internal form (extension elements
        flattened; field.W u64 words per base element).  A 2-D array hashes each row."""
        ctx = ctx or default_context()
        e = np.ascontiguousarray(elements, dtype=np.uint64)
        rows = e.reshape(1, -1) if e.ndim == 1 else e
        if rows.shape[1] == 0:
            rows = np.zeros((rows.shape[0], field.W), dtype=np.uint64)
            width, take = 1, 0
        else:
            width = take = rows.shape[1] // field.W
        d_in = ctx.to_device(rows)
        d_out = ctx.empty_u8(rows.shape[0], 32)
        ctx.call("wf_hash_elements_batch", cls.HASH_ID, field.ID, ptr(d_in), rows.shape[0], width, take, ptr(d_out))
        out = ctx.to_host(d_out)
        return out[0] if e.ndim == 1 else out


    @classmethod
    def hash(cls, data, ctx=None):
        """Hasher::hash(&[u8]) (crypto/src/hash/mod.rs:33-35): digest of a byte string; a list of equal-length byte strings
        is hashed as one batch -> (k, 32)."""
        ctx = ctx or default_context()
        single = isinstance(data, (bytes, bytearray, memoryview))
        msgs = [bytes(data)] if single else [bytes(m) for m in data]
        n = len(msgs[0]) if msgs else 0
        assert all(len(m) == n for m in msgs), "a batch hashes strings of one length"
        stride = max(8, -(-n // 8) * 8)
        buf = np.zeros((len(msgs), stride), dtype=np.uint8)
        for i, m in enumerate(msgs):
            buf[i, :n] = np.frombuffer(m, dtype=np.uint8)
        d_out = ctx.empty_u8(max(len(msgs), 1), 32)
        d_in = ctx.to_device(buf)       # a NAME, alive until the read-back below has waited for the stream: a temporary inside the call
        ctx.call("wf_hash_bytes_batch", cls.HASH_ID, ptr(d_in), len(msgs), stride, n, ptr(d_out))     # expression dies before the kernel runs
        out = ctx.to_host(d_out)[:len(msgs)]
        del d_in
        return out[0] if single else out

    @classmethod
    def merge_many(cls, values, ctx=None):
        """Hasher::merge_many(&[Digest]) (crypto/src/hash/mod.rs:39-41) — values: (k, 32) digests -> (32,), or (n, k, 32)
        for n independent merges -> (n, 32).  Byte hashers hash the concatenated digest bytes (24 per digest for
        Blake3_192); the Rescue hashers hash the digests' elements (overridden below with the same library call)."""
        ctx = ctx or default_context()
        v = np.ascontiguousarray(values).view(np.uint8)
        single = v.ndim == 2
        batch = v.reshape(1, -1, 32) if single else v.reshape(v.shape[0], -1, 32)
        d_in = ctx.to_device(np.ascontiguousarray(batch))
        d_out = ctx.empty_u8(batch.shape[0], 32)
        ctx.call("wf_hash_merge_many_batch", cls.HASH_ID, ptr(d_in), batch.shape[0], batch.shape[1], ptr(d_out))
        out = ctx.to_host(d_out)
        return out[0] if single else out

    @classmethod
    def merge_with_int(cls, seed, value, count=None, ctx=None):
        """Hasher::merge_with_int(seed, value) (crypto/src/hash/mod.rs:44-46); with `count`, the digests for
        value, value+1, ..., value+count-1 as a (count, 32) array (RandomCoin::next for a run of counters)."""
        ctx = ctx or default_context()
        s = np.ascontiguousarray(seed).view(np.uint8).reshape(32)
        n = 1 if count is None else int(count)
        d_out = ctx.empty_u8(max(n, 1), 32)
        ctx.call("wf_hash_merge_with_int_batch", cls.HASH_ID, s.ctypes.data_as(ctypes.c_void_p), int(value), n, ptr(d_out))
        out = ctx.to_host(d_out)[:n]
        return out[0] if count is None else out

    @classmethod
    def digest_as_bytes(cls, digest):
        """Digest::as_bytes (ByteDigest: the bytes themselves, crypto/src/hash/mod.rs:83-101)."""
        return np.ascontiguousarray(digest).view(np.uint8).tobytes()


class Blake3_256(_Hasher):
    """crypto::hash::Blake3_256<f64::BaseElement> (crypto/src/hash/blake/mod.rs:24-66)."""
    HASH_ID = WF_HASH_BLAKE3_256


class Blake3_192(_Hasher):
    """crypto::hash::Blake3_192<B> (crypto/src/hash/blake/mod.rs:68-125): ByteDigest<24>.  The library keeps digests in
    32-byte slots; for this hasher bytes 24..31 of a slot are zero and `digest_as_bytes` returns the 24 meaningful ones."""
    HASH_ID = WF_HASH_BLAKE3_192
    COLLISION_RESISTANCE = 96

    @classmethod
    def digest_as_bytes(cls, digest):
        return np.ascontiguousarray(digest).view(np.uint8).tobytes()[:24]


class Sha3_256(_Hasher):
    """crypto::hash::Sha3_256<B> (crypto/src/hash/sha/mod.rs:21-66); ByteDigest<32>, same call structure as Blake3_256."""
    HASH_ID = WF_HASH_SHA3_256


def _bytes_to_elements(data, field, strict_index=True):
    """The Rescue hashers' byte -> element rule (rp64_256/mod.rs:123-160, rp64_256_jive/mod.rs:119-164, rp62_248/mod.rs:97-141):
    7-byte little-endian chunks, the LAST chunk zero-padded with a 1 byte appended after its data.  Rp62_248 decides "last"
    by the position inside the current rate block rather than by the chunk index (rp62_248/mod.rs:119): for inputs of more
    than 8 chunks its final chunk is taken verbatim (and a short one panics in copy_from_slice) — reproduced with
    strict_index=False."""
    data = bytes(data)
    chunks = [data[i:i + 7] for i in range(0, len(data), 7)]
    out, pos = [], 0
    for index, ch in enumerate(chunks):
        last = (index if strict_index else pos) >= len(chunks) - 1
        if last:
            buf = ch + b"\x01" + bytes(7 - len(ch))
        else:
            if len(ch) != 7:
                raise ValueError("source slice length (%d) does not match destination slice length (7)" % len(ch))
            buf = ch + b"\x00"
        out.append(field.new(int.from_bytes(buf, "little")))
        pos = (pos + 1) % 8
    return field.pack(out) if out else np.zeros(0, dtype=np.uint64)


class Rp64_256(_Hasher):
    """crypto::hash::Rp64_256 (crypto/src/hash/rescue/rp64_256/mod.rs:123-257)."""
    HASH_ID = WF_HASH_RP64_256
    DEVICE_COIN = True      # round 5: the coin's steps run on 16-lane groups (csrc/coin.hip, rescue_coop.cuh), ~17 us each instead of 0.2 ms

    @classmethod
    def hash(cls, data, ctx=None):
        """Hasher::hash (mod.rs:123-178): the sponge over the string's 7-byte chunks — the same absorb / padding as
        hash_elements over those elements (the conversion is host logic, the permutations run on the device)."""
        if not isinstance(data, (bytes, bytearray, memoryview)):
            return np.stack([cls.hash(m, ctx) for m in data])
        return cls.hash_elements(_bytes_to_elements(data, fields.f64), ctx)

    # merge_many: hash_elements over the digests' elements (rp64_256/mod.rs:194-196) — the base class's library call

    @staticmethod
    def digest_as_bytes(digest):
        """ElementDigest::as_bytes — canonical little-endian (rp64_256/digest.rs:36-45)."""
        return fields.to_ints(np.ascontiguousarray(digest).view(np.uint64)).tobytes()


class RpJive64_256(Rp64_256):
    """crypto::hash::RpJive64_256 (crypto/src/hash/rescue/rp64_256_jive/mod.rs:62-313): ElementDigest as Rp64_256;
    merge is the Jive compression, merge_many = hash_elements over the digests' elements (mod.rs:219-221)."""
    HASH_ID = WF_HASH_RPJIVE64_256


class Rp62_248(_Hasher):
    """crypto::hash::Rp62_248 (crypto/src/hash/rescue/rp62_248/mod.rs:62-239): Rescue-Prime over f62; a digest is four
    f62 words (ElementDigest, digest.rs:16), serialised as 31 bytes."""
    HASH_ID = WF_HASH_RP62_248
    DEVICE_COIN = True

    @classmethod
    def hash_elements(cls, elements, ctx=None, field=fields.f62):
        return super().hash_elements(elements, ctx, field)

    @classmethod
    def hash(cls, data, ctx=None):
        """Hasher::hash (rp62_248/mod.rs:97-153), including its position-based last-chunk test."""
        if not isinstance(data, (bytes, bytearray, memoryview)):
            return np.stack([cls.hash(m, ctx) for m in data])
        return cls.hash_elements(_bytes_to_elements(data, fields.f62, strict_index=False), ctx)

    @staticmethod
    def digest_as_bytes(digest):
        """ElementDigest::as_bytes (rp62_248/digest.rs:37-51): 4 x 62 bits packed little-endian."""
        f = fields.f62
        v = [f.as_int(int(w) % f.M) for w in np.ascontiguousarray(digest).view(np.uint64).reshape(4)]
        words = [(v[0] | (v[1] << 62)) & (2**64 - 1), ((v[1] >> 2) | (v[2] << 60)) & (2**64 - 1),
                 ((v[2] >> 4) | (v[3] << 58)) & (2**64 - 1), v[3] >> 6]
        return np.array(words, dtype=np.uint64).tobytes()
