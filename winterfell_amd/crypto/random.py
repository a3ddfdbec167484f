"""Proof-of-work side of RandomCoin (crypto/src/random/mod.rs:43, default.rs:141-146) on the GPU.

The coin itself (reseed / draw) is sequential host logic and stays with the caller (the Rust channel in a real
integration, the oracle's restatement in the tests); what is data-parallel is evaluating
`check_leading_zeros(nonce)` for millions of nonces, which is what `ProverChannel::grind_query_seed`
(prover/src/channel.rs:169-185) does.
"""
import ctypes

import numpy as np

from .._lib import WfError, default_context, ptr

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


class DefaultRandomCoin:
    """crypto::DefaultRandomCoin (crypto/src/random/default.rs): seed / reseed / draw over one of the library's hashers.
    Sequential host logic (one small hash per call), kept here so the Python mirror can drive a whole proof; in a real
    integration this is the reference's own Rust coin.  `field` is the base field (a fields.Field)."""

    def __init__(self, hasher, field, seed_elements, ctx=None):
        self.hasher, self.field, self.ctx = hasher, field, ctx
        self.seed = hasher.hash_elements(np.ascontiguousarray(seed_elements, dtype=np.uint64), ctx, field=field)   # :114-117
        self.counter = 0
        self._ahead = []            # as_bytes of merge_with_int(seed, counter + 1 + k): a run of counters is ONE device batch

    @classmethod
    def from_seed(cls, hasher, field, seed_digest, ctx=None):
        """a coin whose seed digest is already known (e.g. the digest of the empty seed, the same for every proof)"""
        c = cls.__new__(cls)
        c.hasher, c.field, c.ctx = hasher, field, ctx
        c.seed, c.counter, c._ahead = np.array(seed_digest, dtype=np.uint8, copy=True), 0, []
        return c

    def reseed(self, data):
        """seed = merge(seed, data); counter = 0 (:150-153)"""
        self.seed = self.hasher.merge(np.stack([self.seed, np.asarray(data, dtype=np.uint8).reshape(32)]), self.ctx)
        self.counter, self._ahead = 0, []

    def prefetch(self, count):
        """evaluate the next `count` values of next() in one batch (wf_hash_merge_with_int_batch); pure look-ahead — the
        values are exactly those the sequential calls would produce, so the transcript does not change"""
        if count > len(self._ahead):
            first = self.counter + 1 + len(self._ahead)
            rows = self.hasher.merge_with_int(self.seed, first, count - len(self._ahead), ctx=self.ctx)
            self._ahead.extend(self.hasher.digest_as_bytes(r) for r in rows)

    def _next(self):
        """merge_with_int(seed, ++counter) as Digest::as_bytes (:92-98)"""
        if not self._ahead:
            self.prefetch(1)
        self.counter += 1
        return self._ahead.pop(0)

    def draw(self, ext_degree=1):
        """draw::<E> (:185-199): the first ELEMENT_BYTES of next() must decode to canonical base elements; up to 1000 tries.
        Returns ext_degree * W words (internal form)."""
        f = self.field
        nb = 8 * f.W
        for _ in range(1000):
            b = self._next()
            if len(b) < ext_degree * nb:
                raise RuntimeError("digest too short for the requested element")
            vals = [int.from_bytes(b[k * nb:(k + 1) * nb], "little") for k in range(ext_degree)]
            if all(v < f.M for v in vals):
                return f.pack([f.new(v) for v in vals])
        raise RuntimeError("FailedToDrawFieldElement(1000)")

    def draw_many(self, count, ext_degree=1):
        """`count` consecutive draws (what the channel's get_*_coeffs loops do), the hashes batched"""
        self.prefetch(count)
        return np.stack([self.draw(ext_degree) for _ in range(count)])

    def check_leading_zeros(self, value):
        return check_leading_zeros(self.hasher, self.seed, value, ctx=self.ctx)

    # ---- hand-over to / from the device-resident coin (include/winterfell_hip.h: wf_coin_*) --------------------------------
    def to_device(self):
        """the coin's state as a DeviceCoin: what a chain of reseed / draw steps queued on the stream starts from"""
        return DeviceCoin(self.hasher, self.field, self.seed, self.counter, self.ctx)

    def take_back(self, device_coin):
        """continue on the host from where the device coin stopped (waits for the stream)"""
        self.seed, self.counter = device_coin.read()
        self._ahead = []

    def draw_integers(self, num_values, domain_size, nonce):
        """:209-248: reseed with the nonce, then masked 8-byte heads (duplicates are removed by the caller)."""
        assert domain_size & (domain_size - 1) == 0, "domain size must be a power of two"
        assert num_values < domain_size, "number of values must be smaller than domain size"
        self.seed = self.hasher.merge_with_int(self.seed, nonce, ctx=self.ctx)
        self.counter, self._ahead = 0, []
        self.prefetch(num_values)
        return [int.from_bytes(self._next()[:8], "little") & (domain_size - 1) for _ in range(num_values)]


class DeviceCoin:
    """DefaultRandomCoin with its state (seed digest, counter) in device memory: reseed / draw are single-lane kernels queued on
    the context's stream, so a commit -> reseed -> draw -> use chain (the FRI layers, wf_fri_build_layers) runs without a host
    round trip per link.  Same transcript as the host coin, value for value."""

    def __init__(self, hasher, field, seed, counter=0, ctx=None):
        self.hasher, self.field = hasher, field
        self.ctx = ctx or default_context()
        # WF_COIN_BYTES = 64: seed digest [0, 32), counter (little-endian u64) [32, 40), "a draw failed" flag (u32) [40, 44)
        self._image = np.zeros(64, dtype=np.uint8)
        self._image[:32] = np.ascontiguousarray(seed).view(np.uint8).reshape(32)
        self._image[32:40] = np.frombuffer(int(counter).to_bytes(8, "little"), dtype=np.uint8)
        self.state = None                                        # uploaded on first use, or placed by move_to()
        self._host_image = None

    def _upload(self):
        if self.state is None:
            self.state = self.ctx.to_device(self._image)
        return self.state

    def move_to(self, d_state):
        """keep the state in the caller's 64 device bytes (so that it comes back with the caller's own read)"""
        if self.state is None:
            import torch
            d_state.copy_(torch.from_numpy(self._image))         # one host-to-device copy straight into the caller's bytes
        else:
            d_state.copy_(self.state)
        self.state = d_state

    def set_host_image(self, image):
        """the 64 state bytes as the caller has just read them back: read() then needs no transfer of its own"""
        self._host_image = np.array(image, dtype=np.uint8, copy=True)

    def reseed(self, d_digest, d_copy=None):
        """seed = merge(seed, digest) for a digest that is on the device (e.g. nodes[1] of a Merkle tree just built)"""
        self._host_image = None
        self.ctx.call("wf_coin_reseed", self.hasher.HASH_ID, ptr(self._upload()), ptr(d_digest), ptr(d_copy) if d_copy is not None else None)

    def draw(self, ext_degree=1, count=1):
        """`count` draws of an E element -> device tensor of count * ext_degree * W words (internal form)"""
        self._host_image = None
        out = self.ctx.empty_u64(count, ext_degree * self.field.W)
        self.ctx.call("wf_coin_draw", self.hasher.HASH_ID, self.field.ID, ext_degree, ptr(self._upload()), count, ptr(out))
        return out

    def read(self):
        """(seed, counter) after everything queued so far; raises like the reference when a draw ran out of tries"""
        img = self._host_image if self._host_image is not None else self.ctx.to_host(self._upload())
        failed = int.from_bytes(img[40:44].tobytes(), "little")
        if failed & 2:          # wf_coin_grind searched its whole range (the reference panics: channel.rs:169-185)
            raise RuntimeError("nonce not found")
        if failed:
            raise RuntimeError("FailedToDrawFieldElement(1000)")
        return np.array(img[:32], copy=True), int.from_bytes(img[32:40].tobytes(), "little")
