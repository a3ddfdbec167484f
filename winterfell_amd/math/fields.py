"""Host-side helpers for the f64 field's representation (math/src/field/f64/mod.rs).

Only representation conversions live here (they are what `BaseElement::new` / `as_int` do on the host when a
caller prepares inputs or reads results); all bulk arithmetic runs on the GPU.
"""
import numpy as np

M = 0xFFFFFFFF00000001          # f64/mod.rs:46
GENERATOR = 7                   # f64/mod.rs:251
TWO_ADICITY = 32                # f64/mod.rs:255
_R = (1 << 64) % M
_RINV = pow(_R, M - 2, M)


def new(value: int) -> int:
    """BaseElement::new — canonical integer -> internal Montgomery form (f64/mod.rs:72-74)."""
    return (value % M) * _R % M


def as_int(inner: int) -> int:
    """BaseElement::as_int — internal form -> canonical integer (f64/mod.rs:89-91)."""
    return inner * _RINV % M


def from_ints(vals) -> np.ndarray:
    """Vectorised BaseElement::new over canonical integers (< M), exact via object arithmetic."""
    a = np.asarray(vals, dtype=np.uint64)
    out = (a.astype(object) * _R) % M
    return out.astype(np.uint64).reshape(a.shape)


def to_ints(vals) -> np.ndarray:
    a = np.asarray(vals, dtype=np.uint64)
    out = (a.astype(object) * _RINV) % M
    return out.astype(np.uint64).reshape(a.shape)


# ---- field descriptors (what the C ABI's `field` argument selects) --------------------------------------
class Field:
    """A base field of the reference: id for the C ABI, words (u64) per element, modulus, 2-adicity, generator."""

    def __init__(self, wf_id, name, words, modulus, two_adicity, generator, montgomery, max_ext, two_adic_root):
        self.ID, self.name, self.W, self.M = wf_id, name, words, modulus
        self.TWO_ADICITY, self.GENERATOR, self.montgomery, self.MAX_EXT = two_adicity, generator, montgomery, max_ext
        self.TWO_ADIC_ROOT_OF_UNITY = two_adic_root
        self._r = (1 << 64) % modulus if montgomery else 1
        self._rinv = pow(self._r, modulus - 2, modulus)

    def get_root_of_unity(self, n):
        """StarkField::get_root_of_unity(n) (math/src/field/traits.rs:262-268): canonical integer of the 2^n-th root."""
        assert 0 < n <= self.TWO_ADICITY, "order cannot exceed 2^%d" % self.TWO_ADICITY
        return pow(self.TWO_ADIC_ROOT_OF_UNITY, 1 << (self.TWO_ADICITY - n), self.M)

    def new(self, value):
        """BaseElement::new: canonical integer -> internal representation (python int)."""
        return (value % self.M) * self._r % self.M

    def as_int(self, inner):
        return inner * self._rinv % self.M

    def pack(self, inner_vals):
        """python ints (internal form) -> uint64 array of W little-endian words per element."""
        vals = [int(v) for v in np.asarray(inner_vals, dtype=object).reshape(-1)]
        out = np.empty((len(vals), self.W), dtype=np.uint64)
        for k in range(self.W):
            out[:, k] = [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for v in vals]
        return out.reshape(-1)

    def unpack(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, self.W)
        return [sum(int(a[i, k]) << (64 * k) for k in range(self.W)) for i in range(a.shape[0])]

    def from_ints(self, vals):
        return self.pack([self.new(int(v)) for v in np.asarray(vals, dtype=object).reshape(-1)])

    def to_ints(self, arr):
        return [self.as_int(v) for v in self.unpack(arr)]

    def element_words(self, inner):
        """one element (internal form, python int) as a ctypes-ready uint64 array."""
        return self.pack([inner])


f64 = Field(0, "f64", 1, M, 32, 7, True, 3, 7277203076849721926)                                    # math/src/field/f64/mod.rs
f128 = Field(1, "f128", 2, 2**128 - 45 * 2**40 + 1, 40, 3, False, 2, 0x120532e7b364080a86b8723e1920f4aa)           # math/src/field/f128/mod.rs:40,152,157
f62 = Field(2, "f62", 1, 4611624995532046337, 39, 3, True, 3, 4421547261963328785)                  # math/src/field/f62/mod.rs:39,194
