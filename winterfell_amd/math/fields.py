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
