"""Mirror of the reference's `math` crate surface on the hot path (fields' internal forms + fft)."""
from . import fft, fields, utils  # noqa: F401
