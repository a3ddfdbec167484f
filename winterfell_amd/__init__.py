"""winterfell_amd — MI355X (gfx950) implementation of Winterfell's STARK proving hot path.

The product is ``libwinterfell_hip.so`` (hand-written HIP kernels behind the C ABI declared in
``include/winterfell_hip.h``).  This package is the host-side mirror of the reference's own interfaces for that
path, so tests read like the reference's tests:

  winterfell_amd.math.fft      <->  math::fft        (evaluate_poly, interpolate_poly, ..._with_offset, get_twiddles)
  winterfell_amd.crypto        <->  crypto::{Hasher, ElementHasher, MerkleTree}
  winterfell_amd.prover        <->  prover::matrix::{ColMatrix, RowMatrix}, prover::trace::trace_lde::DefaultTraceLde

There is NO CPU fallback: every call goes to the HIP library and raises if it (or a GPU) is missing.
"""
from ._lib import WfError, Context, default_context, load_library  # noqa: F401

__all__ = ["WfError", "Context", "default_context", "load_library"]
