"""Mirror of the `fri` crate's prover-side commit phase (fri/src/prover/mod.rs, fri/src/options.rs)."""
from .prover import FriOptions, FriProver, FriLayer  # noqa: F401
