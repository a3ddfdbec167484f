"""Mirror of the `fri` crate's prover side (fri/src/prover/mod.rs, fri/src/folding/mod.rs, fri/src/proof.rs, fri/src/options.rs)."""
from .prover import FriLayer, FriOptions, FriProof, FriProofLayer, FriProver, apply_drp, fold_positions  # noqa: F401
