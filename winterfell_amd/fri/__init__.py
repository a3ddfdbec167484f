"""Mirror of the `fri` crate's prover side (fri/src/prover/mod.rs, fri/src/prover/channel.rs, fri/src/folding/mod.rs, fri/src/proof.rs,
fri/src/options.rs)."""
from .channel import DefaultProverChannel  # noqa: F401
from .prover import FriLayer, FriOptions, FriProof, FriProofLayer, FriProver, apply_drp, fold_positions  # noqa: F401
