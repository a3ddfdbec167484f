"""Mirror of the reference's `crypto` crate surface on the hot path: Hasher / ElementHasher implementations
(Blake3_256, Rp64_256) and MerkleTree (crypto/src/hash/mod.rs:31-80, crypto/src/merkle/mod.rs)."""
from .hash import Blake3_192, Blake3_256, Rp62_248, Rp64_256, RpJive64_256, Sha3_256  # noqa: F401
from .merkle import MerkleTree, MerkleTreeError, BatchMerkleProof  # noqa: F401
from .random import DefaultRandomCoin, check_leading_zeros, grind_query_seed  # noqa: F401
