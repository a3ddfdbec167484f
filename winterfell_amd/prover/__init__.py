"""Mirror of the prover-side plug-in surface of the hot path: ColMatrix / RowMatrix (prover/src/matrix) and
DefaultTraceLde + build_trace_commitment (prover/src/trace/trace_lde/default/mod.rs)."""
from .matrix import ColMatrix, RowMatrix, PartitionOptions  # noqa: F401
from .trace_lde import DefaultTraceLde, StarkDomain, build_trace_commitment  # noqa: F401
from .constraint_commitment import CompositionPoly, DefaultConstraintCommitment, build_constraint_commitment  # noqa: F401
from .composer import DeepCompositionPoly, TracePolyTable, composition_poly_ood_frame, evaluate_columns_at  # noqa: F401
from .constraints import ConstraintCompositionCoefficients, DefaultConstraintEvaluator  # noqa: F401
from .channel import ProofOptions, ProverChannel  # noqa: F401
from .prove import Proof, prove  # noqa: F401
