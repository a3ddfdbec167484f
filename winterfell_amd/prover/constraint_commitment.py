"""CompositionPoly + DefaultConstraintCommitment (prover/src/constraints/composition_poly.rs:52-140,
prover/src/constraints/commitment/default.rs:26-150) on the GPU."""
import numpy as np

from ..math import fft, fields
from .matrix import ColMatrix, PartitionOptions
from .trace_lde import build_trace_commitment


class CompositionPoly:
    """Column polynomials of the constraint composition polynomial (composition_poly.rs:52-78)."""

    def __init__(self, data: ColMatrix):
        self.data = data

    @classmethod
    def new(cls, composition_trace, domain, num_cols, ext_degree=1, field=fields.f64, ctx=None):
        """composition_trace: evaluations of the combined constraint polynomial over the constraint-evaluation domain
        (ce_n * ext_degree elements).  Interpolates over the coset (fft::interpolate_poly_with_offset,
        composition_poly.rs:72-73) and splits the coefficients into num_cols chunks of trace_length (`segment`, :128-140)."""
        from .._lib import default_context
        ctx = ctx or default_context()
        tr = ctx.to_device(composition_trace) if isinstance(composition_trace, np.ndarray) else composition_trace.clone()
        tr = tr.reshape(-1)
        ce_n = tr.numel() // (ext_degree * field.W)
        assert domain.trace_length < ce_n, "trace length must be smaller than length of composition polynomial trace"   # :63-66
        coeffs = fft.interpolate_poly_with_offset(tr, None, domain.offset, ext_degree=ext_degree, ctx=ctx, field=field)
        n = domain.trace_length
        words = n * ext_degree * field.W
        assert num_cols * n <= ce_n
        cols = coeffs[: num_cols * words].reshape(num_cols, words)      # chunks(trace_len).take(num_cols)
        return cls(ColMatrix(cols, ext_degree, ctx, field))

    def num_columns(self):
        return self.data.num_cols()

    def column_len(self):
        return self.data.num_rows()

    def column_degree(self):
        return self.column_len() - 1


class DefaultConstraintCommitment:
    """ConstraintCommitment (prover/src/constraints/commitment/mod.rs:32-36, default.rs:26-107)."""

    def __init__(self, evaluations, vector_commitment):
        self.evaluations, self.vector_commitment = evaluations, vector_commitment

    def commitment(self):
        return self.vector_commitment.root()

    def query(self, positions):
        return self.evaluations.rows(positions), self.vector_commitment.prove_batch(list(positions))


def build_constraint_commitment(hasher, composition_trace, num_cols, domain, partition_options=None, ext_degree=1,
                                field=fields.f64, ctx=None, fetch_root=True):
    """build_constraint_commitment (commitment/default.rs:109-150) -> (DefaultConstraintCommitment, CompositionPoly)."""
    poly = CompositionPoly.new(composition_trace, domain, num_cols, ext_degree, field, ctx)
    assert poly.num_columns() == num_cols and poly.column_degree() == domain.trace_length - 1
    # evaluate_composition_poly_columns + compute_constraint_evaluation_commitment: the columns already are polynomials
    lde, tree, _ = build_trace_commitment(hasher, poly.data, domain, partition_options or PartitionOptions(), skip_interpolate=True,
                                          fetch_root=fetch_root)
    assert lde.num_cols() == num_cols and lde.num_rows() == domain.lde_domain_size()
    return DefaultConstraintCommitment(lde, tree), poly
