"""DefaultConstraintEvaluator (prover/src/constraints/evaluator/default.rs:30-210) for the built-in AIRs: one library
call evaluates, divides and combines all constraints over the constraint-evaluation domain, reading frames from the
device-resident trace LDE."""
import ctypes

import numpy as np

from .._lib import ptr


class ConstraintCompositionCoefficients:
    """air/src/air/coefficients.rs: {transition: Vec<E>, boundary: Vec<E>} as (k, ext_degree*W) word arrays."""

    def __init__(self, transition, boundary):
        self.transition = np.ascontiguousarray(transition, dtype=np.uint64)
        self.boundary = np.ascontiguousarray(boundary, dtype=np.uint64)


class DefaultConstraintEvaluator:
    def __init__(self, air, composition_coefficients: ConstraintCompositionCoefficients, ext_degree=1):
        f = air.FIELD
        self.air, self.ext_degree, self.cc = air, ext_degree, composition_coefficients
        ew = ext_degree * f.W
        assert self.cc.transition.size == air.num_transition_constraints() * ew, \
            "number of transition constraints must match the number of composition coefficient tuples"   # transition/mod.rs:37-41
        assert self.cc.boundary.size == air.num_assertions() * ew, \
            "number of assertions must match the number of composition coefficient tuples"               # boundary/mod.rs:62-66
        self.assertions = air.sorted_assertions()      # boundary coefficient k belongs to sorted assertion k

    def evaluate(self, trace_lde, domain):
        """ConstraintEvaluator::evaluate (default.rs:52-106) -> CompositionPolyTrace: ce_domain_size elements on the
        device (flat uint64 tensor)."""
        air, f, D = self.air, self.air.FIELD, self.ext_degree
        lde = trace_lde.main_segment_lde
        assert lde.num_rows() == domain.lde_domain_size() == air.lde_domain_size(), \
            "extended trace length is not consistent with evaluation domain"                             # default.rs:57-61
        n = air.trace_length()
        ctx = lde.ctx
        out = ctx.empty_u64(air.ce_domain_size() * D * f.W)
        cols = np.array([a.column for a in self.assertions], dtype=np.uint32)
        steps = np.array([a.first_step for a in self.assertions], dtype=np.uint64)
        vals = f.pack([a.value for a in self.assertions])
        off = f.element_words(int(domain.offset))
        cct, ccb = self.cc.transition.reshape(-1), self.cc.boundary.reshape(-1)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        ctx.call("wf_evaluate_constraints", air.AIR_ID, f.ID, D, ptr(lde.data), lde.row_width, n.bit_length() - 1,
                 domain.blowup.bit_length() - 1, air.ce_blowup_factor().bit_length() - 1, vp(off), vp(cct), len(self.assertions),
                 vp(cols), vp(steps), vp(vals), vp(ccb), ptr(out))
        return out
