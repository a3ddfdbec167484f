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
    def __init__(self, air, composition_coefficients: ConstraintCompositionCoefficients, ext_degree=1, aux_rand_elements=None):
        """DefaultConstraintEvaluator::new(air, aux_rand_elements, composition_coefficients) (default.rs:128-160);
        aux_rand_elements: (NUM_AUX_RANDS, ext_degree*W) words for a multi-segment AIR."""
        f = air.FIELD
        self.air, self.ext_degree, self.cc = air, ext_degree, composition_coefficients
        assert (aux_rand_elements is not None) == air.is_multi_segment(), "expected aux rand elements to be present"   # default.rs:322-324
        self.aux_rand_elements = None if aux_rand_elements is None else np.ascontiguousarray(aux_rand_elements, dtype=np.uint64)
        self.aux_assertions = air.sorted_aux_assertions(self.aux_rand_elements, ext_degree) if air.is_multi_segment() else []
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
        if air.is_multi_segment():
            # evaluate_fragment_full (default.rs:214-271): frames from both segments; boundary coefficients: main assertions first
            aux = trace_lde.aux_segment_lde
            assert aux is not None, "expected aux segment to be present"
            ew = D * f.W
            nm = len(self.assertions)
            xcols = np.array([a.column for a in self.aux_assertions], dtype=np.uint32)
            xsteps = np.array([a.first_step for a in self.aux_assertions], dtype=np.uint64)
            xvals = f.pack([v for a in self.aux_assertions for v in a.value])
            ccm, ccx = np.ascontiguousarray(ccb[:nm * ew]), np.ascontiguousarray(ccb[nm * ew:])
            rnd = self.aux_rand_elements.reshape(-1)
            assert rnd.size == air.NUM_AUX_RANDS * ew
            ctx.call("wf_evaluate_constraints_aux", air.AIR_ID, f.ID, D, ptr(lde.data), lde.row_width, ptr(aux.data), aux.row_width,
                     n.bit_length() - 1, domain.blowup.bit_length() - 1, air.ce_blowup_factor().bit_length() - 1, vp(off), vp(cct), nm, vp(cols),
                     vp(steps), vp(vals), vp(ccm), len(self.aux_assertions), vp(xcols), vp(xsteps), vp(xvals), vp(ccx), vp(rnd), ptr(out))
            return out
        if any(a.stride for a in self.assertions):
            # periodic / sequence assertions (BoundaryConstraint::new for multi-value assertions, air/src/air/boundary/constraint.rs:60-91):
            # strides, value counts and all the values back to back
            for a in self.assertions:
                a.get_num_steps(n)                                          # validate_trace_length
            strides = np.array([a.stride for a in self.assertions], dtype=np.uint64)
            nvals = np.array([len(a.values) for a in self.assertions], dtype=np.uint64)
            allv = f.pack([v for a in self.assertions for v in a.values])
            ctx.call("wf_evaluate_constraints_assertions", air.AIR_ID, f.ID, D, ptr(lde.data), lde.row_width, n.bit_length() - 1,
                     domain.blowup.bit_length() - 1, air.ce_blowup_factor().bit_length() - 1, vp(off), vp(cct), len(self.assertions),
                     vp(cols), vp(steps), vp(strides), vp(nvals), vp(allv), vp(ccb), ptr(out))
            return out
        ctx.call("wf_evaluate_constraints", air.AIR_ID, f.ID, D, ptr(lde.data), lde.row_width, n.bit_length() - 1,
                 domain.blowup.bit_length() - 1, air.ce_blowup_factor().bit_length() - 1, vp(off), vp(cct), len(self.assertions),
                 vp(cols), vp(steps), vp(vals), vp(ccb), ptr(out))
        return out


def evaluate_constraints_dev(air, trace_lde, domain, d_cc, ext_degree=1):
    """DefaultConstraintEvaluator::evaluate for a single-segment AIR with the composition coefficients where a DEVICE coin drew them:
    d_cc = (num_transition_constraints + num_assertions) elements in draw order (transition first; boundary coefficient k belongs to
    sorted assertion k, air/src/air/mod.rs:529-560).  One library call (wf_evaluate_constraints_dev), nothing waits for the stream.
    Returns (CompositionPolyTrace on the device, the sorted assertions)."""
    assert not air.is_multi_segment(), "the auxiliary segment's random elements are the caller's (host) input"
    f, D = air.FIELD, ext_degree
    lde = trace_lde.main_segment_lde
    assert lde.num_rows() == domain.lde_domain_size() == air.lde_domain_size(), \
        "extended trace length is not consistent with evaluation domain"                                 # default.rs:57-61
    n, ctx = air.trace_length(), lde.ctx
    ew = D * f.W
    nt, assertions = air.num_transition_constraints(), air.sorted_assertions()
    assert d_cc.numel() == (nt + len(assertions)) * ew
    d_cc = d_cc.reshape(-1)
    out = ctx.empty_u64(air.ce_domain_size() * ew)
    for a in assertions:
        if a.stride:
            a.get_num_steps(n)                                                  # validate_trace_length
    multi = any(a.stride for a in assertions)
    cols = np.array([a.column for a in assertions], dtype=np.uint32)
    steps = np.array([a.first_step for a in assertions], dtype=np.uint64)
    strides = np.array([a.stride for a in assertions], dtype=np.uint64)
    nvals = np.array([len(a.values) for a in assertions], dtype=np.uint64)
    vals = f.pack([v for a in assertions for v in a.values]) if multi else f.pack([a.value for a in assertions])
    off = f.element_words(int(domain.offset))
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ctx.call("wf_evaluate_constraints_dev", air.AIR_ID, f.ID, D, ptr(lde.data), lde.row_width, n.bit_length() - 1, domain.blowup.bit_length() - 1,
             air.ce_blowup_factor().bit_length() - 1, vp(off), ptr(d_cc), len(assertions), vp(cols), vp(steps), vp(strides) if multi else None,
             vp(nvals) if multi else None, vp(vals), ptr(d_cc[nt * ew:]), ptr(out))
    return out, assertions
