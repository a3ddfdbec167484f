"""The slice of the reference's `air` crate the GPU constraint evaluator needs: Assertion::single, the AirContext
arithmetic (ce_blowup_factor, number of composition columns, transition exemptions) and the example AIRs whose
transition functions are built into the library (include/winterfell_hip.h: WF_AIR_FIB_SMALL ... WF_AIR_RESCUE_RAPS), the last one
with an auxiliary trace segment.

Values are python ints in the field's INTERNAL representation (fields.Field.new / as_int)."""
from .math import fields

MIN_BLOWUP_FACTOR = 2          # air/src/options.rs ProofOptions::MIN_BLOWUP_FACTOR


class Assertion:
    """air/src/air/assertions/mod.rs:37-120: Assertion { column, first_step, stride, values }.  single: stride 0, one value;
    periodic: one value every `stride` steps from `first_step`; sequence: values[i] at first_step + i * stride."""

    def __init__(self, column, step, value, stride=0, values=None):
        self.column, self.first_step, self.stride = column, step, stride
        self.values = [value] if values is None else list(values)
        self.value = self.values[0]

    @classmethod
    def single(cls, column, step, value):
        return cls(column, step, value)

    @staticmethod
    def _validate_stride(stride, first_step, column):
        assert stride & (stride - 1) == 0 and stride > 0, "invalid assertion for column %d: stride must be a power of two, but was %d" % (column, stride)
        assert stride >= 2, "invalid assertion for column %d: stride must be at least 2, but was %d" % (column, stride)     # MIN_STRIDE_LENGTH
        assert first_step < stride, "invalid assertion for column %d: first step must be smaller than stride (%d steps), but was %d" % (
            column, stride, first_step)

    @classmethod
    def periodic(cls, column, first_step, stride, value):
        """assertions/mod.rs:84-98"""
        cls._validate_stride(stride, first_step, column)
        return cls(column, first_step, value, stride)

    @classmethod
    def sequence(cls, column, first_step, stride, values):
        """assertions/mod.rs:100-123: a one-value sequence is a single assertion (stride NO_STRIDE)"""
        cls._validate_stride(stride, first_step, column)
        values = list(values)
        assert len(values) > 0, "invalid assertion for column %d: number of asserted values must be greater than zero" % column
        assert len(values) & (len(values) - 1) == 0, "invalid assertion for column %d: number of asserted values must be a power of two, but was %d" % (
            column, len(values))
        return cls(column, first_step, values[0], 0 if len(values) == 1 else stride, values)

    def is_single(self):
        return self.stride == 0

    def is_periodic(self):
        return self.stride != 0 and len(self.values) == 1

    def is_sequence(self):
        return len(self.values) > 1

    def get_num_steps(self, trace_length):
        """assertions/mod.rs:283-297"""
        if self.is_single():
            return 1
        if self.is_periodic():
            assert self.stride <= trace_length, "invalid trace length"
            return trace_length // self.stride
        assert len(self.values) * self.stride == trace_length, "invalid trace length"
        return len(self.values)

    def sort_key(self):
        """Ord for Assertion: stride, then first_step, then column (assertions/mod.rs:301-315)."""
        return (self.stride, self.first_step, self.column)


class TransitionConstraintDegree:
    """air/src/air/transition/degree.rs:30-116."""

    def __init__(self, base, cycles=()):
        assert base > 0, "transition constraint degree must be at least one, but was zero"
        for c in cycles:
            assert c >= 2 and c & (c - 1) == 0
        self.base, self.cycles = base, list(cycles)

    def get_evaluation_degree(self, trace_length):
        return self.base * (trace_length - 1) + sum((trace_length // c) * (c - 1) for c in self.cycles)

    def min_blowup_factor(self):
        bound = self.base + len(self.cycles) - 1
        npo2 = 1 if bound <= 1 else 1 << (bound - 1).bit_length()
        return max(npo2, MIN_BLOWUP_FACTOR)


class _BuiltinAir:
    AIR_ID = None
    TRACE_WIDTH = None
    FIELD = None
    AUX_TRACE_WIDTH = 0            # TraceInfo::aux_segment_width: columns (over E) of the auxiliary segment, 0 = single segment
    NUM_AUX_RANDS = 0              # TraceInfo::get_num_aux_segment_rand_elements

    def __init__(self, trace_length, degrees, num_assertions, blowup_factor, aux_degrees=(), num_aux_assertions=0):
        """AirContext::new / new_multi_segment (air/src/air/context.rs:65-189): main (and auxiliary) transition constraint degrees,
        the number of assertions against each segment."""
        assert trace_length >= 8 and trace_length & (trace_length - 1) == 0       # TraceInfo::MIN_TRACE_LENGTH
        self._n, self.main_degrees, self.aux_degrees, self.blowup_factor = trace_length, list(degrees), list(aux_degrees), blowup_factor
        self.degrees = self.main_degrees + self.aux_degrees
        self._num_main_assertions, self._num_aux_assertions = num_assertions, num_aux_assertions
        self._num_assertions = num_assertions + num_aux_assertions
        if self.AUX_TRACE_WIDTH:
            assert self.aux_degrees and num_aux_assertions, "a multi-segment AIR needs auxiliary constraints and assertions"   # context.rs:131-150
        else:
            assert not self.aux_degrees and not num_aux_assertions
        degrees = self.degrees
        self._ce_blowup = max(d.min_blowup_factor() for d in degrees)               # context.rs:104-117
        assert blowup_factor >= self._ce_blowup, \
            "blowup factor too small; expected at least %d, but was %d" % (self._ce_blowup, blowup_factor)   # context.rs:119-124

    def trace_length(self):
        return self._n

    def ce_blowup_factor(self):
        return self._ce_blowup

    def ce_domain_size(self):
        return self._n * self._ce_blowup

    def lde_domain_size(self):
        return self._n * self.blowup_factor

    def is_multi_segment(self):
        return self.AUX_TRACE_WIDTH > 0

    def trace_width(self):
        """TraceInfo::width: main + auxiliary columns"""
        return self.TRACE_WIDTH + self.AUX_TRACE_WIDTH

    def num_transition_constraints(self):
        return len(self.degrees)

    def num_main_transition_constraints(self):
        return len(self.main_degrees)

    def num_aux_transition_constraints(self):
        return len(self.aux_degrees)

    def num_main_assertions(self):
        return self._num_main_assertions

    def num_aux_assertions(self):
        return self._num_aux_assertions

    def num_assertions(self):
        return self._num_assertions

    def num_transition_exemptions(self):
        return 1                                                                   # context.rs:134

    def num_constraint_composition_columns(self):
        """context.rs:265-285."""
        highest = max(d.get_evaluation_degree(self._n) for d in self.degrees)
        divisor_degree = self._n - self.num_transition_exemptions()
        return max(-(-(highest - divisor_degree) // self._n), 1)

    def sorted_assertions(self):
        """prepare_assertions (air/src/air/boundary/mod.rs:181-208): the order composition coefficients are dealt in."""
        a = sorted(self.get_assertions(), key=Assertion.sort_key)
        assert len(a) == self._num_main_assertions, \
            "expected %d assertions against main trace segment, but received %d" % (self._num_main_assertions, len(a))   # boundary/mod.rs:68-74
        for x in a:
            assert x.column < self.TRACE_WIDTH and x.first_step < self._n
        return a

    def get_aux_assertions(self, aux_rand_elements, ext_degree):
        """Air::get_aux_assertions (air/src/air/mod.rs:271-279): assertions against the auxiliary segment, values in E (tuples
        of ext_degree internal-form ints); none by default."""
        return []

    def sorted_aux_assertions(self, aux_rand_elements, ext_degree):
        a = sorted(self.get_aux_assertions(aux_rand_elements, ext_degree), key=Assertion.sort_key)
        assert len(a) == self._num_aux_assertions, \
            "expected %d assertions against the auxiliary trace segment, but received %d" % (self._num_aux_assertions, len(a))  # boundary/mod.rs:76-82
        for x in a:
            assert x.column < self.AUX_TRACE_WIDTH and x.first_step < self._n and len(x.value) == ext_degree
        return a


class FibSmall(_BuiltinAir):
    """examples/src/fibonacci/fib_small/air.rs:15-68 (the reference instantiates it over f64; the AIR itself is field
    agnostic and so is the kernel)."""
    AIR_ID, TRACE_WIDTH = 0, 2

    def __init__(self, trace_length, result, blowup_factor=8, field=fields.f64):
        super().__init__(trace_length, [TransitionConstraintDegree(1), TransitionConstraintDegree(1)], 3, blowup_factor)
        self.result, self.FIELD = result, field

    def get_assertions(self):
        one = self.FIELD.new(1)
        last = self._n - 1
        return [Assertion.single(0, 0, one), Assertion.single(1, 0, one), Assertion.single(1, last, self.result)]


class RescueAir(_BuiltinAir):
    """examples/src/rescue/air.rs:54-129 (f128; 14 Rescue rounds per 16-step cycle, 9 periodic columns)."""
    AIR_ID, TRACE_WIDTH, CYCLE_LENGTH = 1, 4, 16
    FIELD = fields.f128

    def __init__(self, trace_length, seed, result, blowup_factor=8):
        deg = [TransitionConstraintDegree(3, [self.CYCLE_LENGTH]) for _ in range(4)]
        super().__init__(trace_length, deg, 4, blowup_factor)
        self.seed, self.result = list(seed), list(result)

    def get_assertions(self):
        last = self._n - 1
        return [Assertion.single(0, 0, self.seed[0]), Assertion.single(1, 0, self.seed[1]),
                Assertion.single(0, last, self.result[0]), Assertion.single(1, last, self.result[1])]


class RescueRapsAir(_BuiltinAir):
    """examples/src/rescue_raps/air.rs:60-253 (f128): two Rescue hash chains side by side whose absorbed seeds are permutations
    of each other; the randomised permutation argument lives in an auxiliary segment of three columns over E built from three
    random elements."""
    AIR_ID, TRACE_WIDTH, CYCLE_LENGTH = 7, 8, 16
    AUX_TRACE_WIDTH, NUM_AUX_RANDS = 3, 3                                          # custom_trace_table.rs:93
    FIELD = fields.f128

    def __init__(self, trace_length, result, blowup_factor=8):
        """result: [[a, b], [c, d]] the final rate registers of the two chains (PublicInputs, air.rs:45-53)"""
        main = [TransitionConstraintDegree(3, [self.CYCLE_LENGTH]) for _ in range(8)]
        aux = [TransitionConstraintDegree(1, [self.CYCLE_LENGTH]), TransitionConstraintDegree(1, [self.CYCLE_LENGTH]), TransitionConstraintDegree(2)]
        super().__init__(trace_length, main, 8, blowup_factor, aux_degrees=aux, num_aux_assertions=2)   # new_multi_segment(.., 8, 2, ..), air.rs:78-87
        self.result = [list(result[0]), list(result[1])]

    def pub_inputs_elements(self):
        return [self.result[0][0], self.result[0][1], self.result[1][0], self.result[1][1]]              # flatten_slice_elements

    def get_assertions(self):
        last = self._n - 1
        return [Assertion.single(2, 0, 0), Assertion.single(3, 0, 0), Assertion.single(6, 0, 0), Assertion.single(7, 0, 0),
                Assertion.single(0, last, self.result[0][0]), Assertion.single(1, last, self.result[0][1]),
                Assertion.single(4, last, self.result[1][0]), Assertion.single(5, last, self.result[1][1])]

    def get_aux_assertions(self, aux_rand_elements, ext_degree):
        one = (self.FIELD.new(1),) + (0,) * (ext_degree - 1)                        # E::ONE, air.rs:236-239
        return [Assertion.single(2, 0, one), Assertion.single(2, self._n - 1, one)]


class Fib8(_BuiltinAir):
    """examples/src/fibonacci/fib8/air.rs:18-77: two registers, eight Fibonacci terms per step; the trace starts at the 7th
    and 8th terms (13, 21)."""
    AIR_ID, TRACE_WIDTH = 2, 2

    def __init__(self, trace_length, result, blowup_factor=8, field=fields.f128):
        super().__init__(trace_length, [TransitionConstraintDegree(1), TransitionConstraintDegree(1)], 3, blowup_factor)
        self.result, self.FIELD = result, field

    def get_assertions(self):
        f = self.FIELD
        return [Assertion.single(0, 0, f.new(13)), Assertion.single(1, 0, f.new(21)), Assertion.single(1, self._n - 1, self.result)]


class MulFib2(_BuiltinAir):
    """examples/src/fibonacci/mulfib2/air.rs:18-72: multiplicative Fibonacci, two registers, degree-2 constraints."""
    AIR_ID, TRACE_WIDTH = 3, 2

    def __init__(self, trace_length, result, blowup_factor=8, field=fields.f128):
        super().__init__(trace_length, [TransitionConstraintDegree(2), TransitionConstraintDegree(2)], 3, blowup_factor)
        self.result, self.FIELD = result, field

    def get_assertions(self):
        f = self.FIELD
        return [Assertion.single(0, 0, f.new(1)), Assertion.single(1, 0, f.new(2)), Assertion.single(0, self._n - 1, self.result)]


class MulFib8(_BuiltinAir):
    """examples/src/fibonacci/mulfib8/air.rs:18-94: multiplicative Fibonacci over eight registers."""
    AIR_ID, TRACE_WIDTH = 4, 8

    def __init__(self, trace_length, result, blowup_factor=8, field=fields.f128):
        super().__init__(trace_length, [TransitionConstraintDegree(2) for _ in range(8)], 3, blowup_factor)
        self.result, self.FIELD = result, field

    def get_assertions(self):
        f = self.FIELD
        return [Assertion.single(0, 0, f.new(1)), Assertion.single(1, 0, f.new(2)), Assertion.single(6, self._n - 1, self.result)]


class Vdf(_BuiltinAir):
    """examples/src/vdf/regular/air.rs:29-71 and vdf/exempt/air.rs (exempt=True: two transition exemptions, the result
    asserted at the second to last step because the last row holds garbage)."""
    TRACE_WIDTH = 1

    def __init__(self, trace_length, seed, result, blowup_factor=8, exempt=False, field=fields.f128):
        super().__init__(trace_length, [TransitionConstraintDegree(3)], 2, blowup_factor)
        self.seed, self.result, self.exempt, self.FIELD = seed, result, exempt, field
        self.AIR_ID = 6 if exempt else 5

    def num_transition_exemptions(self):
        return 2 if self.exempt else 1                                             # vdf/exempt/air.rs:47-48

    def get_assertions(self):
        last = self._n - 2 if self.exempt else self._n - 1
        return [Assertion.single(0, 0, self.seed), Assertion.single(0, last, self.result)]
