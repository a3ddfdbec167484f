"""Verifier-side formulas shared by the tests (verifier/src/evaluator.rs:16-89, verifier/src/lib.rs ood consistency check),
written over python ints with the oracle's scalar field functions — independent of the prover-side code paths they check."""


class Ext:
    """Degree-D extension elements as lists of internal-form python ints, on top of an oracle GenericField."""

    def __init__(self, fld, D, one=None):
        self.f, self.D, self.one = fld, D, one

    def add(self, a, b):
        return [self.f.add(x, y) for x, y in zip(a, b)]

    def sub(self, a, b):
        return [self.f.sub(x, y) for x, y in zip(a, b)]

    def mul(self, a, b):
        return self.f.ext_mul(self.D, a, b)

    def lift(self, v):
        return [v] + [0] * (self.D - 1)

    def horner(self, coeffs, x):
        acc = [0] * self.D
        for c in reversed(coeffs):
            acc = self.add(self.mul(acc, x), c)
        return acc

    def pow(self, a, e):
        r = self.lift(self.one)
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.mul(a, a)
            e >>= 1
        return r


def ood_constraint_equation_holds(E, one, g, n, z, H, transition_evals, cc_transition, ood_cur, assertions, cc_boundary, num_exemptions=1):
    """H(z) == sum_k cc_k C_k(z) / Z_t(z) + sum_groups sum_a cc_a (T_col(z) - value) / (z - g^step), cross-multiplied so no
    extension-field inversion is needed.  All arguments are lists of internal-form ints:
      H: the composition polynomial's value at z;  transition_evals: nt elements (each D ints) = Air::evaluate_transition on
      the OOD frame;  ood_cur: width elements;  assertions: [(column, step, value int)];  g: trace-domain generator."""
    f = E.f
    T = [0] * E.D
    for cc, ev in zip(cc_transition, transition_evals):
        T = E.add(T, E.mul(cc, ev))
    zn = E.pow(z, n)
    num_t = E.sub(zn, E.lift(one))                                   # x^n - 1
    den_t = E.sub(z, E.lift(f.exp(g, n - 1)))                        # transition exemptions: x - g^(n-1) [, x - g^(n-2)] (divisor.rs:43-51)
    for k in range(2, num_exemptions + 1):
        den_t = E.mul(den_t, E.sub(z, E.lift(f.exp(g, n - k))))
    groups = {}
    for (col, step, val), cc in zip(assertions, cc_boundary):
        ev = E.sub(ood_cur[col], E.lift(val))
        groups[step] = E.add(groups.get(step, [0] * E.D), E.mul(cc, ev))
    divs = {step: E.sub(z, E.lift(f.exp(g, step))) for step in groups}
    prod_all = E.lift(one)
    for d in divs.values():
        prod_all = E.mul(prod_all, d)
    lhs = E.mul(E.mul(H, num_t), prod_all)
    rhs = E.mul(E.mul(T, den_t), prod_all)
    for step, B in groups.items():
        other = E.lift(one)
        for s2, d in divs.items():
            if s2 != step:
                other = E.mul(other, d)
        rhs = E.add(rhs, E.mul(E.mul(B, num_t), other))
    return lhs == rhs
