"""Instruction budget of a 32-point register step for the f64 limb DFT, against the 16-point step the passes use (VERDICT r5 item 5:
"build the one lever left or close the file"; kill criterion: more than 290 VALU instructions per element and 2^24 transform).

The planner of csrc/l24.cuh (make_plan: decimation-in-frequency butterflies over four 24-bit limbs, T = 2^24, T^4 = -1, sign / rotation /
known-zero bookkeeping at compile time) is restated here for N = 2^k points with omega_N = 2^(192 / N): for N = 16 the only sub-limb
twiddle is omega_16 = 2^12 (half-limb shift), for N = 32 it is omega_32 = 2^6 — shifts by 6, 12 or 18 bits inside a limb.  The script
counts machine instructions the way the device code issues them (add / sub 1; a sub-limb shift of a limb 3 = and + shift + shift-add, 2 when
the limb below is known zero, 1 when the limb itself is) and tracks the limb magnitudes the bias of the exit must absorb.

  python tools/dft32_budget.py

N = 16 must reproduce the planner's own numbers (222 operations per 16-point DFT, l24.cuh) — the check that the restatement is the planner."""

MAX_MAG = 0x1fdfe000 - 1          # l24.cuh: what the bias vector of the exit absorbs (limbs stay below 2^30 after the bias)


def plan(logn, signed_split=False):
    n = 1 << logn
    unit = 192 // n                      # omega_N = 2^unit
    # limb state: (zero, mag)
    lim = [[(False, (1 << 23) if signed_split else (1 << 24) - 1), (False, (1 << 23) if signed_split else (1 << 24) - 1),
            (False, (1 << 15) + 1 if signed_split else (1 << 16) - 1), (signed_split is False, 1 if signed_split else 0)] for _ in range(n)]
    if signed_split:                      # a signed split carries into limb 3 (one bit): it is no longer known zero
        for e in range(n):
            lim[e][3] = (False, 1)
    ops = {"addsub": 0, "shift3": 0, "shift2": 0, "shift1": 0}
    for s in range(logn):
        half = n >> (s + 1)
        for blk in range(0, n, 2 * half):
            for i in range(half):
                e1, e2 = blk + i, blk + i + half
                for k in range(4):
                    (za, ma), (zb, mb) = lim[e1][k], lim[e2][k]
                    if za and zb:
                        pass
                    elif zb:
                        lim[e2][k] = (False, ma)
                    elif za:
                        lim[e1][k] = (False, mb)
                    else:
                        ops["addsub"] += 2
                        lim[e1][k] = lim[e2][k] = (False, ma + mb)
                bits = (i * (n // (2 * half)) * unit)          # twiddle omega_{2 half}^i = 2^bits
                q, sub = (bits // 24) % 4, bits % 24
                if sub:
                    new = []
                    for k in range(4):
                        (zl, ml), (zh, mh) = lim[e2][k], lim[e2][(k + 3) % 4]
                        if zl and zh:
                            new.append((True, 0))
                        elif zh:
                            ops["shift2"] += 1
                            new.append((False, ((1 << (24 - sub)) - 1) << sub))
                        elif zl:
                            ops["shift1"] += 1
                            new.append((False, (mh >> (24 - sub)) + 1))
                        else:
                            ops["shift3"] += 1
                            new.append((False, (((1 << (24 - sub)) - 1) << sub) + (mh >> (24 - sub)) + 1))
                    lim[e2] = new
                if q:
                    lim[e2] = [lim[e2][(k - q) % 4] for k in range(4)]
    max_mag = max(m for e in lim for z, m in e if not z)
    insts = ops["addsub"] + 3 * ops["shift3"] + 2 * ops["shift2"] + ops["shift1"]
    plan_ops = ops["addsub"] + ops["shift3"] + ops["shift2"] + ops["shift1"]
    return {"n": n, "planner_ops": plan_ops, "machine_insts": insts, "per_element": insts / n, "max_mag": max_mag, "fits_bias": max_mag <= MAX_MAG, **ops}


def main():
    p16, p32, p32s = plan(4), plan(5), plan(5, signed_split=True)
    assert p16["planner_ops"] == 222, p16          # l24.cuh: "222 operations per 16-point DFT"
    for name, p in (("16-point", p16), ("32-point", p32), ("32-point, signed split", p32s)):
        print("%-24s planner ops %4d  machine instructions %4d = %5.1f per element; max |limb| 2^%.2f (%s the bias)" % (
            name, p["planner_ops"], p["machine_insts"], p["per_element"], __import__("math").log2(p["max_mag"]), "fits" if p["fits_bias"] else "EXCEEDS"))
    # per element and register step: split into limbs + DFT + bias of the four output limbs + exit (multiply-accumulate + fold)
    split, split_signed, bias, exit_tw, exit_one = 5, 8, 4, 8 + 7, 4 + 7
    step16 = split + p16["per_element"] + bias
    step32 = split_signed + p32s["per_element"] + bias           # the unsigned split does not fit the bias (see above)
    chain = 38                                                   # the two Montgomery products of the inter-pass progression (gl64.cuh: 4 + 3 + 8 + ~4 each)
    other = 114 - (2 * step16 + exit_tw + exit_one + chain)      # addressing, LDS traffic, loads / stores: what is left of the measured 114
    now = 2 * (2 * step16 + exit_tw + exit_one + chain + other) + (2 * step16 + exit_tw + exit_one + (89 - (2 * step16 + exit_tw + exit_one)))
    print("measured today (SQ_INSTS_VALU, profiles/r05/bench_pmc_summary.json): 114 + 114 + 89 = 317 per element and 2^24 transform")
    print("model of a pass with two 16-point steps: 2 x %.1f (split + DFT + bias) + %d + %d (exits) + %d (progression) + %.1f (addressing, LDS, memory) = 114" % (
        step16, exit_tw, exit_one, chain, other))
    # 2^24 = 32 . 32 | 32 . 32 | 16: two radix-1024 passes of two 32-point steps and a one-step radix-16 last pass
    p_a = 2 * step32 + exit_tw + exit_one + chain + other
    p_last = step16 + exit_one + (89 - (2 * step16 + exit_tw + exit_one))
    total32 = 2 * p_a + p_last
    print("plan 10, 10, 4 with 32-point steps: 2 x (2 x %.1f + %d + %d + %d + %.1f) + %.1f = %.0f per element and transform (today %.0f)" % (
        step32, exit_tw, exit_one, chain, other, p_last, total32, now))
    print("kill criterion: > 290 -> %s" % ("KILLED: the step is not built" if total32 > 290 else "build it"))
    return total32


if __name__ == "__main__":
    main()
