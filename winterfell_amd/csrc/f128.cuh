// Device + host arithmetic for the reference's f128 field  p = 2^128 - 45*2^40 + 1  (math/src/field/f128/mod.rs).
// Values are canonical u128 integers (IS_CANONICAL = true, mod.rs:80), so memory images are compared as is.
// The reference reduces with a 128x64 schoolbook scheme (mod.rs:429-466); any exact modmul gives the same canonical
// result, so this one is built for gfx950 (no 64-bit multiplier, one instruction per limb of a carry chain, cf.
// gl64.cuh): everything is written on 32-bit limbs with explicit carry chains.
//   add 13 limb instructions + select, sub 11, mul = 16 + 6 v_mad_u64_u32 and ~50 carry instructions:
//   the 256-bit product is folded with 2^128 = C (mod p), C = 45 * 2^40 - 1, i.e.  hi * C = ((hi * 45) << 40) - hi
//   (a 32-bit multiply by 45 per limb instead of a 128 x 46-bit product), twice.
// tools/microbench_f128.hip: mul 542 -> 310, add 90 -> 40, sub 79 -> 47 cycles per wave-op against the previous
// __int128 formulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace f128 {

typedef unsigned __int128 u128;
typedef uint32_t u32;
typedef uint64_t u64;

__host__ __device__ constexpr u128 modulus() { return ((u128)0xFFFFFFFFFFFFFFFFull << 64 | 0xFFFFFFFFFFFFFFFFull) - ((u128)45 << 40) + 2; }
constexpr u32 C0 = 0xFFFFFFFFu, C1 = 0x2CFFu;   // C = 45 * 2^40 - 1 = 2^128 mod p, as limbs (C2 = C3 = 0)

#define F128_HD __host__ __device__ __forceinline__

F128_HD u32 funnel(u32 hi, u32 lo, int s) {      // low 32 bits of (hi:lo) >> s, 0 < s < 32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (u32)((((u64)hi << 32) | lo) >> s);
#endif
}

F128_HD void split(u128 a, u32 (&l)[4]) {
    l[0] = (u32)a;
    l[1] = (u32)(a >> 32);
    l[2] = (u32)(a >> 64);
    l[3] = (u32)(a >> 96);
}
F128_HD u128 join(u32 l0, u32 l1, u32 l2, u32 l3) { return ((u128)(((u64)l3 << 32) | l2) << 64) | (((u64)l1 << 32) | l0); }

// canonical value of s + carry * 2^128 when that is below 2p: t = s + C; the answer is t if the sum had already carried
// out (then t cannot wrap) or if adding C wraps (s >= p), else s
F128_HD u128 finish(u32 s0, u32 s1, u32 s2, u32 s3, u32 carry) {
    u32 k;
    const u32 t0 = __builtin_addc(s0, C0, 0u, &k);
    const u32 t1 = __builtin_addc(s1, C1, k, &k);
    const u32 t2 = __builtin_addc(s2, 0u, k, &k);
    const u32 t3 = __builtin_addc(s3, 0u, k, &k);
    const bool use = (carry | k) != 0;
    return use ? join(t0, t1, t2, t3) : join(s0, s1, s2, s3);
}

F128_HD u128 add(u128 a, u128 b) {
    u32 x[4], y[4], c;
    split(a, x);
    split(b, y);
    const u32 s0 = __builtin_addc(x[0], y[0], 0u, &c);
    const u32 s1 = __builtin_addc(x[1], y[1], c, &c);
    const u32 s2 = __builtin_addc(x[2], y[2], c, &c);
    const u32 s3 = __builtin_addc(x[3], y[3], c, &c);
    return finish(s0, s1, s2, s3, c);
}

F128_HD u128 sub(u128 a, u128 b) {
    u32 x[4], y[4], bw, k;
    split(a, x);
    split(b, y);
    const u32 d0 = __builtin_subc(x[0], y[0], 0u, &bw);
    const u32 d1 = __builtin_subc(x[1], y[1], bw, &bw);
    const u32 d2 = __builtin_subc(x[2], y[2], bw, &bw);
    const u32 d3 = __builtin_subc(x[3], y[3], bw, &bw);
    const u32 m = 0u - bw;                         // a < b: + p, which is - C modulo 2^128
    const u32 r0 = __builtin_subc(d0, m & C0, 0u, &k);
    const u32 r1 = __builtin_subc(d1, m & C1, k, &k);
    const u32 r2 = __builtin_subc(d2, 0u, k, &k);
    const u32 r3 = __builtin_subc(d3, 0u, k, &k);
    return join(r0, r1, r2, r3);
}

// r[0..4] = a * (b3:b2:b1:b0): four 32x32+64 multiply-adds, the high word of each feeding the next
F128_HD void mul_row(u32 a, const u32 (&b)[4], u32 (&r)[5]) {
    u64 t = (u64)a * b[0];
    r[0] = (u32)t;
    t = (u64)a * b[1] + (t >> 32);
    r[1] = (u32)t;
    t = (u64)a * b[2] + (t >> 32);
    r[2] = (u32)t;
    t = (u64)a * b[3] + (t >> 32);
    r[3] = (u32)t;
    r[4] = (u32)(t >> 32);
}

F128_HD u128 mul(u128 a, u128 b) {
    u32 x[4], y[4], P[8], r[5], c, bw, k;
    split(a, x);
    split(b, y);
    // ---- 4 x 4 schoolbook product P[0..7], one row of a at a time
    mul_row(x[0], y, r);
    P[0] = r[0]; P[1] = r[1]; P[2] = r[2]; P[3] = r[3]; P[4] = r[4];
    mul_row(x[1], y, r);
    P[1] = __builtin_addc(P[1], r[0], 0u, &c);
    P[2] = __builtin_addc(P[2], r[1], c, &c);
    P[3] = __builtin_addc(P[3], r[2], c, &c);
    P[4] = __builtin_addc(P[4], r[3], c, &c);
    P[5] = __builtin_addc(r[4], 0u, c, &c);
    mul_row(x[2], y, r);
    P[2] = __builtin_addc(P[2], r[0], 0u, &c);
    P[3] = __builtin_addc(P[3], r[1], c, &c);
    P[4] = __builtin_addc(P[4], r[2], c, &c);
    P[5] = __builtin_addc(P[5], r[3], c, &c);
    P[6] = __builtin_addc(r[4], 0u, c, &c);
    mul_row(x[3], y, r);
    P[3] = __builtin_addc(P[3], r[0], 0u, &c);
    P[4] = __builtin_addc(P[4], r[1], c, &c);
    P[5] = __builtin_addc(P[5], r[2], c, &c);
    P[6] = __builtin_addc(P[6], r[3], c, &c);
    P[7] = __builtin_addc(r[4], 0u, c, &c);
    // ---- first fold: hi * C = ((hi * 45) << 40) - hi with hi = P[4..7] < p.  The shift by 40 is a limb (32) and 8 bits: the 8 bits
    // ride on the multiplier, u = hi * (45 << 8) (5 limbs, u4 < 2^14), so that v = u << 32 needs no funnel shifts (round 4: 7
    // instructions of 74 with the second fold's two)
    constexpr u32 K = 45u << 8;
    u32 u[5];
    {
        u64 t = (u64)P[4] * K;
        u[0] = (u32)t;
        t = (u64)P[5] * K + (t >> 32);
        u[1] = (u32)t;
        t = (u64)P[6] * K + (t >> 32);
        u[2] = (u32)t;
        t = (u64)P[7] * K + (t >> 32);
        u[3] = (u32)t;
        u[4] = (u32)(t >> 32);
    }
    // v = u << 32 (limb 0 is zero), w = v - hi >= 0: six limbs, w5 < 2^14
    const u32 v1 = u[0], v2 = u[1], v3 = u[2], v4 = u[3], v5 = u[4];
    const u32 w0 = __builtin_subc(0u, P[4], 0u, &bw);
    const u32 w1 = __builtin_subc(v1, P[5], bw, &bw);
    const u32 w2 = __builtin_subc(v2, P[6], bw, &bw);
    const u32 w3 = __builtin_subc(v3, P[7], bw, &bw);
    const u32 w4 = __builtin_subc(v4, 0u, bw, &bw);
    const u32 w5 = __builtin_subc(v5, 0u, bw, &bw);
    // ---- r = lo + w[0..3]; what is left above 2^128 is T = (w5:w4) + carry < 2^47
    u32 r0 = __builtin_addc(P[0], w0, 0u, &c);
    u32 r1 = __builtin_addc(P[1], w1, c, &c);
    u32 r2 = __builtin_addc(P[2], w2, c, &c);
    u32 r3 = __builtin_addc(P[3], w3, c, &c);
    const u32 T0 = __builtin_addc(w4, 0u, c, &k);
    const u32 T1 = w5 + k;
    // ---- second fold: T * C = ((T * 45) << 40) - T < 2^93, the same way: T * (45 << 8) < 2^61 is two limbs
    u64 t = (u64)T0 * K;
    const u32 x0 = (u32)t;
    t = (u64)T1 * K + (t >> 32);
    const u32 x1 = (u32)t;
    const u32 y1 = x0, y2 = x1;                                    // (T * 45) << 40: limbs 1 and 2, limb 3 is zero
    const u32 z0 = __builtin_subc(0u, T0, 0u, &bw);
    const u32 z1 = __builtin_subc(y1, T1, bw, &bw);
    const u32 z2 = __builtin_subc(y2, 0u, bw, &bw);
    r0 = __builtin_addc(r0, z0, 0u, &c);
    r1 = __builtin_addc(r1, z1, c, &c);
    r2 = __builtin_addc(r2, z2, c, &c);
    r3 = __builtin_addc(r3, 0u, c, &c);
    // r + c * 2^128 < 2^128 + 2^93: one conditional subtraction of p
    return finish(r0, r1, r2, r3, c);
}

// a * w for a TABLE constant w given as the pair (w, w64 = w * 2^64 mod p) — the twiddle tables of the NTT passes (round 5):
//   a * w = a_lo * w + a_hi * w64   (mod p),   a = a_lo + 2^64 a_hi
// two 64 x 128-bit products whose sum is below 2^193, so ONE fold of at most 65 bits replaces the two folds of a 256-bit product:
// 16 + 3 multiply-adds and ~35 carry instructions instead of 16 + 6 and ~50 (mul above).  Exact: the same canonical result.
F128_HD u128 mul_tab(u128 a, u128 w, u128 w64) {
    u32 x[4], y[4], z[4], P[7], r[5], c, bw;
    split(a, x);
    split(w, y);
    split(w64, z);
    mul_row(x[0], y, r);
    P[0] = r[0]; P[1] = r[1]; P[2] = r[2]; P[3] = r[3]; P[4] = r[4];
    mul_row(x[1], y, r);
    P[1] = __builtin_addc(P[1], r[0], 0u, &c);
    P[2] = __builtin_addc(P[2], r[1], c, &c);
    P[3] = __builtin_addc(P[3], r[2], c, &c);
    P[4] = __builtin_addc(P[4], r[3], c, &c);
    P[5] = __builtin_addc(r[4], 0u, c, &c);                      // a_lo * w < 2^192: six limbs
    mul_row(x[2], z, r);
    P[0] = __builtin_addc(P[0], r[0], 0u, &c);
    P[1] = __builtin_addc(P[1], r[1], c, &c);
    P[2] = __builtin_addc(P[2], r[2], c, &c);
    P[3] = __builtin_addc(P[3], r[3], c, &c);
    P[4] = __builtin_addc(P[4], r[4], c, &c);
    P[5] = __builtin_addc(P[5], 0u, c, &c);
    P[6] = c;
    mul_row(x[3], z, r);
    P[1] = __builtin_addc(P[1], r[0], 0u, &c);
    P[2] = __builtin_addc(P[2], r[1], c, &c);
    P[3] = __builtin_addc(P[3], r[2], c, &c);
    P[4] = __builtin_addc(P[4], r[3], c, &c);
    P[5] = __builtin_addc(P[5], r[4], c, &c);
    P[6] += c;                                                   // the sum is below 2^193: P[6] <= 1
    // ---- one fold: hi = P[4..6] < 2^65,  hi * C = ((hi * 45) << 40) - hi = ((hi * K) << 32) - hi with K = 45 << 8
    constexpr u32 K = 45u << 8;
    u64 t = (u64)P[4] * K;
    const u32 u0 = (u32)t;
    t = (u64)P[5] * K + (t >> 32);
    const u32 u1 = (u32)t;
    t = (u64)P[6] * K + (t >> 32);
    const u32 u2 = (u32)t;                                       // hi * K < 2^79: three limbs
    // w = (u << 32) - hi >= 0, below 2^111: limbs 0..3
    const u32 w0 = __builtin_subc(0u, P[4], 0u, &bw);
    const u32 w1 = __builtin_subc(u0, P[5], bw, &bw);
    const u32 w2 = __builtin_subc(u1, P[6], bw, &bw);
    const u32 w3 = __builtin_subc(u2, 0u, bw, &bw);
    const u32 r0 = __builtin_addc(P[0], w0, 0u, &c);
    const u32 r1 = __builtin_addc(P[1], w1, c, &c);
    const u32 r2 = __builtin_addc(P[2], w2, c, &c);
    const u32 r3 = __builtin_addc(P[3], w3, c, &c);
    // r + c * 2^128 < 2^128 + 2^111: one conditional subtraction of p
    return finish(r0, r1, r2, r3, c);
}

#undef F128_HD

}  // namespace f128
