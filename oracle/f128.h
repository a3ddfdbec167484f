/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see f64.h header note).
 *
 * CPU restatement of the reference's 128-bit field  p = 2^128 - 45 * 2^40 + 1  (canonical u128 values,
 * IS_CANONICAL = true).  Follows /root/reference/math/src/field/f128/mod.rs (lines cited per function).
 */
#ifndef ORACLE_F128_H
#define ORACLE_F128_H
#include <stdint.h>

typedef unsigned __int128 u128;

#define F128_M ((((u128)0xFFFFFFFFFFFFFFFFULL) << 64 | 0xFFFFFFFFFFFFFFFFULL) - ((u128)45 << 40) + 2) /* 2^128 - 45*2^40 + 1, mod.rs:40 */
#define F128_TWO_ADICITY 40
/* G = 23953097886125630542083529559205016746 (2^40-th root of unity), mod.rs:43 */
#define F128_G ((((u128)0x120532e7b364080aULL) << 64) | 0x86b8723e1920f4aaULL)

/* add — mod.rs:410-417 */
static inline u128 f128_add(u128 a, u128 b) {
    u128 z = F128_M - b;
    return a < z ? F128_M - z + a : a - z;
}
/* sub — mod.rs:420-426 */
static inline u128 f128_sub(u128 a, u128 b) { return a < b ? F128_M - b + a : a - b; }

/* helpers — mod.rs:568-618 */
static inline void f128_mul_128x64(u128 a, uint64_t b, uint64_t *r0, uint64_t *r1, uint64_t *r2) {
    u128 z_lo = (u128)(uint64_t)a * (u128)b;
    u128 z_hi = (a >> 64) * (u128)b;
    z_hi = z_hi + (z_lo >> 64);
    *r0 = (uint64_t)z_lo;
    *r1 = (uint64_t)z_hi;
    *r2 = (uint64_t)(z_hi >> 64);
}
static inline void f128_mul_by_modulus(uint64_t a, uint64_t *q0, uint64_t *q1, uint64_t *q2) {
    u128 a_lo = (u128)a * F128_M; /* wrapping_mul */
    *q0 = (uint64_t)a_lo;
    *q1 = (uint64_t)(a_lo >> 64);
    *q2 = a == 0 ? 0 : a - 1;
}
static inline void f128_sub_192x192(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t b0, uint64_t b1, uint64_t b2,
                                    uint64_t *z0o, uint64_t *z1o, uint64_t *z2o) {
    u128 z0 = (u128)a0 - (u128)b0;
    u128 z1 = (u128)a1 - ((u128)b1 + (z0 >> 127));
    u128 z2 = (u128)a2 - ((u128)b2 + (z1 >> 127));
    *z0o = (uint64_t)z0;
    *z1o = (uint64_t)z1;
    *z2o = (uint64_t)z2;
}
static inline void f128_mul_reduce(uint64_t *z0, uint64_t *z1, uint64_t *z2) {
    uint64_t q0, q1, q2;
    f128_mul_by_modulus(*z2, &q0, &q1, &q2);
    f128_sub_192x192(*z0, *z1, *z2, q0, q1, q2, z0, z1, z2);
}
static inline void f128_sub_modulus(uint64_t *lo, uint64_t *hi) {
    u128 z = (u128)0 - F128_M;
    z += (u128)*lo;
    z += (u128)*hi << 64;
    *lo = (uint64_t)z;
    *hi = (uint64_t)(z >> 64);
}

/* mul — mod.rs:429-466 */
static inline u128 f128_mul(u128 a, u128 b) {
    uint64_t x0, x1, x2;
    f128_mul_128x64(a, (uint64_t)(b >> 64), &x0, &x1, &x2);
    f128_mul_reduce(&x0, &x1, &x2);
    if (x2 == 1) f128_sub_modulus(&x0, &x1);
    uint64_t y0, y1, y2;
    f128_mul_128x64(a, (uint64_t)b, &y0, &y1, &y2);
    u128 t = (u128)y1 + (u128)x0;
    y1 = (uint64_t)t;
    uint64_t carry = (uint64_t)(t >> 64);
    t = (u128)y2 + (u128)x1 + (u128)carry;
    y2 = (uint64_t)t;
    uint64_t y3 = (uint64_t)(t >> 64);
    if (y3 == 1) f128_sub_modulus(&y1, &y2);
    uint64_t z0 = y0, z1 = y1, z2 = y2;
    f128_mul_reduce(&z0, &z1, &z2);
    if (z2 == 1 || (z1 == (uint64_t)(F128_M >> 64) && z0 >= (uint64_t)F128_M)) f128_sub_modulus(&z0, &z1);
    return ((u128)z1 << 64) + (u128)z0;
}

static inline u128 f128_exp(u128 base, u128 power) { /* traits.rs:126-149 exp_vartime semantics */
    u128 r = 1, b = base;
    while (power) {
        if (power & 1) r = f128_mul(r, b);
        b = f128_mul(b, b);
        power >>= 1;
    }
    return r;
}
static inline u128 f128_inv(u128 a) { return a == 0 ? 0 : f128_exp(a, F128_M - 2); } /* value of mod.rs:470-563 */
static inline u128 f128_root_of_unity(unsigned n) { return f128_exp(F128_G, (u128)1 << (F128_TWO_ADICITY - n)); }

/* quadratic extension x^2 - x - 1 — mod.rs:267-283 */
static inline void f128_ext2_mul(const u128 a[2], const u128 b[2], u128 out[2]) {
    u128 z = f128_mul(a[0], b[0]);
    u128 o0 = f128_add(z, f128_mul(a[1], b[1]));
    u128 o1 = f128_sub(f128_mul(f128_add(a[0], a[1]), f128_add(b[0], b[1])), z);
    out[0] = o0;
    out[1] = o1;
}
#endif
