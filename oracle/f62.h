/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see f64.h header note).
 *
 * CPU restatement of the reference's 62-bit field  p = 2^62 - 111 * 2^39 + 1  in Montgomery form (R = 2^64) whose
 * internal values live in the LAZY range [0, 2M) (math/src/field/f62/mod.rs:61).  The reference itself only ever
 * compares / serialises normalised values (mod.rs:206-211, 237-244), so parity for this field is defined on
 * normalize(x); the oracle's exported functions return normalised words.
 */
#ifndef ORACLE_F62_H
#define ORACLE_F62_H
#include <stdint.h>
typedef unsigned __int128 u128;

#define F62_M 4611624995532046337ULL   /* mod.rs:39 */
#define F62_R2 630444561284293700ULL   /* mod.rs:42 */
#define F62_U 4611624995532046335ULL   /* -M^-1 mod 2^64, mod.rs:48 */
#define F62_G 4421547261963328785ULL   /* 2^39-th root of unity (canonical), mod.rs:54 */
#define F62_TWO_ADICITY 39

static inline uint64_t f62_add(uint64_t a, uint64_t b) { uint64_t z = a + b; return z - (z >> 62) * F62_M; }            /* :539-543 */
static inline uint64_t f62_sub(uint64_t a, uint64_t b) { return a < b ? 2 * F62_M - b + a : a - b; }                     /* :548-554 */
static inline uint64_t f62_mul(uint64_t a, uint64_t b) {                                                               /* :559-564 */
    u128 z = (u128)a * (u128)b;
    uint64_t q = (uint64_t)((u128)(uint64_t)z * (u128)F62_U);
    z = z + (u128)q * (u128)F62_M;
    return (uint64_t)(z >> 64);
}
static inline uint64_t f62_normalize(uint64_t v) { return v >= F62_M ? v - F62_M : v; }                                  /* :621-627 */
static inline uint64_t f62_new(uint64_t v) { return f62_mul(v >= F62_M ? v % F62_M : v, F62_R2); }                       /* :96-104: new() reduces then converts */
static inline uint64_t f62_as_int(uint64_t a) { return f62_normalize(f62_mul(a, 1)); }                                   /* :237-244 */
static inline uint64_t f62_exp(uint64_t base, uint64_t power) {
    uint64_t r = f62_new(1), b = base;
    while (power) {
        if (power & 1) r = f62_mul(r, b);
        b = f62_mul(b, b);
        power >>= 1;
    }
    return r;
}
static inline uint64_t f62_inv(uint64_t a) { return f62_normalize(a) == 0 ? 0 : f62_exp(a, F62_M - 2); }
static inline uint64_t f62_root_of_unity(unsigned n) { return f62_exp(f62_new(F62_G), 1ULL << (F62_TWO_ADICITY - n)); }

/* quadratic extension x^2 - x - 1 — mod.rs:321-326 */
static inline void f62_ext2_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    uint64_t z = f62_mul(a[0], b[0]);
    uint64_t o0 = f62_add(z, f62_mul(a[1], b[1]));
    uint64_t o1 = f62_sub(f62_mul(f62_add(a[0], a[1]), f62_add(b[0], b[1])), z);
    out[0] = o0; out[1] = o1;
}
static inline uint64_t f62_double(uint64_t a) { return f62_add(a, a); }
/* cubic extension x^3 + 2x + 2 — mod.rs:347-371 */
static inline void f62_ext3_mul(const uint64_t a[3], const uint64_t b[3], uint64_t out[3]) {
    uint64_t a0b0 = f62_mul(a[0], b[0]), a1b1 = f62_mul(a[1], b[1]), a2b2 = f62_mul(a[2], b[2]);
    uint64_t s01 = f62_mul(f62_add(a[0], a[1]), f62_add(b[0], b[1]));
    uint64_t m02 = f62_mul(f62_sub(a[0], a[2]), f62_sub(b[2], b[0]));
    uint64_t m12 = f62_mul(f62_sub(a[1], a[2]), f62_sub(b[1], b[2]));
    uint64_t a0b0_a1b1 = f62_add(a0b0, a1b1);
    uint64_t t = f62_double(f62_sub(f62_sub(m12, a1b1), a2b2));
    uint64_t o0 = f62_add(a0b0, t);
    uint64_t o1 = f62_sub(f62_sub(f62_add(s01, t), f62_double(a2b2)), a0b0_a1b1);
    uint64_t o2 = f62_sub(f62_add(m02, a0b0_a1b1), a2b2);
    out[0] = o0; out[1] = o1; out[2] = o2;
}
#endif
