// Device arithmetic for the reference's f64 field  p = 2^64 - 2^32 + 1  (math/src/field/f64/mod.rs).
//
// Values are the reference's internal representation: canonical Montgomery residues x*R mod p, R = 2^64,
// always in [0, p) (f64/mod.rs:60).  Because field arithmetic is exact, any correct implementation that
// keeps results canonical reproduces the reference's memory image bit for bit.
//
// gfx950 notes: there is no 64-bit integer multiplier; a 64x64->128 product is four v_mad_u64_u32.
// Montgomery reduction for this prime is shift/add only (eprint 2022/274, the same form the reference
// uses in mont_red_cst, f64/mod.rs:714-724) so a modmul is 4 mads + ~14 full-rate VALU ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl {

typedef unsigned __int128 u128;

constexpr uint64_t P = 0xffffffff00000001ull;
constexpr uint64_t EPS = 0xffffffffull;  // 2^64 mod p = 2^32 - 1

// a + b mod p for canonical a, b  (f64/mod.rs:319-324: a - (p - b))
__device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) {
    uint64_t t = P - b;
    uint64_t x = a - t;
    uint32_t adj = 0u - (uint32_t)(a < t);
    return x - (uint64_t)adj;
}

// a - b mod p for canonical a, b  (f64/mod.rs:339-343)
__device__ __forceinline__ uint64_t sub(uint64_t a, uint64_t b) {
    uint64_t x = a - b;
    uint32_t adj = 0u - (uint32_t)(a < b);
    return x - (uint64_t)adj;
}

__device__ __forceinline__ uint64_t neg(uint64_t a) { return a ? P - a : 0; }

// Montgomery reduction of a 128-bit value (xh:xl) < p * 2^64  ->  x / 2^64 mod p, canonical.
__device__ __forceinline__ uint64_t mont_red(uint64_t xl, uint64_t xh) {
    uint64_t a = xl + (xl << 32);
    uint64_t e = a < xl;
    uint64_t b = a - (a >> 32) - e;
    uint64_t r = xh - b;
    uint32_t adj = 0u - (uint32_t)(xh < b);
    return r - (uint64_t)adj;
}

// Montgomery product: (aR)(bR)/R = abR  (f64/mod.rs:357-359)
__device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b) {
    u128 x = (u128)a * (u128)b;
    return mont_red((uint64_t)x, (uint64_t)(x >> 64));
}

// Montgomery square: the cross product a0*a1 is computed once (3 multiplies instead of 4)
__device__ __forceinline__ uint64_t sqr(uint64_t a) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32);
    const uint64_t p00 = (uint64_t)a0 * a0, p01 = (uint64_t)a0 * a1, p11 = (uint64_t)a1 * a1;
    // a^2 = p00 + 2*p01*2^32 + p11*2^64
    const uint64_t c2 = p01 >> 31;              // bits of 2*p01 above 2^32 position -> contributes to the high word
    const uint64_t mid = p01 << 33;             // (2*p01 << 32) low 64 bits
    const uint64_t lo = p00 + mid;
    const uint64_t hi = p11 + c2 + (lo < p00);
    return mont_red(lo, hi);
}

// canonical integer of an internal value (mont_to_int, f64/mod.rs:731-737)
__device__ __forceinline__ uint64_t to_int(uint64_t a) { return mont_red(a, 0); }

// Plain (non-Montgomery) reduction of lo + mid*2^64 + hi*2^96 with mid < 2^32, hi < 2^63:
// 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).  Result canonical.
__device__ __forceinline__ uint64_t reduce160(uint64_t lo, uint32_t mid, uint64_t hi) {
    uint64_t t = lo - hi;
    if (lo < hi) t -= EPS;  // + p (mod 2^64)
    uint64_t m = ((uint64_t)mid << 32) - (uint64_t)mid;  // mid * (2^32 - 1)
    uint64_t r = t + m;
    if (r < t) r += EPS;  // wrapped past 2^64: 2^64 = EPS (cannot wrap twice)
    if (r >= P) r -= P;
    return r;
}

// x * 2^S mod p for a compile-time 0 < S < 96.  Multiplying a Montgomery residue by the plain integer
// 2^S gives the Montgomery residue of x*2^S, so this is how the radix-16 butterflies apply the
// twiddles omega_16^j = 2^(12 j)  (omega_64 = 8, f64/mod.rs:17,258-267).
template <int S>
__device__ __forceinline__ uint64_t mul_pow2(uint64_t x) {
    static_assert(S > 0 && S < 96, "shift out of range");
    uint64_t lo, hi;
    uint32_t mid;
    if constexpr (S < 32) {
        lo = x << S;
        mid = (uint32_t)(x >> (64 - S));
        hi = 0;
    } else if constexpr (S == 32) {
        lo = x << 32;
        mid = (uint32_t)(x >> 32);
        hi = 0;
    } else if constexpr (S < 64) {
        lo = x << S;
        mid = (uint32_t)(x >> (64 - S));
        hi = x >> (96 - S);
    } else if constexpr (S == 64) {
        lo = 0;
        mid = (uint32_t)x;
        hi = x >> 32;
    } else {
        lo = 0;
        mid = (uint32_t)(x << (S - 64));
        hi = x >> (96 - S);
    }
    return reduce160(lo, mid, hi);
}

// ---- extension fields (math/src/field/f64/mod.rs:401-499), elements are D consecutive base words ----------
// quadratic extension x^2 - x + 2: 3 base multiplications
__device__ __forceinline__ void ext2_mul(const uint64_t (&a)[2], const uint64_t (&b)[2], uint64_t (&o)[2]) {
    const uint64_t a0b0 = mul(a[0], b[0]);
    const uint64_t a1b1 = mul(a[1], b[1]);
    const uint64_t t = mul(add(a[0], a[1]), add(b[0], b[1]));
    o[0] = sub(a0b0, add(a1b1, a1b1));
    o[1] = sub(t, a0b0);
}
// cubic extension x^3 - x - 1: 6 base multiplications
__device__ __forceinline__ void ext3_mul(const uint64_t (&a)[3], const uint64_t (&b)[3], uint64_t (&o)[3]) {
    const uint64_t a0b0 = mul(a[0], b[0]), a1b1 = mul(a[1], b[1]), a2b2 = mul(a[2], b[2]);
    const uint64_t s01 = mul(add(a[0], a[1]), add(b[0], b[1]));
    const uint64_t s02 = mul(add(a[0], a[2]), add(b[0], b[2]));
    const uint64_t s12 = mul(add(a[1], a[2]), add(b[1], b[2]));
    const uint64_t m = sub(a0b0, a1b1);
    o[0] = sub(add(s12, m), a2b2);
    o[1] = sub(sub(add(s01, s12), add(a1b1, a1b1)), a0b0);
    o[2] = sub(s02, m);
}
template <int D>
__device__ __forceinline__ void ext_mul(const uint64_t (&a)[D], const uint64_t (&b)[D], uint64_t (&o)[D]) {
    if constexpr (D == 1) o[0] = mul(a[0], b[0]);
    else if constexpr (D == 2) ext2_mul(a, b, o);
    else ext3_mul(a, b, o);
}

}  // namespace gl
