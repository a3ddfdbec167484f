// Device + host arithmetic for the reference's f128 field  p = 2^128 - 45*2^40 + 1  (math/src/field/f128/mod.rs).
// Values are canonical u128 integers (IS_CANONICAL = true, mod.rs:80), so memory images are compared as is.
// The reference reduces with a 128x64 schoolbook scheme (mod.rs:429-466); here the 256-bit product is folded with
// 2^128 = c (mod p), c = 45*2^40 - 1 < 2^46 — any exact modmul gives the same canonical result.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace f128 {

typedef unsigned __int128 u128;

__host__ __device__ constexpr u128 modulus() { return ((u128)0xFFFFFFFFFFFFFFFFull << 64 | 0xFFFFFFFFFFFFFFFFull) - ((u128)45 << 40) + 2; }
constexpr uint64_t C = (45ull << 40) - 1;   // 2^128 mod p

__host__ __device__ __forceinline__ u128 add(u128 a, u128 b) {
    const u128 M = modulus();
    u128 s = a + b;
    // a, b < M < 2^128: detect wrap or s >= M
    if (s < a || s >= M) s -= M;
    return s;
}
__host__ __device__ __forceinline__ u128 sub(u128 a, u128 b) {
    const u128 M = modulus();
    u128 d = a - b;
    if (a < b) d += M;
    return d;
}

// 128 x 128 -> 256 (hi, lo)
__host__ __device__ __forceinline__ void mul_wide(u128 a, u128 b, u128 &hi, u128 &lo) {
    const uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    const u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    const u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;     // < 3 * 2^64
    lo = (u128)(uint64_t)p00 | (mid << 64);
    hi = p11 + (p01 >> 64) + (p10 >> 64) + (mid >> 64);
}

__host__ __device__ __forceinline__ u128 mul(u128 a, u128 b) {
    const u128 M = modulus();
    u128 hi, lo;
    mul_wide(a, b, hi, lo);
    // hi * c  (128 x 46 bits -> up to 174 bits): h2 * 2^128 + l2
    const uint64_t h0 = (uint64_t)hi, h1 = (uint64_t)(hi >> 64);
    const u128 q0 = (u128)h0 * C, q1 = (u128)h1 * C;                  // q1 < 2^110
    const u128 l2 = q0 + (q1 << 64);
    const uint64_t carry1 = l2 < q0;
    const uint64_t h2 = (uint64_t)(q1 >> 64) + carry1;                // < 2^47
    // r = lo + l2 + h2 * c   (h2 * c < 2^93), tracking wraps past 2^128 (each wrap is worth +c)
    u128 r = lo + l2;
    uint64_t wraps = r < lo;
    const u128 t = (u128)h2 * C;
    const u128 r2 = r + t;
    wraps += r2 < r;
    r = r2;
    // wraps in {0,1,2}: add wraps * c, which can wrap at most once more (then the remainder is tiny)
    const u128 r3 = r + (u128)wraps * C;
    if (r3 < r) r = r3 + C; else r = r3;
    if (r >= M) r -= M;
    return r;
}

}  // namespace f128
