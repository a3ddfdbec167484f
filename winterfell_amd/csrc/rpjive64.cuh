// Rescue-Prime RpJive64_256 (crypto/src/hash/rescue/rp64_256_jive/mod.rs) for gfx950: width-8 state per lane, 7 rounds,
// Jive compression for 2-to-1 hashing.  Shares the S-box chains with rp64.cuh; the 8x8 circulant MDS (first row
// 23, 8, 13, 10, 7, 6, 21, 8) is applied exactly in integer arithmetic on the 32-bit halves (entries <= 23, so a row
// sum stays below 2^39) and reduced once — the reference's frequency-domain mds_multiply computes the same product.
#pragma once
#include "rp64.cuh"
#include "rpjive64_constants.h"

namespace rpj {

struct ArkTable8 {
    uint64_t v[7][8];
};
constexpr ArkTable8 make_ark8(const uint64_t (&src)[7][8]) {
    ArkTable8 t{};
    for (int r = 0; r < 7; r++)
        for (int i = 0; i < 8; i++) t.v[r][i] = rp64::cx_to_mont(src[r][i]);
    return t;
}
static __constant__ ArkTable8 ARK1_T = make_ark8(RPJ64_ARK1);
static __constant__ ArkTable8 ARK2_T = make_ark8(RPJ64_ARK2);

__device__ __forceinline__ void mds(uint64_t (&st)[8]) {
    constexpr uint32_t ROW[8] = {23, 8, 13, 10, 7, 6, 21, 8};   // M[i][j] = ROW[(j - i) mod 8], mod.rs:417-499
    uint64_t ol[8], oh[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t sl = 0, sh = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t m = ROW[(j - i + 8) & 7];
            sl += (uint64_t)m * (uint32_t)st[j];
            sh += (uint64_t)m * (uint32_t)(st[j] >> 32);
        }
        ol[i] = sl;
        oh[i] = sh;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        // value = ol + oh * 2^32 (< 2^72): lo64 + mid * 2^64, reduced to canonical
        const uint64_t low = ol[i] + (oh[i] << 32);
        const uint32_t carry = low < ol[i];
        const uint32_t mid = (uint32_t)(oh[i] >> 32) + carry;
        st[i] = gl::reduce160(low, mid, 0);
    }
}

__device__ __forceinline__ void permute(uint64_t (&st)[8]) {
#pragma unroll 1
    for (int r = 0; r < 7; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) st[i] = rp64::exp7(st[i]);
        mds(st);
#pragma unroll
        for (int i = 0; i < 8; i++) st[i] = gl::add(st[i], ARK1_T.v[r][i]);
        rp64::inv_sbox_chunked<8>(st);
        mds(st);
#pragma unroll
        for (int i = 0; i < 8; i++) st[i] = gl::add(st[i], ARK2_T.v[r][i]);
    }
}

// apply_jive_summation (mod.rs:355-369)
__device__ __forceinline__ void jive_sum(const uint64_t (&init)[8], const uint64_t (&fin)[8], uint64_t (&digest)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) digest[i] = gl::add(gl::add(init[i], init[4 + i]), gl::add(fin[i], fin[4 + i]));
}

// hash_elements (mod.rs:268-313): capacity[0] = 1 iff n % 4 != 0; rate = state[4..8]; a partial final block is padded by
// OVERWRITING the next rate element with 1 and the rest with 0
template <class E>
__device__ __forceinline__ void hash_elements(const E &e, uint32_t n, uint64_t (&digest)[4]) {
    uint64_t st[8];
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = 0;
    const uint64_t one = rp64::mont_small(1);
    if (n & 3) st[0] = one;
    uint32_t base = 0;
    for (; base + 4 <= n; base += 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) st[4 + i] = gl::add(st[4 + i], e(base + i));
        permute(st);
    }
    const uint32_t left = n - base;
    if (left) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if ((uint32_t)i < left) st[4 + i] = gl::add(st[4 + i], e(base + i));
            else st[4 + i] = (uint32_t)i == left ? one : 0;
        }
        permute(st);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) digest[i] = st[4 + i];
}

// merge (mod.rs:186-200)
__device__ __forceinline__ void merge(const uint64_t (&two)[8], uint64_t (&digest)[4]) {
    uint64_t st[8];
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = two[i];
    permute(st);
    jive_sum(two, st, digest);
}

}  // namespace rpj
