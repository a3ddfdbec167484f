// Rescue-Prime Rp64_256 permutation and sponge for gfx950, one state per lane (12 x u64 in registers).
//
// Stands behind crypto::hash::Rp64_256 (crypto/src/hash/rescue/rp64_256/mod.rs:123-384).  State words are the
// reference's Montgomery residues, so inputs/outputs need no conversion.  Per permutation: 7 rounds of
// x^7 (4 modmuls/elt), MDS, +ARK1, x^(1/7) (72 modmuls/elt), MDS, +ARK2  = 6384 modmuls: VALU-bound.
// The circulant MDS (first row 7,23,8,26,13,10,9,7,6,22,21,8) is applied exactly, in integer arithmetic on
// the 32-bit halves via its 3x4-point real-FFT factorisation (every frequency-domain coefficient is a small
// power of two, cf. crypto/src/hash/mds/mds_f64_12x12.rs:41-98), then reduced once; results are canonical.
#pragma once
#include "gl64.cuh"
#include "rp64_256_constants.h"

namespace rp64 {

// round constants converted to Montgomery form at compile time
struct ArkTable {
    uint64_t v[7][12];
};
constexpr uint64_t cx_to_mont(uint64_t a) {
    return (uint64_t)((((unsigned __int128)a) << 64) % (unsigned __int128)0xffffffff00000001ull);
}
constexpr ArkTable make_ark(const uint64_t (&src)[7][12]) {
    ArkTable t{};
    for (int r = 0; r < 7; r++)
        for (int i = 0; i < 12; i++) t.v[r][i] = cx_to_mont(src[r][i]);
    return t;
}
static __constant__ ArkTable ARK1_T = make_ark(RP64_ARK1);
static __constant__ ArkTable ARK2_T = make_ark(RP64_ARK2);

__device__ __forceinline__ uint64_t exp7(uint64_t x) {
    uint64_t x2 = gl::sqr(x), x4 = gl::sqr(x2), x3 = gl::mul(x2, x);
    return gl::mul(x3, x4);
}

// acc[i] = acc[i]^(2^N) * tail[i] for all 12 lanes of the state; the squaring runs are real loops (small code,
// 12 independent multiplication chains in flight) — exp_acc of crypto/src/hash/rescue/mod.rs:20-28
template <int N, int W>
__device__ __forceinline__ void exp_acc(uint64_t (&acc)[W], const uint64_t (&tail)[W]) {
#pragma unroll 1
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i < W; i++) acc[i] = gl::sqr(acc[i]);
    }
#pragma unroll
    for (int i = 0; i < W; i++) acc[i] = gl::mul(acc[i], tail[i]);
}

// x^(1/7) = x^10540996611094048183 with the 72-multiplication addition chain of rp64_256/mod.rs:351-384
// (the same chain in rp64_256_jive/mod.rs:393-426 for the width-8 state)
template <int W>
__device__ __forceinline__ void inv_sbox(uint64_t (&st)[W]) {
    uint64_t t1[W], t2[W], t3[W], acc[W];
#pragma unroll
    for (int i = 0; i < W; i++) {
        t1[i] = gl::sqr(st[i]);   // x^10b
        t2[i] = gl::sqr(t1[i]);   // x^100b
        t3[i] = t2[i];
    }
    exp_acc<3, W>(t3, t2);        // t3 = x^100100b
#pragma unroll
    for (int i = 0; i < W; i++) acc[i] = t3[i];
    exp_acc<6, W>(acc, t3);       // t4 = x^100100100100b
    {
        uint64_t t4[W];
#pragma unroll
        for (int i = 0; i < W; i++) t4[i] = acc[i];
        exp_acc<12, W>(acc, t4);  // t5
    }
    exp_acc<6, W>(acc, t3);       // t6
    {
        uint64_t t6[W];
#pragma unroll
        for (int i = 0; i < W; i++) t6[i] = acc[i];
        exp_acc<31, W>(acc, t6);  // t7
#pragma unroll
        for (int i = 0; i < W; i++) {
            uint64_t a = gl::sqr(gl::sqr(gl::mul(gl::sqr(acc[i]), t6[i])));
            uint64_t b = gl::mul(gl::mul(t1[i], t2[i]), st[i]);
            st[i] = gl::mul(a, b);
        }
    }
}

// x^(1/7) on all W words, RP_INV_CHUNK at a time in a real loop: the chain keeps five W-word temporaries alive, which for
// W = 12 costs 244 VGPRs (2 waves per SIMD); chunks of 4 need ~100 and run 6 % faster (tools/time_hashers.py).
#ifndef RP_INV_CHUNK
#define RP_INV_CHUNK 4
#endif
template <int W>
__device__ __forceinline__ void inv_sbox_chunked(uint64_t (&st)[W]) {
    static_assert(W % RP_INV_CHUNK == 0, "chunk must divide the state width");
    uint64_t h[RP_INV_CHUNK];
#pragma unroll 1
    for (int part = 0; part < W / RP_INV_CHUNK; part++) {
#pragma unroll
        for (int i = 0; i < W; i++)
            if (i / RP_INV_CHUNK == part) h[i % RP_INV_CHUNK] = st[i];
        inv_sbox<RP_INV_CHUNK>(h);
#pragma unroll
        for (int i = 0; i < W; i++)
            if (i / RP_INV_CHUNK == part) st[i] = h[i % RP_INV_CHUNK];
    }
}

// y = circulant(MDS) * x on small non-negative integers (each x < 2^32): exact integer result (< 2^40).
__device__ __forceinline__ void mds_int(const int64_t (&s)[12], int64_t (&o)[12]) {
    // three 4-point real FFTs over the stride-3 sub-sequences
    int64_t u0, u1r, u1i, u2, u4, u5r, u5i, u6, u8, u9r, u9i, u10;
    {
        int64_t z0 = s[0] + s[6], z2 = s[0] - s[6], z1 = s[3] + s[9], z3 = s[3] - s[9];
        u0 = z0 + z1; u1r = z2; u1i = -z3; u2 = z0 - z1;
    }
    {
        int64_t z0 = s[1] + s[7], z2 = s[1] - s[7], z1 = s[4] + s[10], z3 = s[4] - s[10];
        u4 = z0 + z1; u5r = z2; u5i = -z3; u6 = z0 - z1;
    }
    {
        int64_t z0 = s[2] + s[8], z2 = s[2] - s[8], z1 = s[5] + s[11], z3 = s[5] - s[11];
        u8 = z0 + z1; u9r = z2; u9i = -z3; u10 = z0 - z1;
    }
    // frequency-domain products with the MDS spectrum: block 1 = [16, 8, 16] (cyclic convolution)
    int64_t v0 = 16 * u0 + 16 * u4 + 8 * u8;
    int64_t v4 = 8 * u0 + 16 * u4 + 16 * u8;
    int64_t v8 = 16 * u0 + 8 * u4 + 16 * u8;
    // block 3 = [-8, 1, 1] (negacyclic convolution)
    int64_t v2 = -8 * u2 - u6 - u10;
    int64_t v6 = u2 - 8 * u6 - u10;
    int64_t v10 = u2 + u6 - 8 * u10;
    // block 2 = [(-1+2i), (-1+i), (4+8i)]: complex "i-twisted" convolution
    //   z0 = x0*y0 - i*x1*y2 - i*x2*y1 ; z1 = x0*y1 + x1*y0 - i*x2*y2 ; z2 = x0*y2 + x1*y1 + x2*y0
    // complex products written out: (a+bi)(c+di) = (ac - bd) + (ad + bc) i ; -i*(p+qi) = q - p i
    int64_t p00r = -u1r - 2 * u1i, p00i = 2 * u1r - u1i;        // x0*y0, y0 = -1+2i
    int64_t p01r = -u1r - u1i, p01i = u1r - u1i;                // x0*y1, y1 = -1+i
    int64_t p02r = 4 * u1r - 8 * u1i, p02i = 8 * u1r + 4 * u1i; // x0*y2, y2 = 4+8i
    int64_t p10r = -u5r - 2 * u5i, p10i = 2 * u5r - u5i;        // x1*y0
    int64_t p11r = -u5r - u5i, p11i = u5r - u5i;                // x1*y1
    int64_t p12r = 4 * u5r - 8 * u5i, p12i = 8 * u5r + 4 * u5i; // x1*y2
    int64_t p20r = -u9r - 2 * u9i, p20i = 2 * u9r - u9i;        // x2*y0
    int64_t p21r = -u9r - u9i, p21i = u9r - u9i;                // x2*y1
    int64_t p22r = 4 * u9r - 8 * u9i, p22i = 8 * u9r + 4 * u9i; // x2*y2
    int64_t z0r = p00r + p12i + p21i, z0i = p00i - p12r - p21r;
    int64_t z1r = p01r + p10r + p22i, z1i = p01i + p10i - p22r;
    int64_t z2r = p02r + p11r + p20r, z2i = p02i + p11i + p20i;
    // inverse 4-point real FFTs (unnormalised; the spectrum above is pre-scaled accordingly)
    {
        int64_t a0 = v0 + v2, a1 = v0 - v2, a2 = z0r, a3 = -z0i;
        o[0] = a0 + a2; o[6] = a0 - a2; o[3] = a1 + a3; o[9] = a1 - a3;
    }
    {
        int64_t a0 = v4 + v6, a1 = v4 - v6, a2 = z1r, a3 = -z1i;
        o[1] = a0 + a2; o[7] = a0 - a2; o[4] = a1 + a3; o[10] = a1 - a3;
    }
    {
        int64_t a0 = v8 + v10, a1 = v8 - v10, a2 = z2r, a3 = -z2i;
        o[2] = a0 + a2; o[8] = a0 - a2; o[5] = a1 + a3; o[11] = a1 - a3;
    }
}

__device__ __forceinline__ void mds(uint64_t (&st)[12]) {
    int64_t lo[12], hi[12], ol[12], oh[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (int64_t)(st[i] & 0xffffffffull);
        hi[i] = (int64_t)(st[i] >> 32);
    }
    mds_int(lo, ol);
    mds_int(hi, oh);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        // value = ol + oh * 2^32  (< 2^73): fold as lo64 + mid32 * 2^64, reduce to canonical
        const uint64_t l = (uint64_t)ol[i], h = (uint64_t)oh[i];
        const uint64_t low = l + (h << 32);
        const uint32_t carry = low < l;
        const uint32_t mid = (uint32_t)(h >> 32) + carry;
        st[i] = gl::reduce160(low, mid, 0);
    }
}

__device__ __forceinline__ void permute(uint64_t (&st)[12]) {
#pragma unroll 1
    for (int r = 0; r < 7; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = exp7(st[i]);
        mds(st);
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], ARK1_T.v[r][i]);
        inv_sbox_chunked<12>(st);
        mds(st);
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], ARK2_T.v[r][i]);
    }
}

// Montgomery form of a small integer v: v * 2^64 mod p = v * (2^32 - 1) for v < 2^32
__device__ __forceinline__ uint64_t mont_small(uint32_t v) { return (uint64_t)v * 0xffffffffull; }

// hash_elements (rp64_256/mod.rs:224-257) over n elements fetched by `e(i)`; digest = state[4..8]
template <class E>
__device__ __forceinline__ void hash_elements(const E &e, uint32_t n, uint64_t (&digest)[4]) {
    uint64_t st[12];
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = 0;
    st[0] = mont_small(n);
    for (uint32_t base = 0; base < n; base += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (base + i < n) st[4 + i] = gl::add(st[4 + i], e(base + i));
        permute(st);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) digest[i] = st[4 + i];
}

// merge (rp64_256/mod.rs:181-192)
__device__ __forceinline__ void merge(const uint64_t (&two)[8], uint64_t (&digest)[4]) {
    uint64_t st[12];
    st[0] = mont_small(8);
    st[1] = st[2] = st[3] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) st[4 + i] = two[i];
    permute(st);
#pragma unroll
    for (int i = 0; i < 4; i++) digest[i] = st[4 + i];
}

}  // namespace rp64
