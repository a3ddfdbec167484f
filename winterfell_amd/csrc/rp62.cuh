// Rescue-Prime Rp62_248 (crypto/src/hash/rescue/rp62_248/mod.rs) for gfx950: width-12 state over the 62-bit field per lane,
// alpha = 3, 7 rounds, rate = state[0..8], element count in state[11], digest = state[0..4].  Words are the field's
// Montgomery residues, kept normalised.
#pragma once
#include "fields.cuh"
#include "rp62_248_constants.h"

namespace rp62 {

constexpr uint64_t cx_to_mont(uint64_t a) { return (uint64_t)((((unsigned __int128)a) << 64) % (unsigned __int128)f62::M); }

struct Tables {
    uint64_t mds[12][12], ark1[7][12], ark2[7][12];
};
constexpr Tables make_tables(const uint64_t (&mds)[12][12], const uint64_t (&a1)[7][12], const uint64_t (&a2)[7][12]) {
    Tables t{};
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) t.mds[i][j] = cx_to_mont(mds[i][j]);
    for (int r = 0; r < 7; r++)
        for (int i = 0; i < 12; i++) {
            t.ark1[r][i] = cx_to_mont(a1[r][i]);
            t.ark2[r][i] = cx_to_mont(a2[r][i]);
        }
    return t;
}
static __constant__ Tables TBL = make_tables(RP62_MDS, RP62_ARK1, RP62_ARK2);


// acc = acc^(2^N) * tail for W state words (exp_acc, crypto/src/hash/rescue/mod.rs:20-28)
template <int N, int W>
__device__ __forceinline__ void exp_acc(uint64_t (&acc)[W], const uint64_t (&tail)[W]) {
#pragma unroll 1
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i < W; i++) acc[i] = f62::mul(acc[i], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < W; i++) acc[i] = f62::mul(acc[i], tail[i]);
}

// x^(1/3) = x^3074416663688030891, the 69-multiplication chain of mod.rs:291-318, on W words
template <int W>
__device__ __forceinline__ void inv_sbox_w(uint64_t (&st)[W]) {
    uint64_t t1[W], t2[W], t4[W], t8[W], acc[W];
#pragma unroll
    for (int i = 0; i < W; i++) {
        t1[i] = f62::mul(st[i], st[i]);    // x^10b
        t2[i] = t1[i];
    }
    exp_acc<2, W>(t2, t1);                 // x^1010b
#pragma unroll
    for (int i = 0; i < W; i++) t4[i] = t2[i];
    exp_acc<4, W>(t4, t2);                 // x^10101010b
#pragma unroll
    for (int i = 0; i < W; i++) t8[i] = t4[i];
    exp_acc<8, W>(t8, t4);                 // x^1010101010101010b
#pragma unroll
    for (int i = 0; i < W; i++) acc[i] = t8[i];
    exp_acc<7, W>(acc, t2);
    exp_acc<15, W>(acc, t8);
    exp_acc<16, W>(acc, t8);
    exp_acc<8, W>(acc, t4);
#pragma unroll
    for (int i = 0; i < W; i++) st[i] = f62::mul(st[i], acc[i]);
}

// all 12 words, 4 at a time in a real loop (register pressure: see rp64::inv_sbox_chunked)
__device__ __forceinline__ void inv_sbox(uint64_t (&st)[12]) {
    uint64_t h[4];
#pragma unroll 1
    for (int part = 0; part < 3; part++) {
#pragma unroll
        for (int i = 0; i < 12; i++)
            if (i / 4 == part) h[i % 4] = st[i];
        inv_sbox_w<4>(h);
#pragma unroll
        for (int i = 0; i < 12; i++)
            if (i / 4 == part) st[i] = h[i % 4];
    }
}

__device__ __forceinline__ void mds(uint64_t (&st)[12]) {
    uint64_t r[12];
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        uint64_t acc = f62::mul(TBL.mds[i][0], st[0]);
#pragma unroll
        for (int j = 1; j < 12; j++) acc = f62::add(acc, f62::mul(TBL.mds[i][j], st[j]));
        r[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = r[i];
}

__device__ __forceinline__ void permute(uint64_t (&st)[12]) {
#pragma unroll 1
    for (int r = 0; r < 7; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = f62::mul(f62::mul(st[i], st[i]), st[i]);
        mds(st);
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = f62::add(st[i], TBL.ark1[r][i]);
        inv_sbox(st);
        mds(st);
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = f62::add(st[i], TBL.ark2[r][i]);
    }
}

// BaseElement::new(v) for v < M: v * R^2 * R^-1
__device__ __forceinline__ uint64_t to_mont(uint64_t v) {
    constexpr uint64_t R2 = cx_to_mont(cx_to_mont(1));
    return f62::mul(v, R2);
}

template <class E>
__device__ __forceinline__ void hash_elements(const E &e, uint32_t n, uint64_t (&digest)[4]) {
    uint64_t st[12];
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = 0;
    st[11] = to_mont(n);
    for (uint32_t base = 0; base < n; base += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (base + i < n) st[i] = f62::add(st[i], f62::norm(e(base + i)));
        permute(st);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) digest[i] = st[i];
}

__device__ __forceinline__ void merge(const uint64_t (&two)[8], uint64_t (&digest)[4]) {
    uint64_t st[12];
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = f62::norm(two[i]);
    st[8] = st[9] = st[10] = 0;
    st[11] = to_mont(8);
    permute(st);
#pragma unroll
    for (int i = 0; i < 4; i++) digest[i] = st[i];
}

}  // namespace rp62
