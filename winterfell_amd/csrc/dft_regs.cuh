// In-register radix-2^k DFT building blocks shared by the NTT passes and the FRI fold (f64 field).
#pragma once
#include "gl64.cuh"

namespace {

__host__ __device__ constexpr int brev(int i, int bits) {
    int r = 0;
    for (int k = 0; k < bits; k++) r |= ((i >> k) & 1) << (bits - 1 - k);
    return r;
}

// multiply by omega_16^j = 2^(12 j), j in [0, 8); j is a compile-time constant after unrolling
__device__ __forceinline__ uint64_t mul_w16(uint64_t v, int j) {
    switch (j) {
        case 0: return v;
        case 1: return gl::mul_pow2<12>(v);
        case 2: return gl::mul_pow2<24>(v);
        case 3: return gl::mul_pow2<36>(v);
        case 4: return gl::mul_pow2<48>(v);
        case 5: return gl::mul_pow2<60>(v);
        case 6: return gl::mul_pow2<72>(v);
        default: return gl::mul_pow2<84>(v);
    }
}

// In-register decimation-in-frequency DFT of N = 2^LOGN points; X[k] ends up in x[brev(k)].
template <int LOGN>
__device__ __forceinline__ void dft_dif(uint64_t (&x)[1 << LOGN]) {
    constexpr int N = 1 << LOGN;
#pragma unroll
    for (int s = 0; s < LOGN; s++) {
        const int half = N >> (s + 1);
#pragma unroll
        for (int blk = 0; blk < N; blk += 2 * half) {
#pragma unroll
            for (int i = 0; i < half; i++) {
                uint64_t u = x[blk + i], v = x[blk + i + half];
                x[blk + i] = gl::add(u, v);
                // twiddle omega_{2*half}^i = omega_16^(i * 8 / half)
                x[blk + i + half] = mul_w16(gl::sub(u, v), (i * 8) / half);
            }
        }
    }
}

__device__ __forceinline__ uint64_t series_at(const uint64_t *lo, const uint64_t *hi, uint32_t log_lo, uint64_t i) {
    uint64_t l = lo[i & ((1ull << log_lo) - 1)];
    uint64_t h = hi[i >> log_lo];
    return gl::mul(l, h);
}


}  // namespace
