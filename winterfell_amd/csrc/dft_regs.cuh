// In-register radix-2^k DFT building blocks shared by the NTT passes and the FRI fold (generic over the field).
#pragma once
#include "fields.cuh"

namespace {

__host__ __device__ constexpr int brev(int i, int bits) {
    int r = 0;
    for (int k = 0; k < bits; k++) r |= ((i >> k) & 1) << (bits - 1 - k);
    return r;
}

// In-register decimation-in-frequency DFT of N = 2^LOGN points; X[k] ends up in x[brev(k)].
// w16: the field's omega_16^j table (unused by fields whose small roots are powers of two).
// TAB: w16 is in the field's NTT table layout (F::TAB_WORDS words per entry, F::mul_w16_tab) instead of one word per entry.
template <class F, int LOGN, bool TAB = false>
__device__ __forceinline__ void dft_dif(typename F::T (&x)[1 << LOGN], const typename F::T *w16) {
    typedef typename F::T T;
    constexpr int N = 1 << LOGN;
#pragma unroll
    for (int s = 0; s < LOGN; s++) {
        const int half = N >> (s + 1);
#pragma unroll
        for (int blk = 0; blk < N; blk += 2 * half) {
#pragma unroll
            for (int i = 0; i < half; i++) {
                const T u = x[blk + i], v = x[blk + i + half];
                x[blk + i] = F::add(u, v);
                // twiddle omega_{2*half}^i = omega_16^(i * 8 / half)
                x[blk + i + half] = TAB ? F::mul_w16_tab(F::sub(u, v), (i * 8) / half, w16) : F::mul_w16(F::sub(u, v), (i * 8) / half, w16);
            }
        }
    }
}

template <class F>
__device__ __forceinline__ typename F::T series_at(const typename F::T *lo, const typename F::T *hi, uint32_t log_lo, uint64_t i) {
    return F::mul(lo[i & ((1ull << log_lo) - 1)], hi[i >> log_lo]);
}
// the same for indices known to fit 32 bits (every series here has at most 2^32 entries): 32-bit index arithmetic lets
// the loads use the scalar-base + 32-bit-offset addressing form instead of 64-bit address arithmetic per element
template <class F>
__device__ __forceinline__ typename F::T series_at32(const typename F::T *lo, const typename F::T *hi, uint32_t log_lo, uint32_t i) {
    return F::mul(lo[i & ((1u << log_lo) - 1u)], hi[i >> log_lo]);
}

}  // namespace
