// Workgroup-level half of Montgomery's batch inversion (math/src/utils/mod.rs:194-215 batches 1024 values per inversion).
// Every lane brings the product P of its own run of values (prefix products kept in registers) and gets 1 / P back:
//   inclusive prefix and suffix products of the 256 lane products (two LDS scans, 8 steps each),
//   ONE field inversion of the total by wavefront 0 (a^(p-2): ~190 multiplications that the other three wavefronts do not
//   execute — per-lane inversions cost every wavefront those 190 instructions),
//   1 / P_t = (1 / total) * prefix_{t-1} * suffix_{t+1}.
// Per element of a 16-value run: 3 + 18/16 + 190/(4*16) = 7.1 multiplications instead of 3 + 190/16 = 14.9.
#pragma once
#include "fields.cuh"

// a^(modulus - 2), exponent passed as two words
template <class F>
__device__ __forceinline__ typename F::T batch_inv_fermat(typename F::T a, typename F::T one, uint64_t e_lo, uint64_t e_hi) {
    typename F::T r = one;
    bool started = false;
    for (int bit = 127; bit >= 0; bit--) {
        const bool set = ((bit >= 64 ? e_hi : e_lo) >> (bit & 63)) & 1;
        if (started) r = F::mul(r, r);
        if (set) {
            r = started ? F::mul(r, a) : a;
            started = true;
        }
    }
    return r;
}

// all 256 threads of the workgroup must call this (lanes without work pass `one`); sA / sB: 256 elements of LDS each
template <class F>
__device__ __forceinline__ typename F::T block_inverse_of_products(typename F::T P, typename F::T one, uint64_t e_lo, uint64_t e_hi,
                                                                    typename F::T *sA, typename F::T *sB) {
    typedef typename F::T T;
    const int t = threadIdx.x;
    sA[t] = P;
    sB[t] = P;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        T a = sA[t], b = sB[t];
        const T ua = t >= d ? sA[t - d] : one, ub = t + d < 256 ? sB[t + d] : one;
        __syncthreads();
        if (t >= d) a = F::mul(a, ua);
        if (t + d < 256) b = F::mul(b, ub);
        sA[t] = a;
        sB[t] = b;
        __syncthreads();
    }
    const T pre = t > 0 ? sA[t - 1] : one, suf = t < 255 ? sB[t + 1] : one;
    const T total = sA[255];
    __syncthreads();
    if (t < 64) {                                            // one wavefront inverts the workgroup's total
        const T it = batch_inv_fermat<F>(total, one, e_lo, e_hi);
        if (t == 0) sA[0] = it;
    }
    __syncthreads();
    return F::mul(F::mul(sA[0], pre), suf);
}
