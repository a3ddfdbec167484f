// math::utils helpers that the reference's hot-path functions are built from, exposed on their own
// (math/src/utils/mod.rs): get_power_series_with_offset (:69-79, the batched-exp series behind get_twiddles,
// get_evaluation_offsets and the FRI inverse offsets) and batch_inversion (:169-215, Montgomery's trick; zeros stay zero).
// Base fields only; inputs and outputs are device vectors in the reference's internal representation.
#include <string.h>

#include <vector>

#include "batch_inv.cuh"
#include "fields.cuh"
#include "wf_internal.h"

namespace {

constexpr int RUN = 16;      // consecutive elements per lane

template <class T>
struct Pow2Table {           // base^(2^k) in internal form; index k covers exponents up to 2^36
    T v[36];
};

// out[i] = s * b^i: lane t starts at i0 = t * RUN with s * b^i0 (product of the set bits' table entries) and walks RUN steps
template <class F>
__global__ __launch_bounds__(256) void power_series_kernel(Pow2Table<typename F::T> tbl, typename F::T s, uint64_t n, typename F::T *out) {
    typedef typename F::T T;
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * RUN;
    if (i0 >= n) return;
    T cur = s;
    for (int k = 0; k < 36; k++)
        if ((i0 >> k) & 1) cur = F::mul(cur, tbl.v[k]);
    const T b = tbl.v[0];
#pragma unroll
    for (int k = 0; k < RUN; k++) {
        if (i0 + k < n) out[i0 + k] = cur;
        cur = F::mul(cur, b);
    }
}

// serial_batch_inversion (utils/mod.rs:194-215) over runs of RUN elements per lane: prefix products skipping zeros, one field
// inversion per WORKGROUP (batch_inv.cuh), walk back; zero inputs give zero outputs
template <class F>
__global__ __launch_bounds__(256) void batch_inversion_kernel(const typename F::T *in, uint64_t n, typename F::T one, uint64_t e_lo,
                                                              uint64_t e_hi, typename F::T *out) {
    typedef typename F::T T;
    __shared__ T sA[256], sB[256];
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * RUN;
    const uint32_t cnt = i0 >= n ? 0u : (n - i0 < RUN ? (uint32_t)(n - i0) : (uint32_t)RUN);
    T v[RUN], pre[RUN];
    T last = one;
#pragma unroll
    for (int k = 0; k < RUN; k++) {
        if ((uint32_t)k < cnt) {
            v[k] = F::load_norm(in[i0 + k]);
            pre[k] = last;
            if (!F::is_zero(v[k])) last = F::mul(last, v[k]);
        }
    }
    last = block_inverse_of_products<F>(last, one, e_lo, e_hi, sA, sB);
#pragma unroll
    for (int k = RUN - 1; k >= 0; k--) {
        if ((uint32_t)k < cnt) {
            if (F::is_zero(v[k])) {
                out[i0 + k] = F::zero();
            } else {
                out[i0 + k] = F::mul(pre[k], last);
                last = F::mul(last, v[k]);
            }
        }
    }
}

template <class HF>
int power_series(wf_ctx *ctx, const void *h_b, const void *h_s, uint64_t n, void *d_out) {
    typedef typename HF::T T;
    typedef typename HF::Dev F;
    T b, s;
    memcpy(&b, h_b, sizeof(T));
    memcpy(&s, h_s, sizeof(T));
    if (!HF::valid_internal(b) || !HF::valid_internal(s)) return WF_ERR_INVALID_ARG;
    Pow2Table<T> tbl;
    T cur = HF::from_internal(b);
    for (int k = 0; k < 36; k++) {
        tbl.v[k] = HF::to_internal(cur);
        cur = HF::mulmod(cur, cur);
    }
    const uint64_t lanes = (n + RUN - 1) / RUN, blocks = (lanes + 255) / 256;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    wf_prof_begin(ctx, "power_series");
    hipLaunchKernelGGL(power_series_kernel<F>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, tbl, HF::to_internal(HF::from_internal(s)), n,
                       (T *)d_out);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

template <class HF>
int batch_inversion(wf_ctx *ctx, const void *d_values, uint64_t n, void *d_out) {
    typedef typename HF::T T;
    typedef typename HF::Dev F;
    const uint64_t lanes = (n + RUN - 1) / RUN, blocks = (lanes + 255) / 256;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    unsigned __int128 e;
    if (sizeof(T) > 8) e = (unsigned __int128)f128::modulus() - 2;
    else if (HF::Dev::ID == WF_FIELD_F64) e = (unsigned __int128)gl::P - 2;
    else e = (unsigned __int128)f62::M - 2;
    wf_prof_begin(ctx, "batch_inversion");
    hipLaunchKernelGGL(batch_inversion_kernel<F>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, (const T *)d_values, n,
                       HF::to_internal(HF::from_u64(1)), (uint64_t)e, (uint64_t)(e >> 64), (T *)d_out);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

}  // namespace

#define WF_DISPATCH_FIELD(field, FN, ...)                          \
    switch (field) {                                                \
        case WF_FIELD_F64: return FN<HostF64>(__VA_ARGS__);         \
        case WF_FIELD_F128: return FN<HostF128>(__VA_ARGS__);       \
        case WF_FIELD_F62: return FN<HostF62>(__VA_ARGS__);         \
        default: return WF_ERR_UNSUPPORTED;                         \
    }

extern "C" int wf_get_power_series_with_offset(wf_ctx *ctx, int field, const void *h_b, const void *h_s, uint64_t n, void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !h_b || !h_s || (n && !d_out)) return WF_ERR_INVALID_ARG;
    if (n == 0) return WF_OK;
    WF_DISPATCH_FIELD(field, power_series, ctx, h_b, h_s, n, d_out);
}

extern "C" int wf_batch_inversion(wf_ctx *ctx, int field, const void *d_values, uint64_t n, void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || (n && (!d_values || !d_out))) return WF_ERR_INVALID_ARG;
    if (n == 0) return WF_OK;
    WF_DISPATCH_FIELD(field, batch_inversion, ctx, d_values, n, d_out);
}
