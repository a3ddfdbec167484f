// Out-of-domain evaluation and DEEP composition on the GPU (all three fields, extension degree 1..3).
//
// Reference behaviour reproduced:
//   ColMatrix::evaluate_columns_at           prover/src/matrix/col_matrix.rs (polynom::eval per column, math/src/polynom/mod.rs:55-61)
//   TracePolyTable::get_ood_frame            prover/src/trace/poly_table.rs:68-76        (points z and z*g)
//   CompositionPoly::get_ood_frame           prover/src/constraints/composition_poly.rs:101-108
//   DeepCompositionPoly::add_trace_polys     prover/src/composer/mod.rs:67-169
//     acc_trace_poly = mul_acc + constant    composer/mod.rs:200-210, math/src/utils/mod.rs:138-145
//     merge_compositions = syn_div + add     composer/mod.rs:186-198, math/src/polynom/mod.rs:499-506
//
// Field arithmetic is exact, so any evaluation order gives the reference's words as long as results are canonical:
//  * Horner evaluation of an n-coefficient column at x is split 256 ways: lane t evaluates the stride-256 sub-polynomial
//    sum_m c[t + 256 m] y^m at y = x^256 (coalesced loads), the workgroup sums x^t * P_t; long columns are cut into
//    segments that are recombined by a second tiny kernel.
//  * syn_div_in_place(p, 1, b) returns q_i = sum_{k>i} p_k b^(k-i-1) (the remainder is dropped, so the "- T(z) * cc"
//    constant of acc_trace_poly never reaches the output): a suffix Horner scan.  Tiles of 1024 coefficients compute
//    their tile polynomial at b, the tile carries are the same problem one level up with base b^1024 (recursion), and a
//    final pass replays every tile with its carry (in-LDS suffix scan over 256 lanes x 4 coefficients).
//  * both compositions (divisors x - z and x - z*g) share the accumulated column sum S_j = sum_i cc_i * T_i[j].
#include <string.h>

#include "fields.cuh"
#include "wf_internal.h"

namespace {

constexpr int NPOW = 32;        // b^(2^k), k < 32
constexpr int NTBL = 256;       // b^t, t < 256
constexpr int TILE_LOG = 10;    // syn_div tile: 1024 coefficients = 256 lanes x 4

// per-base power tables, D words per entry:  [0, NPOW): b^(2^k)   [NPOW, NPOW + NTBL): b^t
template <class F, int D>
struct Pows {
    const typename F::T *p;
    __device__ __forceinline__ void pow2(int k, typename F::T (&o)[D]) const {
#pragma unroll
        for (int d = 0; d < D; d++) o[d] = p[k * D + d];
    }
    __device__ __forceinline__ void tbl(int t, typename F::T (&o)[D]) const {
#pragma unroll
        for (int d = 0; d < D; d++) o[d] = p[(NPOW + t) * D + d];
    }
};

template <class T, int D>
struct Elem {
    T v[D];
};

template <class T, int D>
struct Elems4 {
    Elem<T, D> e[4];
};

// base given either by value (src == nullptr; workgroup g of the launch takes bases.e[g] and writes table g, so the tables of
// up to four points cost ONE chain of sequential squarings instead of four) or as entry `src_k` of another table (the
// recursion's b^1024)
template <class F, int D>
__global__ __launch_bounds__(256) void pows_kernel(Elems4<typename F::T, D> bases, const typename F::T *src, int src_k,
                                                   typename F::T one, typename F::T *out, uint64_t out_stride) {
    typedef typename F::T T;
    __shared__ T sq[NPOW][D];
    const int t = threadIdx.x;
    const Elem<T, D> base = bases.e[blockIdx.x];
    out += blockIdx.x * out_stride;
    if (t == 0) {
        T cur[D], nxt[D];
#pragma unroll
        for (int d = 0; d < D; d++) cur[d] = src ? src[src_k * D + d] : F::load_norm(base.v[d]);
        for (int k = 0; k < NPOW; k++) {
#pragma unroll
            for (int d = 0; d < D; d++) sq[k][d] = cur[d];
            F::template ext_mul<D>(cur, cur, nxt);
#pragma unroll
            for (int d = 0; d < D; d++) cur[d] = nxt[d];
        }
    }
    __syncthreads();
    if (t < NPOW) {
#pragma unroll
        for (int d = 0; d < D; d++) out[t * D + d] = sq[t][d];
    }
    T acc[D], tmp[D], f[D];
#pragma unroll
    for (int d = 0; d < D; d++) acc[d] = d == 0 ? one : F::zero();
    for (int k = 0; k < 8; k++) {
        if ((t >> k) & 1) {
#pragma unroll
            for (int d = 0; d < D; d++) f[d] = sq[k][d];
            F::template ext_mul<D>(acc, f, tmp);
#pragma unroll
            for (int d = 0; d < D; d++) acc[d] = tmp[d];
        }
    }
#pragma unroll
    for (int d = 0; d < D; d++) out[(NPOW + t) * D + d] = acc[d];
}

// out = [z, z * g] (g a base-field element in internal form: coordinate-wise), z read from DEVICE memory — the out-of-domain point
// where a device coin drew it (wf_polys_evaluate_at_dev / wf_deep_compose_dev)
template <class F, int D>
__global__ void ood_points_kernel(const typename F::T *z, typename F::T g, typename F::T *out) {
    const int d = threadIdx.x;
    if (d < D) {
        const typename F::T v = F::load_norm(z[d]);
        out[d] = v;
        out[D + d] = F::mul(v, g);
    }
}

// workgroup sum of one degree-D element per lane; result valid in lane 0
template <class F, int D>
__device__ __forceinline__ void block_sum(typename F::T (&v)[D], typename F::T (*buf)[D]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int d = 0; d < D; d++) buf[t][d] = v[d];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int d = 0; d < D; d++) buf[t][d] = F::add(buf[t][d], buf[t + s][d]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int d = 0; d < D; d++) v[d] = buf[0][d];
    __syncthreads();
}

// acc = acc * y + c   (c of degree PD <= D, lifted)
template <class F, int PD, int D>
__device__ __forceinline__ void horner_step(typename F::T (&acc)[D], const typename F::T (&y)[D], const typename F::T *c) {
    typename F::T tmp[D];
    F::template ext_mul<D>(acc, y, tmp);
#pragma unroll
    for (int d = 0; d < D; d++) acc[d] = d < PD ? F::add(tmp[d], F::load_norm(c[d])) : tmp[d];
}

// partial[p][col][seg] = sum_{i < seg_len} c[seg * seg_len + i] x_p^i      grid (nseg, cols), NP points per launch
template <class F, int PD, int D, int NP>
__global__ __launch_bounds__(256) void eval_partial_kernel(const typename F::T *polys, uint64_t col_stride, uint32_t log_seg,
                                                           uint32_t log_thr, const typename F::T *pw, uint64_t pw_stride,
                                                           typename F::T *partials) {
    typedef typename F::T T;
    __shared__ T buf[256][D];
    const uint32_t t = threadIdx.x, seg = blockIdx.x, col = blockIdx.y, nseg = gridDim.x, cols = gridDim.y;
    const uint32_t thr = 1u << log_thr;                       // active lanes (256, or n when n < 256)
    const uint32_t steps = 1u << (log_seg - log_thr);
    const T *c = polys + (uint64_t)col * col_stride + ((uint64_t)seg << log_seg) * PD;
    T acc[NP][D], y[NP][D];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        Pows<F, D> P{pw + p * pw_stride};
        P.pow2(log_thr, y[p]);
#pragma unroll
        for (int d = 0; d < D; d++) acc[p][d] = F::zero();
    }
    if (t < thr) {
        for (uint32_t m = steps; m-- > 0;) {
            const T *e = c + ((uint64_t)m * thr + t) * PD;
            T cv[PD];
#pragma unroll
            for (int d = 0; d < PD; d++) cv[d] = e[d];
#pragma unroll
            for (int p = 0; p < NP; p++) horner_step<F, PD, D>(acc[p], y[p], cv);
        }
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        Pows<F, D> P{pw + p * pw_stride};
        T xt[D], v[D];
        P.tbl(t, xt);
        F::template ext_mul<D>(acc[p], xt, v);
        block_sum<F, D>(v, buf);
        if (t == 0) {
            T *o = partials + (((uint64_t)p * cols + col) * nseg + seg) * D;
#pragma unroll
            for (int d = 0; d < D; d++) o[d] = v[d];
        }
    }
}

// out[p][col] = sum_seg partial[p][col][seg] * (x_p^seg_len)^seg
template <class F, int D>
__global__ __launch_bounds__(64) void eval_combine_kernel(const typename F::T *partials, uint32_t cols, uint32_t nseg,
                                                          uint32_t log_seg, uint32_t np, const typename F::T *pw,
                                                          uint64_t pw_stride, typename F::T *out) {
    typedef typename F::T T;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= np * cols) return;
    const uint32_t p = gid / cols;
    Pows<F, D> P{pw + p * pw_stride};
    T X[D], acc[D];
    P.pow2(log_seg, X);
#pragma unroll
    for (int d = 0; d < D; d++) acc[d] = F::zero();
    const T *src = partials + (uint64_t)gid * nseg * D;
    for (uint32_t s = nseg; s-- > 0;) horner_step<F, D, D>(acc, X, src + (uint64_t)s * D);
#pragma unroll
    for (int d = 0; d < D; d++) out[(uint64_t)gid * D + d] = acc[d];
}

// the same recombination for MANY segments (up to 1024): one workgroup per (point, column); lane t runs Horner over its
// C = nseg / lanes consecutive partials with X = x^seg_len, lifts the result by X^(t * C) (product of the table's
// x^(2^k) entries over the set bits of t * C) and the workgroup sums.  Short segments keep the per-lane Horner chain of
// eval_partial_kernel at 4 steps; the 64-segment limit of the sequential kernel made it 64.
template <class F, int D>
__global__ __launch_bounds__(256) void eval_combine_wide_kernel(const typename F::T *partials, uint32_t cols, uint32_t log_nseg, uint32_t log_seg,
                                                                const typename F::T *pw, uint64_t pw_stride, typename F::T one, typename F::T *out) {
    typedef typename F::T T;
    __shared__ T buf[256][D];
    const uint32_t gid = blockIdx.x, p = gid / cols, t = threadIdx.x;
    Pows<F, D> P{pw + p * pw_stride};
    const uint32_t log_thr = log_nseg < 8 ? log_nseg : 8, log_c = log_nseg - log_thr, C = 1u << log_c;
    T acc[D];
#pragma unroll
    for (int d = 0; d < D; d++) acc[d] = F::zero();
    if (t < (1u << log_thr)) {
        T X[D];
        P.pow2(log_seg, X);
        const T *src = partials + ((uint64_t)gid << log_nseg) * D + (uint64_t)t * C * D;
        for (uint32_t s = C; s-- > 0;) horner_step<F, D, D>(acc, X, src + (uint64_t)s * D);
        const uint32_t e = t << log_c;                       // lift by X^e = x^(e * 2^log_seg)
        for (uint32_t k = 0; k < log_nseg; k++) {
            if ((e >> k) & 1) {
                T f[D], tmp[D];
                P.pow2(log_seg + k, f);
                F::template ext_mul<D>(acc, f, tmp);
#pragma unroll
                for (int d = 0; d < D; d++) acc[d] = tmp[d];
            }
        }
    }
    (void)one;
    block_sum<F, D>(acc, buf);
    if (t == 0) {
#pragma unroll
        for (int d = 0; d < D; d++) out[(uint64_t)gid * D + d] = acc[d];
    }
}

// S[j] = sum_i cc_i * T_i[j]   (main: base-field columns, k.mul_base(b); aux / quot: columns over E, full products)
template <class F, int D>
__global__ __launch_bounds__(256) void deep_acc_kernel(const typename F::T *main_polys, uint32_t c_main, uint64_t main_stride,
                                                       const typename F::T *aux, uint32_t c_aux, uint64_t aux_stride,
                                                       const typename F::T *quot, uint32_t c_q, uint64_t q_stride, uint64_t n,
                                                       const typename F::T *cc, typename F::T *S) {
    typedef typename F::T T;
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    T acc[D];
#pragma unroll
    for (int d = 0; d < D; d++) acc[d] = F::zero();
    uint32_t i = 0;
    for (uint32_t k = 0; k < c_main; k++, i++) {
        const T v = F::load_norm(main_polys[(uint64_t)k * main_stride + j]);
#pragma unroll
        for (int d = 0; d < D; d++) acc[d] = F::add(acc[d], F::mul(F::load_norm(cc[i * D + d]), v));
    }
    for (uint32_t k = 0; k < c_aux + c_q; k++, i++) {
        const T *src = k < c_aux ? aux + (uint64_t)k * aux_stride + j * D : quot + (uint64_t)(k - c_aux) * q_stride + j * D;
        T e[D], c[D], t[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            e[d] = F::load_norm(src[d]);
            c[d] = F::load_norm(cc[i * D + d]);
        }
        F::template ext_mul<D>(c, e, t);
#pragma unroll
        for (int d = 0; d < D; d++) acc[d] = F::add(acc[d], t[d]);
    }
#pragma unroll
    for (int d = 0; d < D; d++) S[j * D + d] = acc[d];
}

// A[tile] = sum_{i < 1024} S[tile * 1024 + i] b^i
template <class F, int D>
__global__ __launch_bounds__(256) void syndiv_partials_kernel(const typename F::T *S, const typename F::T *pw, typename F::T *A) {
    typedef typename F::T T;
    __shared__ T buf[256][D];
    const uint32_t t = threadIdx.x;
    Pows<F, D> P{pw};
    T y[D], acc[D], xt[D], v[D];
    P.pow2(8, y);
#pragma unroll
    for (int d = 0; d < D; d++) acc[d] = F::zero();
    const T *c = S + ((uint64_t)blockIdx.x << TILE_LOG) * D;
    for (int m = 3; m >= 0; m--) horner_step<F, D, D>(acc, y, c + ((uint64_t)m * 256 + t) * D);
    P.tbl(t, xt);
    F::template ext_mul<D>(acc, xt, v);
    block_sum<F, D>(v, buf);
    if (t == 0) {
#pragma unroll
        for (int d = 0; d < D; d++) A[(uint64_t)blockIdx.x * D + d] = v[d];
    }
}

// Replays one tile (len_tile = 2^log_tile <= 1024 coefficients) with its carry H(tile end):
//   H(j) = S[j] + b * H(j + 1);   out[j - 1] (+)= H(j)  for j >= 1;   out[total - 1] (+)= 0.
// Lane t owns L = 2^log_l consecutive coefficients; the lanes' partial sums are chained by a suffix scan with the
// multipliers (b^L)^(2^s).
template <class F, int D>
__global__ __launch_bounds__(256) void syndiv_final_kernel(const typename F::T *S, const typename F::T *carries, uint32_t log_tile,
                                                           uint32_t log_l, const typename F::T *pw, typename F::T *out,
                                                           uint64_t total, int accumulate) {
    typedef typename F::T T;
    __shared__ T tile[(1 << TILE_LOG) * D];
    __shared__ T scan[256][D];
    const uint32_t t = threadIdx.x;
    const uint32_t len = 1u << log_tile, L = 1u << log_l, active = len >> log_l;
    const uint64_t t0 = (uint64_t)blockIdx.x << log_tile;
    Pows<F, D> P{pw};
    for (uint32_t w = t; w < len * D; w += 256) tile[w] = F::load_norm(S[t0 * D + w]);
    __syncthreads();
    T b[D], a[D];
    P.pow2(0, b);
#pragma unroll
    for (int d = 0; d < D; d++) a[d] = F::zero();
    if (t < active) {
        for (uint32_t i = L; i-- > 0;) horner_step<F, D, D>(a, b, &tile[(t * L + i) * D]);
        if (t == active - 1 && carries) {          // H(tile end) enters through the last lane: a += b^L * carry
            T bl[D], c[D], tmp[D];
            P.pow2(log_l, bl);
#pragma unroll
            for (int d = 0; d < D; d++) c[d] = carries[(uint64_t)blockIdx.x * D + d];
            F::template ext_mul<D>(bl, c, tmp);
#pragma unroll
            for (int d = 0; d < D; d++) a[d] = F::add(a[d], tmp[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < D; d++) scan[t][d] = a[d];
    __syncthreads();
    // inclusive suffix scan: V_t = sum_{u >= t} a_u (b^L)^(u - t)
    for (uint32_t s = 0; (1u << s) < active; s++) {
        T other[D], m[D], tmp[D];
        const bool has = t + (1u << s) < active;
        if (has) {
#pragma unroll
            for (int d = 0; d < D; d++) other[d] = scan[t + (1u << s)][d];
        }
        __syncthreads();
        if (has) {
            P.pow2(log_l + s, m);
            F::template ext_mul<D>(m, other, tmp);
#pragma unroll
            for (int d = 0; d < D; d++) scan[t][d] = F::add(scan[t][d], tmp[d]);
        }
        __syncthreads();
    }
    if (t >= active) return;
    T h[D];                                         // H(end of this lane's chunk)
    if (t + 1 < active) {
#pragma unroll
        for (int d = 0; d < D; d++) h[d] = scan[t + 1][d];
    } else {
#pragma unroll
        for (int d = 0; d < D; d++) h[d] = carries ? carries[(uint64_t)blockIdx.x * D + d] : F::zero();
        if (t0 + len == total && !accumulate) {     // q_{n-1} = 0
#pragma unroll
            for (int d = 0; d < D; d++) out[(total - 1) * D + d] = F::zero();
        }
    }
    for (uint32_t i = L; i-- > 0;) {
        horner_step<F, D, D>(h, b, &tile[(t * L + i) * D]);
        const uint64_t j = t0 + (uint64_t)t * L + i;
        if (j == 0) break;                          // H(0) is the dropped remainder
        T *o = out + (j - 1) * D;
#pragma unroll
        for (int d = 0; d < D; d++) o[d] = accumulate ? F::add(o[d], h[d]) : h[d];
    }
}

template <class HF, int D>
struct Deep {
    typedef typename HF::T T;
    typedef typename HF::Dev F;
    static constexpr size_t PW_WORDS = (size_t)(NPOW + NTBL) * D;

    static int make_pows(wf_ctx *ctx, const T *base, const T *src, int src_k, T *d_out) {
        Elems4<T, D> b;
        memset(&b, 0, sizeof(b));
        for (int d = 0; d < D; d++) b.e[0].v[d] = base ? base[d] : 0;
        hipLaunchKernelGGL((pows_kernel<F, D>), dim3(1), dim3(256), 0, ctx->stream, b, src, src_k, HF::to_internal(HF::from_u64(1)), d_out,
                           (uint64_t)PW_WORDS);
        WF_HIP(hipGetLastError());
        return WF_OK;
    }
    // tables of `count` <= 4 points (count * D words at `bases`) in one launch, table g at d_out + g * PW_WORDS
    static int make_pows_multi(wf_ctx *ctx, const T *bases, uint32_t count, T *d_out) {
        Elems4<T, D> b;
        memset(&b, 0, sizeof(b));
        for (uint32_t g = 0; g < count; g++)
            for (int d = 0; d < D; d++) b.e[g].v[d] = bases[g * D + d];
        hipLaunchKernelGGL((pows_kernel<F, D>), dim3(count), dim3(256), 0, ctx->stream, b, (const T *)nullptr, 0, HF::to_internal(HF::from_u64(1)),
                           d_out, (uint64_t)PW_WORDS);
        WF_HIP(hipGetLastError());
        return WF_OK;
    }

    // words of scratch syn_div needs for an array of 2^log_len elements (tile sums, carries and power tables of
    // every recursion level)
    static size_t syndiv_words(uint32_t log_len) {
        size_t w = 0;
        while (log_len > TILE_LOG) {
            log_len -= TILE_LOG;
            w += 2 * ((size_t)D << log_len) + PW_WORDS;
        }
        return w;
    }

    // out[j] (+)= sum_{k > j} S[k] b^(k - j - 1), pw = power tables of b
    static int syn_div(wf_ctx *ctx, const T *S, uint32_t log_len, const T *pw, T *out, bool accumulate, T *scratch) {
        const uint64_t len = 1ull << log_len;
        if (log_len <= TILE_LOG) {
            const uint32_t log_l = log_len >= 2 ? 2 : log_len;
            hipLaunchKernelGGL((syndiv_final_kernel<F, D>), dim3(1), dim3(256), 0, ctx->stream, S, (const T *)nullptr, log_len, log_l, pw,
                               out, len, accumulate ? 1 : 0);
            WF_HIP(hipGetLastError());
            return WF_OK;
        }
        const uint32_t log_tiles = log_len - TILE_LOG;
        const uint64_t tiles = 1ull << log_tiles;
        T *A = scratch, *C = A + tiles * D, *pw_up = C + tiles * D, *rest = pw_up + PW_WORDS;
        wf_prof_begin(ctx, "syndiv_partials");
        hipLaunchKernelGGL((syndiv_partials_kernel<F, D>), dim3((uint32_t)tiles), dim3(256), 0, ctx->stream, S, pw, A);
        wf_prof_end(ctx);
        WF_HIP(hipGetLastError());
        // carries: the same recurrence over the tile sums with base b^1024
        WF_TRY(make_pows(ctx, nullptr, pw, TILE_LOG, pw_up));
        WF_TRY(syn_div(ctx, A, log_tiles, pw_up, C, false, rest));
        wf_prof_begin(ctx, "syndiv_final");
        hipLaunchKernelGGL((syndiv_final_kernel<F, D>), dim3((uint32_t)tiles), dim3(256), 0, ctx->stream, S, (const T *)C, (uint32_t)TILE_LOG,
                           2u, pw, out, len, accumulate ? 1 : 0);
        wf_prof_end(ctx);
        WF_HIP(hipGetLastError());
        return WF_OK;
    }

    static int load_elem(const void *h, uint32_t idx, T (&o)[D]) {
        memcpy(o, (const uint8_t *)h + (size_t)idx * D * sizeof(T), D * sizeof(T));
        for (int d = 0; d < D; d++)
            if (!HF::valid_internal(o[d])) return WF_ERR_INVALID_ARG;
        return WF_OK;
    }

    // z * g (g = trace-domain generator, a base-field element): coordinate-wise
    static void mul_base(const T (&z)[D], T g_canon, T (&o)[D]) {
        for (int d = 0; d < D; d++) o[d] = HF::to_internal(HF::mulmod(HF::from_internal(z[d]), g_canon));
    }

    // the same with the point in device memory (d_point: one element; with_next: also at d_point * g, g = the generator of the 2^log_n
    // domain — the frame {z, z g} of get_ood_frame) and the values left on the device: d_out[num_points][num_cols] elements.
    // Nothing waits for the stream.
    template <int PD>
    static int evaluate_at_dev(wf_ctx *ctx, const void *d_polys, uint32_t num_cols, uint64_t col_stride, uint32_t log_n,
                               const void *d_point, bool with_next, void *d_out_user) {
        const uint32_t num_points = with_next ? 2 : 1;
        const uint32_t log_thr = log_n < 8 ? log_n : 8;
        uint32_t log_seg = log_n;
        while (log_seg > 10 && log_n - log_seg < 10 && ((uint64_t)num_cols << (log_n - log_seg)) < 4096) log_seg--;
        if (log_seg < log_thr) log_seg = log_thr;
        const uint32_t nseg = 1u << (log_n - log_seg);
        void *tmp;
        const size_t words = (size_t)num_points * PW_WORDS + (size_t)num_points * num_cols * nseg * D + 2 * D;
        WF_TRY(wf_scratch(ctx, 1, words * sizeof(T), &tmp));
        T *pw = (T *)tmp, *partials = pw + (size_t)num_points * PW_WORDS, *pts = partials + (size_t)num_points * num_cols * nseg * D;
        T *d_out = (T *)d_out_user;
        hipLaunchKernelGGL((ood_points_kernel<F, D>), dim3(1), dim3(64), 0, ctx->stream, (const T *)d_point,
                           HF::to_internal(HF::root_of_unity(log_n)), pts);
        WF_HIP(hipGetLastError());
        for (uint32_t g = 0; g < num_points; g++) WF_TRY(make_pows(ctx, nullptr, pts, (int)g, pw + g * PW_WORDS));
        const T *polys = (const T *)d_polys;
        wf_prof_begin(ctx, "poly_eval_at");
        const dim3 grid(nseg, num_cols), block(256);
        if (num_points == 2)
            hipLaunchKernelGGL((eval_partial_kernel<F, PD, D, 2>), grid, block, 0, ctx->stream, polys, col_stride, log_seg, log_thr, (const T *)pw,
                               (uint64_t)PW_WORDS, partials);
        else
            hipLaunchKernelGGL((eval_partial_kernel<F, PD, D, 1>), grid, block, 0, ctx->stream, polys, col_stride, log_seg, log_thr, (const T *)pw,
                               (uint64_t)PW_WORDS, partials);
        const uint32_t total = num_points * num_cols;
        if (nseg > 16)
            hipLaunchKernelGGL((eval_combine_wide_kernel<F, D>), dim3(total), dim3(256), 0, ctx->stream, (const T *)partials, num_cols,
                               log_n - log_seg, log_seg, (const T *)pw, (uint64_t)PW_WORDS, HF::to_internal(HF::from_u64(1)), d_out);
        else
            hipLaunchKernelGGL((eval_combine_kernel<F, D>), dim3((total + 63) / 64), dim3(64), 0, ctx->stream, (const T *)partials, num_cols, nseg,
                               log_seg, num_points, (const T *)pw, (uint64_t)PW_WORDS, d_out);
        wf_prof_end(ctx);
        WF_HIP(hipGetLastError());
        return WF_OK;
    }

    template <int PD>
    static int evaluate_at(wf_ctx *ctx, const void *d_polys, uint32_t num_cols, uint64_t col_stride, uint32_t log_n,
                           const void *h_points, uint32_t num_points, void *h_out) {
        // segments: enough workgroups to fill the chip, at least 256 coefficients each
        const uint32_t log_thr = log_n < 8 ? log_n : 8;
        // segments of >= 1024 coefficients (4 Horner steps per lane), up to 1024 of them per column, ~4096 workgroups in all
        uint32_t log_seg = log_n;
        while (log_seg > 10 && log_n - log_seg < 10 && ((uint64_t)num_cols << (log_n - log_seg)) < 4096) log_seg--;
        if (log_seg < log_thr) log_seg = log_thr;
        const uint32_t nseg = 1u << (log_n - log_seg);
        void *tmp;
        const size_t words = (size_t)num_points * PW_WORDS + (size_t)num_points * num_cols * nseg * D + (size_t)num_points * num_cols * D;
        WF_TRY(wf_scratch(ctx, 1, words * sizeof(T), &tmp));
        T *pw = (T *)tmp, *partials = pw + (size_t)num_points * PW_WORDS, *d_out = partials + (size_t)num_points * num_cols * nseg * D;
        for (uint32_t p = 0; p < num_points; p += 4) {
            T xs[4 * D];
            const uint32_t cnt = num_points - p < 4 ? num_points - p : 4;
            for (uint32_t g = 0; g < cnt; g++) {
                T x[D];
                WF_TRY(load_elem(h_points, p + g, x));
                for (int d = 0; d < D; d++) xs[g * D + d] = x[d];
            }
            WF_TRY(make_pows_multi(ctx, xs, cnt, pw + p * PW_WORDS));
        }
        const T *polys = (const T *)d_polys;
        wf_prof_begin(ctx, "poly_eval_at");
        for (uint32_t p = 0; p < num_points;) {
            const dim3 grid(nseg, num_cols), block(256);
            T *part = partials + (size_t)p * num_cols * nseg * D;
            if (num_points - p >= 2) {
                hipLaunchKernelGGL((eval_partial_kernel<F, PD, D, 2>), grid, block, 0, ctx->stream, polys, col_stride, log_seg, log_thr,
                                   (const T *)(pw + p * PW_WORDS), (uint64_t)PW_WORDS, part);
                p += 2;
            } else {
                hipLaunchKernelGGL((eval_partial_kernel<F, PD, D, 1>), grid, block, 0, ctx->stream, polys, col_stride, log_seg, log_thr,
                                   (const T *)(pw + p * PW_WORDS), (uint64_t)PW_WORDS, part);
                p += 1;
            }
        }
        const uint32_t total = num_points * num_cols;
        if (nseg > 16)
            hipLaunchKernelGGL((eval_combine_wide_kernel<F, D>), dim3(total), dim3(256), 0, ctx->stream, (const T *)partials, num_cols,
                               log_n - log_seg, log_seg, (const T *)pw, (uint64_t)PW_WORDS, HF::to_internal(HF::from_u64(1)), d_out);
        else
            hipLaunchKernelGGL((eval_combine_kernel<F, D>), dim3((total + 63) / 64), dim3(64), 0, ctx->stream, (const T *)partials, num_cols, nseg,
                               log_seg, num_points, (const T *)pw, (uint64_t)PW_WORDS, d_out);
        wf_prof_end(ctx);
        WF_HIP(hipGetLastError());
        WF_TRY(wf_copy_d2h(ctx, h_out, d_out, (size_t)total * D * sizeof(T)));
        return wf_check_status(ctx);
    }

    static int compose(wf_ctx *ctx, const void *d_main, uint32_t c_main, uint64_t main_stride, const void *d_aux, uint32_t c_aux,
                       uint64_t aux_stride, const void *d_quot, uint32_t c_q, uint64_t q_stride, uint32_t log_n, const void *h_z,
                       const void *h_cc_trace, const void *h_cc_constraints, void *d_out, const void *d_z = nullptr,
                       const void *d_cc_all = nullptr) {
        // d_z / d_cc_all: the point and the c_main + c_aux + c_q coefficients (trace columns first) in DEVICE memory, where a device
        // coin drew them (wf_deep_compose_dev): nothing is copied and nothing waits for the stream
        const uint64_t n = 1ull << log_n;
        const uint32_t c_total = c_main + c_aux + c_q;
        const bool dev = d_z != nullptr;
        T z[D], zg[D];
        std::vector<T> cc(dev ? 0 : (size_t)c_total * D);
        if (!dev) {
            WF_TRY(load_elem(h_z, 0, z));
            mul_base(z, HF::root_of_unity(log_n), zg);
            for (uint32_t i = 0; i < c_total; i++) {
                T e[D];
                WF_TRY(i < c_main + c_aux ? load_elem(h_cc_trace, i, e) : load_elem(h_cc_constraints, i - c_main - c_aux, e));
                for (int d = 0; d < D; d++) cc[(size_t)i * D + d] = e[d];
            }
        }
        void *tmp0, *tmp1;
        const size_t cc_words = (size_t)c_total * D;
        const size_t small_words = 2 * PW_WORDS + cc_words + 2 * D + syndiv_words(log_n);
        WF_TRY(wf_scratch(ctx, 0, (size_t)n * D * sizeof(T), &tmp0));
        WF_TRY(wf_scratch(ctx, 1, small_words * sizeof(T), &tmp1));
        T *S = (T *)tmp0;
        T *pw_z = (T *)tmp1, *pw_zg = pw_z + PW_WORDS, *d_cc = pw_zg + PW_WORDS, *pts = d_cc + cc_words, *rest = pts + 2 * D;
        if (dev) {
            hipLaunchKernelGGL((ood_points_kernel<F, D>), dim3(1), dim3(64), 0, ctx->stream, (const T *)d_z,
                               HF::to_internal(HF::root_of_unity(log_n)), pts);
            WF_HIP(hipGetLastError());
            WF_TRY(make_pows(ctx, nullptr, pts, 0, pw_z));
            WF_TRY(make_pows(ctx, nullptr, pts, 1, pw_zg));
            d_cc = (T *)d_cc_all;
        } else {
            WF_TRY(wf_copy_h2d(ctx, d_cc, cc.data(), cc.size() * sizeof(T)));   // synchronises: cc is a stack-lifetime host buffer
            WF_TRY(make_pows(ctx, z, nullptr, 0, pw_z));
            WF_TRY(make_pows(ctx, zg, nullptr, 0, pw_zg));
        }
        wf_prof_begin(ctx, "deep_acc");
        hipLaunchKernelGGL((deep_acc_kernel<F, D>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const T *)d_main, c_main,
                           main_stride, (const T *)d_aux, c_aux, aux_stride, (const T *)d_quot, c_q, q_stride, n, (const T *)d_cc, S);
        wf_prof_end(ctx);
        WF_HIP(hipGetLastError());
        // merge_compositions: S / (x - z) + S / (x - z g)
        WF_TRY(syn_div(ctx, S, log_n, pw_z, (T *)d_out, false, rest));
        WF_TRY(syn_div(ctx, S, log_n, pw_zg, (T *)d_out, true, rest));
        return WF_OK;
    }
};

template <class HF>
int evaluate_at_dispatch(wf_ctx *ctx, uint32_t pD, uint32_t D, const void *d_polys, uint32_t num_cols, uint64_t col_stride,
                         uint32_t log_n, const void *h_points, uint32_t num_points, void *h_out) {
    if (D < 1 || D > (uint32_t)HF::Dev::MAX_EXT || (pD != 1 && pD != D)) return WF_ERR_UNSUPPORTED;
    if (log_n > HF::TWO_ADICITY || log_n > 32) return WF_ERR_DOMAIN_TOO_LARGE;
    if (D == 1) return Deep<HF, 1>::template evaluate_at<1>(ctx, d_polys, num_cols, col_stride, log_n, h_points, num_points, h_out);
    if (D == 2)
        return pD == 1 ? Deep<HF, 2>::template evaluate_at<1>(ctx, d_polys, num_cols, col_stride, log_n, h_points, num_points, h_out)
                       : Deep<HF, 2>::template evaluate_at<2>(ctx, d_polys, num_cols, col_stride, log_n, h_points, num_points, h_out);
    if constexpr (HF::Dev::MAX_EXT >= 3)
        return pD == 1 ? Deep<HF, 3>::template evaluate_at<1>(ctx, d_polys, num_cols, col_stride, log_n, h_points, num_points, h_out)
                       : Deep<HF, 3>::template evaluate_at<3>(ctx, d_polys, num_cols, col_stride, log_n, h_points, num_points, h_out);
    return WF_ERR_UNSUPPORTED;
}

template <class HF>
int evaluate_at_dev_dispatch(wf_ctx *ctx, uint32_t pD, uint32_t D, const void *d_polys, uint32_t num_cols, uint64_t col_stride, uint32_t log_n,
                             const void *d_point, bool with_next, void *d_out) {
    if (D < 1 || D > (uint32_t)HF::Dev::MAX_EXT || (pD != 1 && pD != D)) return WF_ERR_UNSUPPORTED;
    if (log_n > HF::TWO_ADICITY || log_n > 32) return WF_ERR_DOMAIN_TOO_LARGE;
    if (D == 1) return Deep<HF, 1>::template evaluate_at_dev<1>(ctx, d_polys, num_cols, col_stride, log_n, d_point, with_next, d_out);
    if (D == 2)
        return pD == 1 ? Deep<HF, 2>::template evaluate_at_dev<1>(ctx, d_polys, num_cols, col_stride, log_n, d_point, with_next, d_out)
                       : Deep<HF, 2>::template evaluate_at_dev<2>(ctx, d_polys, num_cols, col_stride, log_n, d_point, with_next, d_out);
    if constexpr (HF::Dev::MAX_EXT >= 3)
        return pD == 1 ? Deep<HF, 3>::template evaluate_at_dev<1>(ctx, d_polys, num_cols, col_stride, log_n, d_point, with_next, d_out)
                       : Deep<HF, 3>::template evaluate_at_dev<3>(ctx, d_polys, num_cols, col_stride, log_n, d_point, with_next, d_out);
    return WF_ERR_UNSUPPORTED;
}

template <class HF>
int compose_dev_dispatch(wf_ctx *ctx, uint32_t D, const void *d_main, uint32_t c_main, uint64_t main_stride, const void *d_aux, uint32_t c_aux,
                         uint64_t aux_stride, const void *d_quot, uint32_t c_q, uint64_t q_stride, uint32_t log_n, const void *d_z,
                         const void *d_cc, void *d_out) {
    if (D < 1 || D > (uint32_t)HF::Dev::MAX_EXT) return WF_ERR_UNSUPPORTED;
    if (log_n > HF::TWO_ADICITY || log_n > 32) return WF_ERR_DOMAIN_TOO_LARGE;
    if (D == 1)
        return Deep<HF, 1>::compose(ctx, d_main, c_main, main_stride, d_aux, c_aux, aux_stride, d_quot, c_q, q_stride, log_n, nullptr, nullptr,
                                    nullptr, d_out, d_z, d_cc);
    if (D == 2)
        return Deep<HF, 2>::compose(ctx, d_main, c_main, main_stride, d_aux, c_aux, aux_stride, d_quot, c_q, q_stride, log_n, nullptr, nullptr,
                                    nullptr, d_out, d_z, d_cc);
    if constexpr (HF::Dev::MAX_EXT >= 3)
        return Deep<HF, 3>::compose(ctx, d_main, c_main, main_stride, d_aux, c_aux, aux_stride, d_quot, c_q, q_stride, log_n, nullptr, nullptr,
                                    nullptr, d_out, d_z, d_cc);
    return WF_ERR_UNSUPPORTED;
}

template <class HF>
int compose_dispatch(wf_ctx *ctx, uint32_t D, const void *d_main, uint32_t c_main, uint64_t main_stride, const void *d_aux,
                     uint32_t c_aux, uint64_t aux_stride, const void *d_quot, uint32_t c_q, uint64_t q_stride, uint32_t log_n,
                     const void *h_z, const void *h_cc_trace, const void *h_cc_constraints, void *d_out) {
    if (D < 1 || D > (uint32_t)HF::Dev::MAX_EXT) return WF_ERR_UNSUPPORTED;
    if (log_n > HF::TWO_ADICITY || log_n > 32) return WF_ERR_DOMAIN_TOO_LARGE;
    if (D == 1)
        return Deep<HF, 1>::compose(ctx, d_main, c_main, main_stride, d_aux, c_aux, aux_stride, d_quot, c_q, q_stride, log_n, h_z,
                                    h_cc_trace, h_cc_constraints, d_out);
    if (D == 2)
        return Deep<HF, 2>::compose(ctx, d_main, c_main, main_stride, d_aux, c_aux, aux_stride, d_quot, c_q, q_stride, log_n, h_z,
                                    h_cc_trace, h_cc_constraints, d_out);
    if constexpr (HF::Dev::MAX_EXT >= 3)
        return Deep<HF, 3>::compose(ctx, d_main, c_main, main_stride, d_aux, c_aux, aux_stride, d_quot, c_q, q_stride, log_n, h_z,
                                    h_cc_trace, h_cc_constraints, d_out);
    return WF_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int wf_polys_evaluate_at(wf_ctx *ctx, int field, uint32_t poly_ext_degree, uint32_t ext_degree, const void *d_polys,
                                    uint32_t num_cols, uint64_t col_stride, uint32_t log_n, const void *h_points,
                                    uint32_t num_points, void *h_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_polys || !h_points || !h_out) return WF_ERR_INVALID_ARG;
    if (num_cols == 0 || num_points == 0) return WF_OK;
    if (col_stride < ((uint64_t)poly_ext_degree << log_n) || num_cols > 65535) return WF_ERR_INVALID_ARG;
    switch (field) {
        case WF_FIELD_F64: return evaluate_at_dispatch<HostF64>(ctx, poly_ext_degree, ext_degree, d_polys, num_cols, col_stride, log_n, h_points, num_points, h_out);
        case WF_FIELD_F128: return evaluate_at_dispatch<HostF128>(ctx, poly_ext_degree, ext_degree, d_polys, num_cols, col_stride, log_n, h_points, num_points, h_out);
        case WF_FIELD_F62: return evaluate_at_dispatch<HostF62>(ctx, poly_ext_degree, ext_degree, d_polys, num_cols, col_stride, log_n, h_points, num_points, h_out);
        default: return WF_ERR_UNSUPPORTED;
    }
}

extern "C" int wf_deep_compose(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_main_polys, uint32_t num_main,
                               uint64_t main_stride, const void *d_aux_polys, uint32_t num_aux, uint64_t aux_stride,
                               const void *d_quotient_polys, uint32_t num_quotient, uint64_t quotient_stride, uint32_t log_n,
                               const void *h_z, const void *h_cc_trace, const void *h_cc_constraints, void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !h_z || !d_out || log_n == 0) return WF_ERR_INVALID_ARG;
    if ((num_main && (!d_main_polys || main_stride < (1ull << log_n))) ||
        (num_aux && (!d_aux_polys || aux_stride < ((uint64_t)ext_degree << log_n))) ||
        (num_quotient && (!d_quotient_polys || quotient_stride < ((uint64_t)ext_degree << log_n))))
        return WF_ERR_INVALID_ARG;
    if (num_main + num_aux + num_quotient == 0) return WF_ERR_INVALID_ARG;
    if ((num_main + num_aux && !h_cc_trace) || (num_quotient && !h_cc_constraints)) return WF_ERR_INVALID_ARG;
    switch (field) {
        case WF_FIELD_F64: return compose_dispatch<HostF64>(ctx, ext_degree, d_main_polys, num_main, main_stride, d_aux_polys, num_aux, aux_stride, d_quotient_polys, num_quotient, quotient_stride, log_n, h_z, h_cc_trace, h_cc_constraints, d_out);
        case WF_FIELD_F128: return compose_dispatch<HostF128>(ctx, ext_degree, d_main_polys, num_main, main_stride, d_aux_polys, num_aux, aux_stride, d_quotient_polys, num_quotient, quotient_stride, log_n, h_z, h_cc_trace, h_cc_constraints, d_out);
        case WF_FIELD_F62: return compose_dispatch<HostF62>(ctx, ext_degree, d_main_polys, num_main, main_stride, d_aux_polys, num_aux, aux_stride, d_quotient_polys, num_quotient, quotient_stride, log_n, h_z, h_cc_trace, h_cc_constraints, d_out);
        default: return WF_ERR_UNSUPPORTED;
    }
}

extern "C" int wf_polys_evaluate_at_dev(wf_ctx *ctx, int field, uint32_t poly_ext_degree, uint32_t ext_degree, const void *d_polys,
                                        uint32_t num_cols, uint64_t col_stride, uint32_t log_n, const void *d_point, int with_next,
                                        void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_polys || !d_point || !d_out) return WF_ERR_INVALID_ARG;
    if (num_cols == 0) return WF_OK;
    if (col_stride < ((uint64_t)poly_ext_degree << log_n) || num_cols > 65535) return WF_ERR_INVALID_ARG;   // grid.y = num_cols
    switch (field) {
        case WF_FIELD_F64: return evaluate_at_dev_dispatch<HostF64>(ctx, poly_ext_degree, ext_degree, d_polys, num_cols, col_stride, log_n, d_point, with_next != 0, d_out);
        case WF_FIELD_F128: return evaluate_at_dev_dispatch<HostF128>(ctx, poly_ext_degree, ext_degree, d_polys, num_cols, col_stride, log_n, d_point, with_next != 0, d_out);
        case WF_FIELD_F62: return evaluate_at_dev_dispatch<HostF62>(ctx, poly_ext_degree, ext_degree, d_polys, num_cols, col_stride, log_n, d_point, with_next != 0, d_out);
        default: return WF_ERR_UNSUPPORTED;
    }
}

extern "C" int wf_deep_compose_dev(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_main_polys, uint32_t num_main,
                                   uint64_t main_stride, const void *d_aux_polys, uint32_t num_aux, uint64_t aux_stride,
                                   const void *d_quotient_polys, uint32_t num_quotient, uint64_t quotient_stride, uint32_t log_n,
                                   const void *d_z, const void *d_cc, void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_z || !d_cc || !d_out || log_n == 0) return WF_ERR_INVALID_ARG;
    // the same argument checks as the host-coin twin (a short stride would make deep_acc_kernel read out of bounds)
    if ((num_main && (!d_main_polys || main_stride < (1ull << log_n))) ||
        (num_aux && (!d_aux_polys || aux_stride < ((uint64_t)ext_degree << log_n))) ||
        (num_quotient && (!d_quotient_polys || quotient_stride < ((uint64_t)ext_degree << log_n))))
        return WF_ERR_INVALID_ARG;
    if (num_main + num_aux + num_quotient == 0) return WF_ERR_INVALID_ARG;
    switch (field) {
        case WF_FIELD_F64: return compose_dev_dispatch<HostF64>(ctx, ext_degree, d_main_polys, num_main, main_stride, d_aux_polys, num_aux, aux_stride, d_quotient_polys, num_quotient, quotient_stride, log_n, d_z, d_cc, d_out);
        case WF_FIELD_F128: return compose_dev_dispatch<HostF128>(ctx, ext_degree, d_main_polys, num_main, main_stride, d_aux_polys, num_aux, aux_stride, d_quotient_polys, num_quotient, quotient_stride, log_n, d_z, d_cc, d_out);
        case WF_FIELD_F62: return compose_dev_dispatch<HostF62>(ctx, ext_degree, d_main_polys, num_main, main_stride, d_aux_polys, num_aux, aux_stride, d_quotient_polys, num_quotient, quotient_stride, log_n, d_z, d_cc, d_out);
        default: return WF_ERR_UNSUPPORTED;
    }
}
