// extern "C" entry points standing behind math::fft and prover::matrix (see include/winterfell_hip.h).
// Field-generic: every implementation is a template over the host field policy and dispatched on the `field` id.
#include <string.h>

#include "dft_regs.cuh"
#include "tables.cuh"
#include "wf_internal.h"

namespace {

template <class F>
__global__ void twiddles_kernel(const typename F::T *lo, const typename F::T *hi, uint32_t log_lo, uint32_t log_n, int inverse,
                                typename F::T *out) {
    // out[bitrev(i)] = omega^i (or omega^-i), i < n/2   — get_twiddles = power series then permute (mod.rs:464-467)
    const uint64_t half = 1ull << (log_n - 1);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const uint64_t n = 1ull << log_n;
    const uint64_t e = inverse ? ((n - i) & (n - 1)) : i;
    const uint32_t bits = log_n - 1;
    const uint64_t r = bits ? (__brevll(i) >> (64 - bits)) : 0;
    out[r] = series_at<F>(lo, hi, log_lo, e);
}

// tmp[bc][u][m]  ->  lde[(u + b*m)][bc] (row-major, zero padded to row_width).  One workgroup moves a tile of
// TM consecutive m x all b cosets x a group of 8 base columns through LDS so that both sides are coalesced:
// reads are runs of TM elements along m; writes are 8-element row segments of b*TM consecutive LDE rows
// (one contiguous region when row_width = 8).  LDS layout [ml][u][c] with rows padded by one element.
template <class T>
__global__ __launch_bounds__(256) void lde_transpose_kernel(const T *tmp, T *lde, uint32_t base_cols, uint64_t row_width,
                                                           uint32_t log_n, uint32_t log_b, uint32_t log_tm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *tile = reinterpret_cast<T *>(smem);
    const uint64_t n = 1ull << log_n;
    const uint32_t b = 1u << log_b, TM = 1u << log_tm;
    const uint32_t row = b * 8 + 1;
    const uint64_t m_tiles = (n + TM - 1) >> log_tm;
    const uint64_t mt = blockIdx.x % m_tiles;
    const uint32_t cg = (uint32_t)(blockIdx.x / m_tiles);   // column group of 8
    const uint32_t total = b * 8 * TM;
    for (uint32_t idx = threadIdx.x; idx < total; idx += 256) {
        const uint32_t ml = idx & (TM - 1), uc = idx >> log_tm;
        const uint32_t u = uc & (b - 1), cl = uc >> log_b;
        const uint32_t bc = cg * 8 + cl;
        const uint64_t m = (mt << log_tm) + ml;
        T v = 0;
        if (bc < base_cols && m < n) v = tmp[(((uint64_t)bc << log_b) + u) * n + m];
        tile[ml * row + u * 8 + cl] = v;
    }
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < total; idx += 256) {
        const uint32_t cl = idx & 7, u = (idx >> 3) & (b - 1), ml = idx >> (3 + log_b);
        const uint64_t m = (mt << log_tm) + ml;
        if (m < n) lde[(u + ((uint64_t)m << log_b)) * row_width + cg * 8 + cl] = tile[ml * row + u * 8 + cl];
    }
}

template <class HF>
int check_ext(uint32_t ext_degree) {
    return (ext_degree >= 1 && ext_degree <= (uint32_t)HF::Dev::MAX_EXT) ? WF_OK : WF_ERR_UNSUPPORTED;
}
template <class HF>
int check_domain(uint32_t log_n) {
    return (log_n > HF::TWO_ADICITY || log_n > 32) ? WF_ERR_DOMAIN_TOO_LARGE : WF_OK;
}

template <class HF>
void set_const(NttJob &j, typename HF::T canon) {
    const typename HF::T v = HF::to_internal(canon);
    j.has_post_const = true;
    memcpy(j.post_const, &v, sizeof(v));
}

// ---- implementations -------------------------------------------------------------------------------------
template <class HF>
int get_twiddles(wf_ctx *ctx, uint32_t log_n, int inverse, void *d_out) {
    typedef typename HF::Dev F;
    WF_TRY(check_domain<HF>(log_n));
    SeriesTable om;
    WF_TRY(wf_get_omega_table<HF>(ctx, log_n, &om));
    const uint64_t half = 1ull << (log_n - 1);
    wf_prof_begin(ctx, "twiddles");
    hipLaunchKernelGGL(twiddles_kernel<F>, dim3((uint32_t)((half + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const typename F::T *)om.d_lo, (const typename F::T *)om.d_hi, om.log_lo, log_n, inverse,
                       (typename F::T *)d_out);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

template <class HF>
int fft_inplace_batch(wf_ctx *ctx, uint32_t D, void *d, uint32_t log_n, uint32_t batch, bool inverse, uint64_t col_stride) {
    WF_TRY(check_ext<HF>(D));
    WF_TRY(check_domain<HF>(log_n));
    if (log_n == 0) return WF_OK;  // a constant polynomial evaluates to itself
    NttJob j;
    j.field = HF::Dev::ID;
    j.src = d;
    j.dst = d;
    j.log_n = log_n;
    j.nvec = batch * D;
    j.src_inner = j.dst_inner = D;
    j.src_vec_stride = j.dst_vec_stride = col_stride;
    j.src_es = j.dst_es = D;
    j.inverse = inverse;
    if (inverse) set_const<HF>(j, HF::invmod(HF::from_u64(1ull << log_n)));
    return wf_ntt_run(ctx, j);
}

template <class HF>
int evaluate_with_offset(wf_ctx *ctx, uint32_t D, const void *d_p, uint32_t log_n, const void *h_offset, uint32_t log_blowup,
                         void *d_result) {
    WF_TRY(check_ext<HF>(D));
    WF_TRY(check_domain<HF>(log_n + log_blowup));
    typename HF::T off;
    WF_TRY(wf_load_offset<HF>(h_offset, &off));
    wf_ctx::LdeTables t;
    uint64_t los, his;
    WF_TRY(wf_get_lde_tables<HF>(ctx, off, log_n, log_blowup, &t, &los, &his));
    const uint32_t b = 1u << log_blowup;
    // output vector v = d*b + u : element m at  d_result[(u + b*m)*D + d]
    NttJob j;
    j.field = HF::Dev::ID;
    j.src = d_p;
    j.dst = d_result;
    j.log_n = log_n;
    j.nvec = D * b;
    j.src_div = b;          // input component d = v / b
    j.src_inner = 1;
    j.src_vec_stride = 1;
    j.src_es = D;
    j.dst_inner = b;        // (v / b) * 1 + (v % b) * D
    j.dst_vec_stride = 1;
    j.dst_inner_stride = D;
    j.dst_es = b * D;
    j.pre_lo = t.d_lo;
    j.pre_hi = t.d_hi;
    j.pre_log_lo = t.log_lo;
    j.pre_mod = b;
    j.pre_lo_stride = los;
    j.pre_hi_stride = his;
    return wf_ntt_run(ctx, j);
}

template <class HF>
int interpolate_with_offset(wf_ctx *ctx, uint32_t D, void *d_evals, uint32_t log_n, const void *h_offset) {
    WF_TRY(check_ext<HF>(D));
    WF_TRY(check_domain<HF>(log_n));
    typename HF::T off;
    WF_TRY(wf_load_offset<HF>(h_offset, &off));
    // coefficient k is scaled by (1/n) * offset^-k   (serial.rs:96-100)
    SeriesTable st;
    WF_TRY(wf_get_series_table<HF>(ctx, HF::invmod(off), HF::invmod(HF::from_u64(1ull << log_n)), log_n, &st));
    NttJob j;
    j.field = HF::Dev::ID;
    j.src = d_evals;
    j.dst = d_evals;
    j.log_n = log_n;
    j.nvec = D;
    j.src_inner = j.dst_inner = D;
    j.src_vec_stride = j.dst_vec_stride = (uint64_t)D << log_n;
    j.src_es = j.dst_es = D;
    j.inverse = true;
    j.post_lo = st.d_lo;
    j.post_hi = st.d_hi;
    j.post_log_lo = st.log_lo;
    return wf_ntt_run(ctx, j);
}

// hash >= 0: the caller also wants the row hashes (wf_build_trace_commitment, one partition); when the narrow-row path
// is taken and a row is one 8-column group, transpose and leaf hashing run as ONE kernel and *fused is set
template <class HF>
int evaluate_polys_over(wf_ctx *ctx, uint32_t D, const void *d_polys, uint32_t num_cols, uint64_t col_stride, uint32_t log_n,
                        uint32_t log_blowup, const void *h_offset, void *d_lde, int hash, void *d_leaves, int *fused) {
    typedef typename HF::T T;
    WF_TRY(check_ext<HF>(D));
    WF_TRY(check_domain<HF>(log_n + log_blowup));
    if (col_stride < ((uint64_t)D << log_n)) return WF_ERR_INVALID_ARG;
    T off;
    WF_TRY(wf_load_offset<HF>(h_offset, &off));
    wf_ctx::LdeTables t;
    uint64_t los, his;
    WF_TRY(wf_get_lde_tables<HF>(ctx, off, log_n, log_blowup, &t, &los, &his));
    const uint32_t b = 1u << log_blowup;
    const uint32_t base_cols = num_cols * D;
    const uint64_t n = 1ull << log_n;
    const uint64_t row_width = wf_row_width(num_cols, D);
    if (log_blowup > 8) return WF_ERR_DOMAIN_TOO_LARGE;
    NttJob j;
    j.field = HF::Dev::ID;
    j.src = d_polys;
    j.log_n = log_n;
    j.nvec = base_cols * b;
    j.src_div = b;              // input base column bc = v / b
    j.src_inner = D;            // (bc / D) * col_stride + (bc % D)
    j.src_vec_stride = col_stride;
    j.src_es = D;
    j.dst_vec_stride = n;
    j.pre_lo = t.d_lo;
    j.pre_hi = t.d_hi;
    j.pre_log_lo = t.log_lo;
    j.pre_mod = b;
    j.pre_lo_stride = los;
    j.pre_hi_stride = his;
    // Output vector v = bc*b + u is coset u of base column bc; element m of it is LDE row u + b*m.
#ifndef WF_RM_MAX_LOG_I
#define WF_RM_MAX_LOG_I 5
#endif
    // columns per contiguous row segment of the fused row-major store: as many as the matrix has, up to 32 (a tile of the
    // last pass spans 16 or 32 (vector, column) pairs, so a row then receives 128-512 contiguous bytes per workgroup)
    uint32_t log_i = 0;
    const uint32_t max_log_i = sizeof(T) > 8 && WF_RM_MAX_LOG_I > 4 ? 4 : WF_RM_MAX_LOG_I;   // 256-byte segments
    while (log_i < max_log_i && (2u << log_i) <= base_cols) log_i++;
    // Blake3_256 leaves wanted and a padded row fits the columns of a last-pass tile (f64, <= 32 columns; <= 16 when the last pass has
    // radix 256): the last pass itself stores the rows, stages them in LDS and hashes them (ntt_pass<..., RH>) — for rows wider than
    // one 8-column group this replaces the row-major store below AND the row-hash kernel's second read of the whole matrix
#ifdef WF_NO_ROWS_HASH_PASS      // A/B builds: exactly the paths a build without the rows + leaves pass takes (wide rows: fused row-major store)
    const bool rows_hash = false;
#else
    const bool rows_hash = hash == WF_HASH_BLAKE3_256 && d_leaves && (row_width == 8 || ctx->rows_hash_wide) &&
                           wf_ntt_rows_mode_ok(HF::Dev::ID, log_n, log_blowup, base_cols);
#endif
    if (!rows_hash && ((size_t)sizeof(T) << log_i) >= 64) {
        // wide rows: the last pass stores straight into the row-major matrix, >= 64 contiguous bytes per row and column
        // group, and zeroes the padding columns (NttJob::rowmajor).  Measured on 64 x 2^22 f128 columns: the separate
        // transpose (12.8 ms of 154) disappears and the last pass costs the same.
        j.dst = d_lde;
        j.rowmajor = true;
        j.rm_log_b = log_blowup;
        j.rm_base_cols = base_cols;
        j.rm_row_width = row_width;
        j.rm_log_i = log_i;
        return wf_ntt_run(ctx, j);
    }
    if (rows_hash) {
        // narrow rows, Blake3_256 leaves wanted: the last pass itself assembles the rows in LDS, hashes them and stores rows +
        // leaves (ntt_pass<..., RH>): no coset-major buffer, no transpose launch
        j.dst = d_lde;
        j.rm_log_b = log_blowup;
        j.rm_base_cols = base_cols;
        j.rm_row_width = row_width;
        j.rh_leaves = d_leaves;
        WF_TRY(wf_ntt_run(ctx, j));
        *fused = 1;
        return WF_OK;
    }
    // narrow rows (fewer than 64 bytes of real columns): scattered 8..32-byte stores cost more than they save (measured
    // 299 vs 278 us for 4 f64 columns x 2^20 rows), so the cosets go to a coset-major buffer tmp[bc][u][m] first ...
    void *tmpv;
    WF_TRY(wf_scratch(ctx, 1, (size_t)base_cols * b * n * sizeof(T), &tmpv));
    j.dst = tmpv;
    WF_TRY(wf_ntt_run(ctx, j));
    // ... and one transpose through LDS writes the row-major matrix (and zero-fills the padding columns)
    uint32_t log_tm = log_blowup <= 3 ? 5 : (log_blowup >= 6 ? 2 : 8 - log_blowup);
    if (sizeof(T) > 8 && log_tm > 2) log_tm -= 1;   // keep the LDS tile <= ~33 KiB for 16-byte elements
    if (log_tm > log_n) log_tm = log_n;
    const uint64_t m_tiles = (n + (1ull << log_tm) - 1) >> log_tm;
    const uint64_t blocks = m_tiles * (row_width / 8);
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    const size_t lds_bytes = ((size_t)1 << log_tm) * (b * 8 + 1) * sizeof(T);
    if (hash >= 0 && row_width == 8 && ((uint64_t)b << log_tm) <= 256) {
        int done = 0;
        WF_TRY(wf_lde_transpose_hash(ctx, hash, HF::Dev::ID, D, tmpv, d_lde, base_cols, log_n, log_blowup, log_tm, d_leaves, &done));
        if (done) {
            *fused = 1;
            return WF_OK;
        }
    }
    wf_prof_begin(ctx, "lde_transpose");
    hipLaunchKernelGGL(lde_transpose_kernel<T>, dim3((uint32_t)blocks), dim3(256), lds_bytes, ctx->stream, (const T *)tmpv,
                       (T *)d_lde, base_cols, row_width, log_n, log_blowup, log_tm);
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

extern "C" int wf_fft_get_twiddles(wf_ctx *ctx, int field, uint32_t log_n, int inverse, void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_out || log_n == 0) return WF_ERR_INVALID_ARG;
    WF_DISPATCH_FIELD(field, get_twiddles, ctx, log_n, inverse, d_out);
}

extern "C" int wf_fft_evaluate_poly(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_p, uint32_t log_n, uint32_t batch) {
    WF_ENTER(ctx);
    if (!ctx || !d_p || batch == 0) return WF_ERR_INVALID_ARG;
    const uint64_t cs = (uint64_t)ext_degree << log_n;
    WF_DISPATCH_FIELD(field, fft_inplace_batch, ctx, ext_degree, d_p, log_n, batch, false, cs);
}

extern "C" int wf_fft_interpolate_poly(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_evals, uint32_t log_n,
                                       uint32_t batch) {
    WF_ENTER(ctx);
    if (!ctx || !d_evals || batch == 0) return WF_ERR_INVALID_ARG;
    const uint64_t cs = (uint64_t)ext_degree << log_n;
    WF_DISPATCH_FIELD(field, fft_inplace_batch, ctx, ext_degree, d_evals, log_n, batch, true, cs);
}

extern "C" int wf_fft_evaluate_poly_with_offset(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_p, uint32_t log_n,
                                                const void *h_offset, uint32_t log_blowup, void *d_result) {
    WF_ENTER(ctx);
    if (!ctx || !d_p || !d_result || log_n == 0) return WF_ERR_INVALID_ARG;
    WF_DISPATCH_FIELD(field, evaluate_with_offset, ctx, ext_degree, d_p, log_n, h_offset, log_blowup, d_result);
}

extern "C" int wf_fft_interpolate_poly_with_offset(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_evals, uint32_t log_n,
                                                   const void *h_offset) {
    WF_ENTER(ctx);
    if (!ctx || !d_evals || log_n == 0) return WF_ERR_INVALID_ARG;
    WF_DISPATCH_FIELD(field, interpolate_with_offset, ctx, ext_degree, d_evals, log_n, h_offset);
}

// ---- prover::matrix ---------------------------------------------------------------------------------
extern "C" uint64_t wf_row_width(uint32_t num_cols, uint32_t ext_degree) {
    const uint64_t bc = (uint64_t)num_cols * ext_degree;
    return 8 * ((bc + 7) / 8);
}

extern "C" int wf_interpolate_columns(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_cols, uint32_t num_cols,
                                      uint64_t col_stride, uint32_t log_n) {
    WF_ENTER(ctx);
    if (!ctx || !d_cols || num_cols == 0) return WF_ERR_INVALID_ARG;
    if (col_stride < ((uint64_t)ext_degree << log_n)) return WF_ERR_INVALID_ARG;
    WF_DISPATCH_FIELD(field, fft_inplace_batch, ctx, ext_degree, d_cols, log_n, num_cols, true, col_stride);
}

int wf_evaluate_polys_over_fused(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_polys, uint32_t num_cols, uint64_t col_stride,
                                 uint32_t log_n, uint32_t log_blowup, const void *h_offset, void *d_lde, int hash, void *d_leaves, int *fused) {
    *fused = 0;
    if (!ctx || !d_polys || !d_lde || num_cols == 0 || log_n == 0) return WF_ERR_INVALID_ARG;
    WF_DISPATCH_FIELD(field, evaluate_polys_over, ctx, ext_degree, d_polys, num_cols, col_stride, log_n, log_blowup, h_offset, d_lde, hash,
                      d_leaves, fused);
}

extern "C" int wf_evaluate_polys_over(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_polys, uint32_t num_cols,
                                      uint64_t col_stride, uint32_t log_n, uint32_t log_blowup, const void *h_offset,
                                      void *d_lde) {
    WF_ENTER(ctx);
    int fused;
    return wf_evaluate_polys_over_fused(ctx, field, ext_degree, d_polys, num_cols, col_stride, log_n, log_blowup, h_offset, d_lde, -1, nullptr,
                                        &fused);
}

extern "C" int wf_evaluate_columns_over(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_polys, uint32_t num_cols,
                                        uint64_t col_stride, uint32_t log_n, uint32_t log_blowup, const void *h_offset, void *d_out,
                                        uint64_t out_col_stride) {
    WF_ENTER(ctx);
    if (!ctx || !d_polys || !d_out || num_cols == 0 || log_n == 0 || ext_degree == 0) return WF_ERR_INVALID_ARG;
    if (col_stride < ((uint64_t)ext_degree << log_n) || out_col_stride < ((uint64_t)ext_degree << (log_n + log_blowup)))
        return WF_ERR_INVALID_ARG;
    const size_t es = field == WF_FIELD_F128 ? 16 : 8;
    // one coset evaluation per column (each already a batch of ext_degree * blowup size-n transforms)
    for (uint32_t k = 0; k < num_cols; k++) {
        const void *src = (const uint8_t *)d_polys + (size_t)k * col_stride * es;
        void *dst = (uint8_t *)d_out + (size_t)k * out_col_stride * es;
        WF_TRY(wf_fft_evaluate_poly_with_offset(ctx, field, ext_degree, src, log_n, h_offset, log_blowup, dst));
    }
    return WF_OK;
}
