// extern "C" entry points standing behind math::fft and prover::matrix (see include/winterfell_hip.h).
#include "gl64.cuh"
#include "wf_internal.h"

namespace {

__global__ void twiddles_kernel(const uint64_t *lo, const uint64_t *hi, uint32_t log_lo, uint32_t log_n, int inverse,
                                uint64_t *out) {
    // out[bitrev(i)] = omega^i (or omega^-i), i < n/2   — get_twiddles = power series then permute (mod.rs:464-467)
    const uint64_t half = 1ull << (log_n - 1);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const uint64_t n = 1ull << log_n;
    const uint64_t e = inverse ? ((n - i) & (n - 1)) : i;
    const uint64_t w = gl::mul(lo[e & ((1ull << log_lo) - 1)], hi[e >> log_lo]);
    const uint32_t bits = log_n - 1;
    uint64_t r = bits ? (__brevll(i) >> (64 - bits)) : 0;
    out[r] = w;
}

// tmp[bc][u][m]  ->  lde[(u + b*m)][bc] (row-major, zero padded to row_width).  One workgroup moves a tile of
// TM consecutive m x all b cosets x a group of 8 base columns through LDS so that both sides are coalesced:
// reads are runs of TM elements along m; writes are 64-byte row segments of b*TM consecutive LDE rows
// (one contiguous region when row_width = 8).  LDS layout [ml][u][c] with rows padded by one element.
__global__ __launch_bounds__(256) void lde_transpose_kernel(const uint64_t *tmp, uint64_t *lde, uint32_t base_cols,
                                                           uint64_t row_width, uint32_t log_n, uint32_t log_b,
                                                           uint32_t log_tm) {
    extern __shared__ __attribute__((aligned(16))) uint64_t tile[];
    const uint64_t n = 1ull << log_n;
    const uint32_t b = 1u << log_b, TM = 1u << log_tm;
    const uint32_t row = b * 8 + 1;
    const uint64_t m_tiles = (n + TM - 1) >> log_tm;
    const uint64_t mt = blockIdx.x % m_tiles;
    const uint32_t cg = (uint32_t)(blockIdx.x / m_tiles);   // column group of 8
    const uint32_t total = b * 8 * TM;
    for (uint32_t idx = threadIdx.x; idx < total; idx += 256) {
        const uint32_t ml = idx & (TM - 1), uc = idx >> log_tm;      // uc = c * b + u  (u fastest among the rest)
        const uint32_t u = uc & (b - 1), cl = uc >> log_b;
        const uint32_t bc = cg * 8 + cl;
        const uint64_t m = (mt << log_tm) + ml;
        uint64_t v = 0;
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

int check_field_ext(int field, uint32_t ext_degree) {
    if (field != WF_FIELD_F64) return WF_ERR_UNSUPPORTED;
    if (ext_degree < 1 || ext_degree > 3) return WF_ERR_UNSUPPORTED;
    return WF_OK;
}

int check_domain(uint32_t log_n) { return log_n > 32 ? WF_ERR_DOMAIN_TOO_LARGE : WF_OK; }

}  // namespace

// LDE pre-scale tables: for coset u (rows u + b*m of the LDE), series (offset * g^u)^j, j < n, g = omega_{n*b}.
static int get_lde_tables(wf_ctx *ctx, uint64_t offset_canon, uint32_t log_n, uint32_t log_b, wf_ctx::LdeTables *out,
                          uint64_t *lo_stride, uint64_t *hi_stride) {
    using namespace hostgl;
    const uint32_t log_lo = log_n < 12 ? log_n : 12;
    const uint64_t nlo = 1ull << log_lo, nhi = 1ull << (log_n - log_lo);
    *lo_stride = nlo;
    *hi_stride = nhi;
    auto key = std::make_tuple(offset_canon, log_n, log_b);
    auto it = ctx->lde_tables.find(key);
    if (it == ctx->lde_tables.end()) {
        const uint32_t b = 1u << log_b;
        const uint64_t g = root_of_unity(log_n + log_b);
        std::vector<uint64_t> lo(nlo * b), hi(nhi * b);
        for (uint32_t u = 0; u < b; u++) {
            const uint64_t base = mulmod(offset_canon, powmod(g, u));
            uint64_t cur = 1;
            for (uint64_t i = 0; i < nlo; i++) {
                lo[u * nlo + i] = to_mont(cur);
                cur = mulmod(cur, base);
            }
            const uint64_t step = cur;
            cur = 1;
            for (uint64_t i = 0; i < nhi; i++) {
                hi[u * nhi + i] = to_mont(cur);
                cur = mulmod(cur, step);
            }
        }
        wf_ctx::LdeTables t;
        t.log_lo = log_lo;
        void *p;
        WF_HIP(hipMalloc(&p, lo.size() * 8));
        ctx->owned.push_back(p);
        t.d_lo = (uint64_t *)p;
        WF_HIP(hipMalloc(&p, hi.size() * 8));
        ctx->owned.push_back(p);
        t.d_hi = (uint64_t *)p;
        WF_HIP(hipMemcpyAsync(t.d_lo, lo.data(), lo.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        WF_HIP(hipMemcpyAsync(t.d_hi, hi.data(), hi.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        WF_HIP(hipStreamSynchronize(ctx->stream));
        it = ctx->lde_tables.emplace(key, t).first;
    }
    *out = it->second;
    return WF_OK;
}

static int load_offset(const void *h_offset, uint64_t *canon) {
    if (!h_offset) return WF_ERR_INVALID_ARG;
    const uint64_t m = *(const uint64_t *)h_offset;
    if (m >= hostgl::P) return WF_ERR_INVALID_ARG;
    *canon = hostgl::from_mont(m);
    if (*canon == 0) return WF_ERR_ZERO_OFFSET;
    return WF_OK;
}

extern "C" int wf_fft_get_twiddles(wf_ctx *ctx, int field, uint32_t log_n, int inverse, void *d_out) {
    if (!ctx || !d_out || log_n == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_field_ext(field, 1));
    WF_TRY(check_domain(log_n));
    SeriesTable om;
    WF_TRY(wf_get_omega_table(ctx, log_n, &om));
    const uint64_t half = 1ull << (log_n - 1);
    wf_prof_begin(ctx, "twiddles");
    hipLaunchKernelGGL(twiddles_kernel, dim3((uint32_t)((half + 255) / 256)), dim3(256), 0, ctx->stream, om.d_lo, om.d_hi,
                       om.log_lo, log_n, inverse, (uint64_t *)d_out);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

static int fft_inplace_batch(wf_ctx *ctx, uint32_t D, void *d, uint32_t log_n, uint32_t batch, bool inverse,
                             uint64_t col_stride) {
    NttJob j;
    j.src = (const uint64_t *)d;
    j.dst = (uint64_t *)d;
    j.log_n = log_n;
    j.nvec = batch * D;
    j.src_inner = j.dst_inner = D;
    j.src_vec_stride = j.dst_vec_stride = col_stride;
    j.src_es = j.dst_es = D;
    j.inverse = inverse;
    if (inverse) j.post_const = hostgl::to_mont(hostgl::invmod(1ull << log_n));
    return wf_ntt_f64_run(ctx, j);
}

extern "C" int wf_fft_evaluate_poly(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_p, uint32_t log_n,
                                    uint32_t batch) {
    if (!ctx || !d_p || batch == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_field_ext(field, ext_degree));
    WF_TRY(check_domain(log_n));
    if (log_n == 0) return WF_OK;  // a constant polynomial evaluates to itself
    return fft_inplace_batch(ctx, ext_degree, d_p, log_n, batch, false, (uint64_t)ext_degree << log_n);
}

extern "C" int wf_fft_interpolate_poly(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_evals, uint32_t log_n,
                                       uint32_t batch) {
    if (!ctx || !d_evals || batch == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_field_ext(field, ext_degree));
    WF_TRY(check_domain(log_n));
    if (log_n == 0) return WF_OK;
    return fft_inplace_batch(ctx, ext_degree, d_evals, log_n, batch, true, (uint64_t)ext_degree << log_n);
}

extern "C" int wf_fft_evaluate_poly_with_offset(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_p,
                                                uint32_t log_n, const void *h_offset, uint32_t log_blowup,
                                                void *d_result) {
    if (!ctx || !d_p || !d_result || log_n == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_field_ext(field, ext_degree));
    WF_TRY(check_domain(log_n + log_blowup));
    uint64_t off;
    WF_TRY(load_offset(h_offset, &off));
    wf_ctx::LdeTables t;
    uint64_t los, his;
    WF_TRY(get_lde_tables(ctx, off, log_n, log_blowup, &t, &los, &his));
    const uint32_t D = ext_degree, b = 1u << log_blowup;
    // output vector v = d*b + u : element m at  d_result[(u + b*m)*D + d]
    NttJob j;
    j.src = (const uint64_t *)d_p;
    j.dst = (uint64_t *)d_result;
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
    return wf_ntt_f64_run(ctx, j);
}

extern "C" int wf_fft_interpolate_poly_with_offset(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_evals,
                                                   uint32_t log_n, const void *h_offset) {
    if (!ctx || !d_evals || log_n == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_field_ext(field, ext_degree));
    WF_TRY(check_domain(log_n));
    uint64_t off;
    WF_TRY(load_offset(h_offset, &off));
    // coefficient k is scaled by (1/n) * offset^-k   (serial.rs:96-100)
    SeriesTable st;
    WF_TRY(wf_get_series_table(ctx, hostgl::invmod(off), hostgl::invmod(1ull << log_n), log_n, &st));
    const uint32_t D = ext_degree;
    NttJob j;
    j.src = (const uint64_t *)d_evals;
    j.dst = (uint64_t *)d_evals;
    j.log_n = log_n;
    j.nvec = D;
    j.src_inner = j.dst_inner = D;
    j.src_vec_stride = j.dst_vec_stride = (uint64_t)D << log_n;
    j.src_es = j.dst_es = D;
    j.inverse = true;
    j.post_lo = st.d_lo;
    j.post_hi = st.d_hi;
    j.post_log_lo = st.log_lo;
    return wf_ntt_f64_run(ctx, j);
}

// ---- prover::matrix ---------------------------------------------------------------------------------
extern "C" uint64_t wf_row_width(uint32_t num_cols, uint32_t ext_degree) {
    const uint64_t bc = (uint64_t)num_cols * ext_degree;
    return 8 * ((bc + 7) / 8);
}

extern "C" int wf_interpolate_columns(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_cols, uint32_t num_cols,
                                      uint64_t col_stride, uint32_t log_n) {
    if (!ctx || !d_cols || num_cols == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_field_ext(field, ext_degree));
    WF_TRY(check_domain(log_n));
    if (col_stride < ((uint64_t)ext_degree << log_n)) return WF_ERR_INVALID_ARG;
    if (log_n == 0) return WF_OK;
    return fft_inplace_batch(ctx, ext_degree, d_cols, log_n, num_cols, true, col_stride);
}

extern "C" int wf_evaluate_polys_over(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_polys,
                                      uint32_t num_cols, uint64_t col_stride, uint32_t log_n, uint32_t log_blowup,
                                      const void *h_offset, void *d_lde) {
    if (!ctx || !d_polys || !d_lde || num_cols == 0 || log_n == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_field_ext(field, ext_degree));
    WF_TRY(check_domain(log_n + log_blowup));
    if (col_stride < ((uint64_t)ext_degree << log_n)) return WF_ERR_INVALID_ARG;
    uint64_t off;
    WF_TRY(load_offset(h_offset, &off));
    wf_ctx::LdeTables t;
    uint64_t los, his;
    WF_TRY(get_lde_tables(ctx, off, log_n, log_blowup, &t, &los, &his));
    const uint32_t D = ext_degree, b = 1u << log_blowup;
    const uint32_t base_cols = num_cols * D;
    const uint64_t n = 1ull << log_n;
    const uint64_t row_width = wf_row_width(num_cols, D);
    void *tmpv;
    WF_TRY(wf_scratch(ctx, 1, (size_t)base_cols * b * n * 8, &tmpv));
    uint64_t *tmp = (uint64_t *)tmpv;
    // coset transforms: output vector v = bc*b + u  ->  tmp[bc][u][m]
    NttJob j;
    j.src = (const uint64_t *)d_polys;
    j.dst = tmp;
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
    WF_TRY(wf_ntt_f64_run(ctx, j));
    // transpose into the row-major matrix (zero-fills the padding columns)
    uint32_t log_tm = log_blowup <= 3 ? 5 : (log_blowup >= 6 ? 2 : 8 - log_blowup);
    if (log_tm > log_n) log_tm = log_n;
    const uint64_t m_tiles = (n + (1ull << log_tm) - 1) >> log_tm;
    const uint64_t blocks = m_tiles * (row_width / 8);
    if (blocks > 0x7fffffffull || log_blowup > 8) return WF_ERR_DOMAIN_TOO_LARGE;
    const size_t lds_bytes = ((size_t)1 << log_tm) * (b * 8 + 1) * 8;
    wf_prof_begin(ctx, "lde_transpose");
    hipLaunchKernelGGL(lde_transpose_kernel, dim3((uint32_t)blocks), dim3(256), lds_bytes, ctx->stream, tmp,
                       (uint64_t *)d_lde, base_cols, row_width, log_n, log_blowup, log_tm);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}
