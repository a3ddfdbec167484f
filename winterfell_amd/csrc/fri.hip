// FRI commit phase on the GPU (f64 field and its extensions).
//
// Reference behaviour reproduced (fri/src/prover/mod.rs:179-239, 321-336; fri/src/folding/mod.rs:86-118,181-188;
// utils/core/src/lib.rs:166-183):
//   layer commit: t[i][j] = e[i + j*len/N]  (transpose_slice), leaf_i = H::hash_elements(t[i]), MerkleTree over leaves
//   fold:         per row i: N-point inverse DFT of t[i], coefficient k scaled by (1/N) * (offset^-1 * g^-i)^k,
//                 Horner evaluation at alpha  (apply_drp); the same domain offset is used at every layer (mod.rs:216)
// Layers are inherently sequential (alpha_k depends on root_k): the host draws alpha between the two calls.
#include "dft_regs.cuh"
#include "gl64.cuh"
#include "wf_internal.h"

namespace {

template <int D>
__global__ __launch_bounds__(256) void fri_transpose_kernel(const uint64_t *ev, uint64_t *out, uint32_t log_rc,
                                                            uint32_t log_nf) {
    // one row per lane: reads are coalesced along i for every j; a row (N*D words) is written contiguously
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t rc = 1ull << log_rc;
    if (i >= rc) return;
    const uint32_t N = 1u << log_nf;
    for (uint32_t j = 0; j < N; j++) {
#pragma unroll
        for (int d = 0; d < D; d++) out[(i * N + j) * D + d] = ev[(i + (uint64_t)j * rc) * D + d];
    }
}

template <int LOG_NF, int D>
__global__ __launch_bounds__(256) void fri_fold_kernel(const uint64_t *t, uint64_t *out, uint32_t log_rc,
                                                       const uint64_t *io_lo, const uint64_t *io_hi, uint32_t io_log_lo,
                                                       uint64_t inv_n, uint64_t a0, uint64_t a1, uint64_t a2) {
    constexpr int N = 1 << LOG_NF;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1ull << log_rc)) return;
    uint64_t comp[D][N];
#pragma unroll
    for (int j = 0; j < N; j++)
#pragma unroll
        for (int d = 0; d < D; d++) comp[d][j] = t[(i * N + j) * D + d];
    // forward DFT per component (bit-reversed registers); inverse coefficient k = X[(N - k) mod N]
#pragma unroll
    for (int d = 0; d < D; d++) dft_dif<LOG_NF>(comp[d]);
    const uint64_t io = series_at(io_lo, io_hi, io_log_lo, i);   // offset^-1 * g^-i
    uint64_t scale[N];
    scale[0] = inv_n;
#pragma unroll
    for (int k = 1; k < N; k++) scale[k] = gl::mul(scale[k - 1], io);
    const uint64_t alpha[3] = {a0, a1, a2};
    uint64_t al[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; d++) { al[d] = alpha[d]; acc[d] = 0; }
#pragma unroll
    for (int k = N - 1; k >= 0; k--) {
        uint64_t tmp[D];
        gl::ext_mul<D>(acc, al, tmp);
        const int src = brev((N - k) & (N - 1), LOG_NF);
#pragma unroll
        for (int d = 0; d < D; d++) acc[d] = gl::add(tmp[d], gl::mul(comp[d][src], scale[k]));
    }
#pragma unroll
    for (int d = 0; d < D; d++) out[i * D + d] = acc[d];
}

template <int D>
int launch_fold(wf_ctx *ctx, uint32_t log_nf, const uint64_t *t, uint64_t *out, uint32_t log_rc, const SeriesTable &io,
                uint64_t inv_n, const uint64_t *alpha) {
    const uint64_t rc = 1ull << log_rc;
    const dim3 grid((uint32_t)((rc + 255) / 256)), block(256);
    const uint64_t a0 = alpha[0], a1 = D > 1 ? alpha[1] : 0, a2 = D > 2 ? alpha[2] : 0;
    wf_prof_begin(ctx, "fri_fold");
    switch (log_nf) {
        case 1: hipLaunchKernelGGL((fri_fold_kernel<1, D>), grid, block, 0, ctx->stream, t, out, log_rc, io.d_lo, io.d_hi, io.log_lo, inv_n, a0, a1, a2); break;
        case 2: hipLaunchKernelGGL((fri_fold_kernel<2, D>), grid, block, 0, ctx->stream, t, out, log_rc, io.d_lo, io.d_hi, io.log_lo, inv_n, a0, a1, a2); break;
        case 3: hipLaunchKernelGGL((fri_fold_kernel<3, D>), grid, block, 0, ctx->stream, t, out, log_rc, io.d_lo, io.d_hi, io.log_lo, inv_n, a0, a1, a2); break;
        default: hipLaunchKernelGGL((fri_fold_kernel<4, D>), grid, block, 0, ctx->stream, t, out, log_rc, io.d_lo, io.d_hi, io.log_lo, inv_n, a0, a1, a2); break;
    }
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

int check_args(int field, uint32_t D, uint32_t log_len, uint32_t folding, uint32_t *log_nf) {
    if (field != WF_FIELD_F64 || D < 1 || D > 3) return WF_ERR_UNSUPPORTED;
    if (folding != 2 && folding != 4 && folding != 8 && folding != 16) return WF_ERR_UNSUPPORTED;  // mod.rs:187-195
    uint32_t l = 0;
    while ((1u << l) < folding) l++;
    if (log_len < l || log_len > 32) return WF_ERR_INVALID_ARG;
    *log_nf = l;
    return WF_OK;
}

}  // namespace

extern "C" int wf_fri_layer_commit(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_evals,
                                   uint32_t log_len, uint32_t folding, void *d_transposed, void *d_leaves, void *d_nodes,
                                   void *h_root) {
    if (!ctx || !d_evals || !d_transposed || !d_leaves || !d_nodes) return WF_ERR_INVALID_ARG;
    uint32_t log_nf;
    WF_TRY(check_args(field, ext_degree, log_len, folding, &log_nf));
    const uint32_t log_rc = log_len - log_nf;
    const uint64_t rc = 1ull << log_rc;
    const dim3 grid((uint32_t)((rc + 255) / 256)), block(256);
    const uint64_t *ev = (const uint64_t *)d_evals;
    uint64_t *tr = (uint64_t *)d_transposed;
    wf_prof_begin(ctx, "fri_transpose");
    if (ext_degree == 1) hipLaunchKernelGGL(fri_transpose_kernel<1>, grid, block, 0, ctx->stream, ev, tr, log_rc, log_nf);
    else if (ext_degree == 2) hipLaunchKernelGGL(fri_transpose_kernel<2>, grid, block, 0, ctx->stream, ev, tr, log_rc, log_nf);
    else hipLaunchKernelGGL(fri_transpose_kernel<3>, grid, block, 0, ctx->stream, ev, tr, log_rc, log_nf);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    const uint32_t row_words = folding * ext_degree;
    // build_layer_commitment: leaf = hash_elements(row); V::new(leaves)
    WF_TRY(wf_hash_elements_batch(ctx, hash, field, d_transposed, rc, row_words, row_words, d_leaves));
    WF_TRY(wf_merkle_build(ctx, hash, d_leaves, rc, d_nodes));
    if (h_root) WF_TRY(wf_memcpy_d2h(ctx, h_root, (const uint8_t *)d_nodes + 32, 32));
    return WF_OK;
}

extern "C" int wf_fri_apply_drp(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_transposed, uint32_t log_len,
                                uint32_t folding, const void *h_domain_offset, const void *h_alpha, void *d_folded) {
    if (!ctx || !d_transposed || !h_domain_offset || !h_alpha || !d_folded) return WF_ERR_INVALID_ARG;
    uint32_t log_nf;
    WF_TRY(check_args(field, ext_degree, log_len, folding, &log_nf));
    const uint32_t log_rc = log_len - log_nf;
    const uint64_t off_m = *(const uint64_t *)h_domain_offset;
    if (off_m >= hostgl::P) return WF_ERR_INVALID_ARG;
    const uint64_t off = hostgl::from_mont(off_m);
    if (off == 0) return WF_ERR_ZERO_OFFSET;
    // inv_offsets[i] = offset^-1 * (g^-1)^i, g = root of unity of the layer's domain (folding/mod.rs:181-188)
    SeriesTable io;
    const uint64_t g_inv = hostgl::invmod(hostgl::root_of_unity(log_len));
    WF_TRY(wf_get_series_table(ctx, g_inv, hostgl::invmod(off), log_rc == 0 ? 0 : log_rc, &io));
    const uint64_t inv_n = hostgl::to_mont(hostgl::invmod(folding));
    const uint64_t *alpha = (const uint64_t *)h_alpha;
    for (uint32_t d = 0; d < ext_degree; d++)
        if (alpha[d] >= hostgl::P) return WF_ERR_INVALID_ARG;
    const uint64_t *t = (const uint64_t *)d_transposed;
    uint64_t *out = (uint64_t *)d_folded;
    if (ext_degree == 1) return launch_fold<1>(ctx, log_nf, t, out, log_rc, io, inv_n, alpha);
    if (ext_degree == 2) return launch_fold<2>(ctx, log_nf, t, out, log_rc, io, inv_n, alpha);
    return launch_fold<3>(ctx, log_nf, t, out, log_rc, io, inv_n, alpha);
}
