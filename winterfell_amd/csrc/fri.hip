// FRI commit phase on the GPU (f64 field and its extensions).
//
// Reference behaviour reproduced (fri/src/prover/mod.rs:179-239, 321-336; fri/src/folding/mod.rs:86-118,181-188;
// utils/core/src/lib.rs:166-183):
//   layer commit: t[i][j] = e[i + j*len/N]  (transpose_slice), leaf_i = H::hash_elements(t[i]), MerkleTree over leaves
//   fold:         per row i: N-point inverse DFT of t[i], coefficient k scaled by (1/N) * (offset^-1 * g^-i)^k,
//                 Horner evaluation at alpha  (apply_drp); the same domain offset is used at every layer (mod.rs:216)
// Layers are inherently sequential (alpha_k depends on root_k): either the host draws alpha between the two calls
// (wf_fri_layer_commit / wf_fri_apply_drp), or the coin lives on the device and wf_fri_build_layers queues the whole loop.
#include <string.h>

#include "dft_regs.cuh"
#include "tables.cuh"
#include "wf_internal.h"

namespace {

template <class T, int D>
__global__ __launch_bounds__(256) void fri_transpose_kernel(const T *ev, T *out, uint32_t log_rc, uint32_t log_nf) {
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

template <class T>
struct FoldConsts {
    T inv_n;
    T alpha[3];
};

template <class F, int LOG_NF, int D>
__global__ __launch_bounds__(256) void fri_fold_kernel(const typename F::T *t, typename F::T *out, uint64_t row_start,
                                                       uint64_t num_rows, const typename F::T *io_lo, const typename F::T *io_hi,
                                                       uint32_t io_log_lo, const typename F::T *w16,
                                                       FoldConsts<typename F::T> cst, const typename F::T *d_alpha) {
    typedef typename F::T T;
    constexpr int N = 1 << LOG_NF;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // row within the shard
    if (i >= num_rows) return;
    T comp[D][N];
#pragma unroll
    for (int j = 0; j < N; j++)
#pragma unroll
        for (int d = 0; d < D; d++) comp[d][j] = F::load_norm(t[(i * N + j) * D + d]);
    // forward DFT per component (bit-reversed registers); inverse coefficient k = X[(N - k) mod N]
#pragma unroll
    for (int d = 0; d < D; d++) dft_dif<F, LOG_NF>(comp[d], w16);
    const T io = series_at<F>(io_lo, io_hi, io_log_lo, row_start + i);   // offset^-1 * g^-(global row)
    T scale[N];
    scale[0] = cst.inv_n;
#pragma unroll
    for (int k = 1; k < N; k++) scale[k] = F::mul(scale[k - 1], io);
    T al[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; d++) { al[d] = d_alpha ? d_alpha[d] : cst.alpha[d]; acc[d] = F::zero(); }   // alpha from the host, or where the device coin drew it
#pragma unroll
    for (int k = N - 1; k >= 0; k--) {
        T tmp[D];
        F::template ext_mul<D>(acc, al, tmp);
        const int src = brev((N - k) & (N - 1), LOG_NF);
#pragma unroll
        for (int d = 0; d < D; d++) acc[d] = F::add(tmp[d], F::mul(comp[d][src], scale[k]));
    }
#pragma unroll
    for (int d = 0; d < D; d++) out[i * D + d] = acc[d];
}

template <class HF, int D>
int launch_fold(wf_ctx *ctx, uint32_t log_nf, const void *t_, void *out_, uint64_t row_start, uint64_t rc, const SeriesTable &io,
                const FoldConsts<typename HF::T> &cst, const void *d_alpha_) {
    typedef typename HF::Dev F;
    typedef typename F::T T;
    const T *t = (const T *)t_;
    T *out = (T *)out_;
    const T *lo = (const T *)io.d_lo, *hi = (const T *)io.d_hi, *d_alpha = (const T *)d_alpha_;
    void *w256, *w16v;
    WF_TRY(wf_get_small_tables<HF>(ctx, &w256, &w16v));
    const T *w16 = (const T *)w16v;
    const dim3 grid((uint32_t)((rc + 255) / 256)), block(256);
    wf_prof_begin(ctx, "fri_fold");
    switch (log_nf) {
        case 1: hipLaunchKernelGGL((fri_fold_kernel<F, 1, D>), grid, block, 0, ctx->stream, t, out, row_start, rc, lo, hi, io.log_lo, w16, cst, d_alpha); break;
        case 2: hipLaunchKernelGGL((fri_fold_kernel<F, 2, D>), grid, block, 0, ctx->stream, t, out, row_start, rc, lo, hi, io.log_lo, w16, cst, d_alpha); break;
        case 3: hipLaunchKernelGGL((fri_fold_kernel<F, 3, D>), grid, block, 0, ctx->stream, t, out, row_start, rc, lo, hi, io.log_lo, w16, cst, d_alpha); break;
        default: hipLaunchKernelGGL((fri_fold_kernel<F, 4, D>), grid, block, 0, ctx->stream, t, out, row_start, rc, lo, hi, io.log_lo, w16, cst, d_alpha); break;
    }
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

int check_args(uint32_t max_ext, uint32_t D, uint32_t log_len, uint32_t folding, uint32_t *log_nf) {
    if (D < 1 || D > max_ext) return WF_ERR_UNSUPPORTED;
    if (folding != 2 && folding != 4 && folding != 8 && folding != 16) return WF_ERR_UNSUPPORTED;  // mod.rs:187-195
    uint32_t l = 0;
    while ((1u << l) < folding) l++;
    if (log_len < l || log_len > 32) return WF_ERR_INVALID_ARG;
    *log_nf = l;
    return WF_OK;
}

template <class HF>
int layer_commit(wf_ctx *ctx, int hash, uint32_t D, const void *d_evals, uint32_t log_len, uint32_t folding, void *d_transposed,
                 void *d_leaves, void *d_nodes, void *h_root) {
    typedef typename HF::T T;
    uint32_t log_nf;
    WF_TRY(check_args(HF::Dev::MAX_EXT, D, log_len, folding, &log_nf));
    const uint32_t log_rc = log_len - log_nf;
    const uint64_t rc = 1ull << log_rc;
    const dim3 grid((uint32_t)((rc + 255) / 256)), block(256);
    const T *ev = (const T *)d_evals;
    T *tr = (T *)d_transposed;
    // transpose + leaf hashes in one pass where that is the faster arrangement (hash_kernels.hip)
    int fused = 0;
    WF_TRY(wf_fri_transpose_hash(ctx, hash, HF::Dev::ID, D, d_evals, log_rc, log_nf, d_transposed, d_leaves, &fused));
    if (fused) {
        WF_TRY(wf_merkle_build(ctx, hash, d_leaves, rc, d_nodes));
        if (h_root) WF_TRY(wf_memcpy_d2h(ctx, h_root, (const uint8_t *)d_nodes + 32, 32));
        return WF_OK;
    }
    wf_prof_begin(ctx, "fri_transpose");
    if (D == 1) hipLaunchKernelGGL((fri_transpose_kernel<T, 1>), grid, block, 0, ctx->stream, ev, tr, log_rc, log_nf);
    else if (D == 2) hipLaunchKernelGGL((fri_transpose_kernel<T, 2>), grid, block, 0, ctx->stream, ev, tr, log_rc, log_nf);
    else hipLaunchKernelGGL((fri_transpose_kernel<T, 3>), grid, block, 0, ctx->stream, ev, tr, log_rc, log_nf);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    const uint32_t row_elems = folding * D;
    // build_layer_commitment: leaf = hash_elements(row); V::new(leaves)
    WF_TRY(wf_hash_elements_batch(ctx, hash, HF::Dev::ID, d_transposed, rc, row_elems, row_elems, d_leaves));
    WF_TRY(wf_merkle_build(ctx, hash, d_leaves, rc, d_nodes));
    if (h_root) WF_TRY(wf_memcpy_d2h(ctx, h_root, (const uint8_t *)d_nodes + 32, 32));
    return WF_OK;
}

template <class HF>
int apply_drp(wf_ctx *ctx, uint32_t D, const void *d_transposed, uint32_t log_len, uint32_t folding, uint64_t row_start,
              uint64_t num_rows, const void *h_domain_offset, const void *h_alpha, const void *d_alpha, void *d_folded) {
    typedef typename HF::T T;
    uint32_t log_nf;
    WF_TRY(check_args(HF::Dev::MAX_EXT, D, log_len, folding, &log_nf));
    const uint32_t log_rc = log_len - log_nf;
    if (row_start + num_rows > (1ull << log_rc) || row_start + num_rows < row_start) return WF_ERR_INVALID_ARG;
    if (num_rows == 0) return WF_OK;
    T off;
    WF_TRY(wf_load_offset<HF>(h_domain_offset, &off));
    // inv_offsets[i] = offset^-1 * (g^-1)^i, g = root of unity of the layer's domain (folding/mod.rs:181-188)
    SeriesTable io;
    const T g_inv = HF::invmod(HF::root_of_unity(log_len));
    WF_TRY(wf_get_series_table<HF>(ctx, g_inv, HF::invmod(off), log_rc, &io));
    FoldConsts<T> cst;
    cst.inv_n = HF::to_internal(HF::invmod(HF::from_u64(folding)));
    for (uint32_t d = 0; d < 3; d++) cst.alpha[d] = 0;
    for (uint32_t d = 0; d < D && h_alpha; d++) {
        memcpy(&cst.alpha[d], (const uint8_t *)h_alpha + d * sizeof(T), sizeof(T));
        if (!HF::valid_internal(cst.alpha[d])) return WF_ERR_INVALID_ARG;
    }
    if (D == 1) return launch_fold<HF, 1>(ctx, log_nf, d_transposed, d_folded, row_start, num_rows, io, cst, d_alpha);
    if (D == 2) return launch_fold<HF, 2>(ctx, log_nf, d_transposed, d_folded, row_start, num_rows, io, cst, d_alpha);
    if constexpr (HF::Dev::MAX_EXT >= 3) return launch_fold<HF, 3>(ctx, log_nf, d_transposed, d_folded, row_start, num_rows, io, cst, d_alpha);
    return WF_ERR_UNSUPPORTED;
}

// FriProver::build_layers' loop (mod.rs:179-199) with the channel's coin on the device: nothing in here waits for the stream
template <class HF>
int build_layers(wf_ctx *ctx, int hash, uint32_t D, const void *d_evals, uint32_t log_len, uint32_t folding, uint32_t num_layers,
                 const void *h_domain_offset, void *d_coin, void *const *d_transposed, void *const *d_leaves, void *const *d_nodes,
                 void *const *d_folded, void *d_roots, void *d_alphas) {
    typedef typename HF::T T;
    uint32_t log_nf;
    WF_TRY(check_args(HF::Dev::MAX_EXT, D, log_len, folding, &log_nf));
    if ((uint64_t)num_layers * log_nf > log_len) return WF_ERR_INVALID_ARG;
    const void *ev = d_evals;
    for (uint32_t k = 0; k < num_layers; k++) {
        if (!d_transposed[k] || !d_leaves[k] || !d_nodes[k] || !d_folded[k]) return WF_ERR_INVALID_ARG;
        WF_TRY(layer_commit<HF>(ctx, hash, D, ev, log_len, folding, d_transposed[k], d_leaves[k], d_nodes[k], nullptr));
        void *alpha = (uint8_t *)d_alphas + (size_t)k * D * sizeof(T);
        // channel.commit_fri_layer(root) and channel.draw_fri_alpha(), one launch
        WF_TRY(wf_coin_reseed_draw(ctx, hash, HF::Dev::ID, D, d_coin, (const uint8_t *)d_nodes[k] + 32, (uint8_t *)d_roots + (size_t)k * 32, alpha));
        WF_TRY(apply_drp<HF>(ctx, D, d_transposed[k], log_len, folding, 0, 1ull << (log_len - log_nf), h_domain_offset, nullptr, alpha, d_folded[k]));
        ev = d_folded[k];
        log_len -= log_nf;
    }
    return WF_OK;
}

}  // namespace

extern "C" int wf_fri_build_layers(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_evals, uint32_t log_len,
                                   uint32_t folding, uint32_t num_layers, const void *h_domain_offset, void *d_coin,
                                   void *const *d_transposed, void *const *d_leaves, void *const *d_nodes, void *const *d_folded,
                                   void *d_roots, void *d_alphas) {
    if (!ctx || !d_evals || !h_domain_offset || !d_coin || !d_roots || !d_alphas) return WF_ERR_INVALID_ARG;
    if (num_layers == 0) return WF_OK;
    if (!d_transposed || !d_leaves || !d_nodes || !d_folded) return WF_ERR_INVALID_ARG;
    switch (field) {
        case WF_FIELD_F64: return build_layers<HostF64>(ctx, hash, ext_degree, d_evals, log_len, folding, num_layers, h_domain_offset, d_coin, d_transposed, d_leaves, d_nodes, d_folded, d_roots, d_alphas);
        case WF_FIELD_F128: return build_layers<HostF128>(ctx, hash, ext_degree, d_evals, log_len, folding, num_layers, h_domain_offset, d_coin, d_transposed, d_leaves, d_nodes, d_folded, d_roots, d_alphas);
        case WF_FIELD_F62: return build_layers<HostF62>(ctx, hash, ext_degree, d_evals, log_len, folding, num_layers, h_domain_offset, d_coin, d_transposed, d_leaves, d_nodes, d_folded, d_roots, d_alphas);
        default: return WF_ERR_UNSUPPORTED;
    }
}

extern "C" int wf_fri_layer_commit(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_evals,
                                   uint32_t log_len, uint32_t folding, void *d_transposed, void *d_leaves, void *d_nodes,
                                   void *h_root) {
    if (!ctx || !d_evals || !d_transposed || !d_leaves || !d_nodes) return WF_ERR_INVALID_ARG;
    switch (field) {
        case WF_FIELD_F64: return layer_commit<HostF64>(ctx, hash, ext_degree, d_evals, log_len, folding, d_transposed, d_leaves, d_nodes, h_root);
        case WF_FIELD_F128: return layer_commit<HostF128>(ctx, hash, ext_degree, d_evals, log_len, folding, d_transposed, d_leaves, d_nodes, h_root);
        case WF_FIELD_F62: return layer_commit<HostF62>(ctx, hash, ext_degree, d_evals, log_len, folding, d_transposed, d_leaves, d_nodes, h_root);
        default: return WF_ERR_UNSUPPORTED;
    }
}

extern "C" int wf_fri_apply_drp_rows(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_transposed_rows, uint32_t log_len,
                                     uint32_t folding, uint64_t row_start, uint64_t num_rows, const void *h_domain_offset,
                                     const void *h_alpha, void *d_folded) {
    if (!ctx || !d_transposed_rows || !h_domain_offset || !h_alpha || !d_folded) return WF_ERR_INVALID_ARG;
    switch (field) {
        case WF_FIELD_F64: return apply_drp<HostF64>(ctx, ext_degree, d_transposed_rows, log_len, folding, row_start, num_rows, h_domain_offset, h_alpha, nullptr, d_folded);
        case WF_FIELD_F128: return apply_drp<HostF128>(ctx, ext_degree, d_transposed_rows, log_len, folding, row_start, num_rows, h_domain_offset, h_alpha, nullptr, d_folded);
        case WF_FIELD_F62: return apply_drp<HostF62>(ctx, ext_degree, d_transposed_rows, log_len, folding, row_start, num_rows, h_domain_offset, h_alpha, nullptr, d_folded);
        default: return WF_ERR_UNSUPPORTED;
    }
}

extern "C" int wf_fri_apply_drp(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_transposed, uint32_t log_len,
                                uint32_t folding, const void *h_domain_offset, const void *h_alpha, void *d_folded) {
    uint32_t log_nf = 0;
    while ((1u << log_nf) < folding && log_nf < 5) log_nf++;
    if (log_len < log_nf) return WF_ERR_INVALID_ARG;
    return wf_fri_apply_drp_rows(ctx, field, ext_degree, d_transposed, log_len, folding, 0, 1ull << (log_len - log_nf), h_domain_offset,
                                 h_alpha, d_folded);
}
