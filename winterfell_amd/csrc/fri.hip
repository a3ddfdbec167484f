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

#include "fri_fold.cuh"
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

// set_remainder (mod.rs:230-239): the first `size` coefficients in reverse order
template <class T>
__global__ __launch_bounds__(256) void reverse_prefix_kernel(const T *src, T *dst, uint32_t size, uint32_t ew) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size * ew) return;
    const uint32_t e = i / ew, w = i - e * ew;
    dst[i] = src[(size - 1 - e) * ew + w];
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
    const T io = series_at<F>(io_lo, io_hi, io_log_lo, row_start + i);   // offset^-1 * g^-(global row)
    T al[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; d++) al[d] = d_alpha ? d_alpha[d] : cst.alpha[d];   // alpha from the host, or where the device coin drew it
    fri_fold_row<F, LOG_NF, D>(comp, io, cst.inv_n, al, w16, acc);
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

// where the layer's coin step goes when the tree's last launch can take it along (wf_merkle_build_coin): nullptr = plain tree
struct CoinStep {
    void *coin, *root_out, *alpha_out;
    int done;
};
static int tree_maybe_coin(wf_ctx *ctx, int hash, int field, uint32_t D, const void *d_leaves, uint64_t rows, void *d_nodes, CoinStep *cs) {
#ifndef WF_NO_MERKLE_COIN
    if (cs) return wf_merkle_build_coin(ctx, hash, d_leaves, rows, d_nodes, field, D, cs->coin, cs->root_out, cs->alpha_out, &cs->done);
#endif
    return wf_merkle_build(ctx, hash, d_leaves, rows, d_nodes);
}

template <class HF>
int layer_commit(wf_ctx *ctx, int hash, uint32_t D, const void *d_evals, uint32_t log_len, uint32_t folding, void *d_transposed,
                 void *d_leaves, void *d_nodes, void *h_root, CoinStep *cs = nullptr) {
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
        WF_TRY(tree_maybe_coin(ctx, hash, HF::Dev::ID, D, d_leaves, rc, d_nodes, cs));
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
    WF_TRY(tree_maybe_coin(ctx, hash, HF::Dev::ID, D, d_leaves, rc, d_nodes, cs));
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

// layer k's fold and layer k + 1's transpose + leaf hashes in one launch (fri_rows.hip: f64, BLAKE3 family, rows <= 128 bytes);
// *done = 0 when the shape has no fused kernel
template <class HF>
int fold_commit(wf_ctx *ctx, int hash, uint32_t D, const void *d_transposed, uint32_t log_len, uint32_t log_nf, const void *h_domain_offset,
                const void *d_alpha, void *d_folded, void *d_transposed_next, void *d_leaves_next, int *done) {
    typedef typename HF::T T;
    *done = 0;
    if constexpr (sizeof(T) != 8) {
        return WF_OK;
    } else {
        if (HF::Dev::ID != WF_FIELD_F64) return WF_OK;
        const uint32_t log_rc = log_len - log_nf;
        if (log_rc < log_nf) return WF_OK;
        T off;
        WF_TRY(wf_load_offset<HF>(h_domain_offset, &off));
        SeriesTable io;
        const T g_inv = HF::invmod(HF::root_of_unity(log_len));
        WF_TRY(wf_get_series_table<HF>(ctx, g_inv, HF::invmod(off), log_rc, &io));
        void *w256, *w16;
        WF_TRY(wf_get_small_tables<HF>(ctx, &w256, &w16));
        const T inv_n = HF::to_internal(HF::invmod(HF::from_u64(1ull << log_nf)));
        const T g_step = HF::to_internal(HF::powmod(g_inv, 1ull << (log_rc - log_nf)));      // rows i2 + j * rc2: g^-(rc2) per step of j
        return wf_fri_fold_commit(ctx, hash, WF_FIELD_F64, D, log_nf, d_transposed, 1ull << log_rc, io.d_lo, io.d_hi, io.log_lo, w16, (uint64_t)inv_n, d_alpha,
                                  (uint64_t)g_step, d_folded, d_transposed_next, d_leaves_next, done);
    }
}

// FriProver::build_layers' loop (mod.rs:179-199) with the channel's coin on the device: nothing in here waits for the stream.
//   commit layer 0;  for every layer k: reseed with its root, draw alpha_k, fold — together with the first half of layer k + 1's
//   commit where a fused kernel exists — and build layer k + 1's tree
template <class HF>
int build_layers(wf_ctx *ctx, int hash, uint32_t D, const void *d_evals, uint32_t log_len, uint32_t folding, uint32_t num_layers,
                 const void *h_domain_offset, void *d_coin, void *const *d_transposed, void *const *d_leaves, void *const *d_nodes,
                 void *const *d_folded, void *d_roots, void *d_alphas, uint32_t blowup, void *d_remainder) {
    typedef typename HF::T T;
    uint32_t log_nf;
    WF_TRY(check_args(HF::Dev::MAX_EXT, D, log_len, folding, &log_nf));
    if ((uint64_t)num_layers * log_nf > log_len) return WF_ERR_INVALID_ARG;
    if (d_remainder && (blowup == 0 || (blowup & (blowup - 1)) || ((uint64_t)blowup >> (log_len - num_layers * log_nf)) > 1)) return WF_ERR_INVALID_ARG;
    for (uint32_t k = 0; k < num_layers; k++)
        if (!d_transposed[k] || !d_leaves[k] || !d_nodes[k] || !d_folded[k]) return WF_ERR_INVALID_ARG;
    // The layers from the first one with at most 1024 rows on, and the remainder, run as ONE launch where a fused kernel exists
    // (wf_fri_tail, fri_rows.hip: f64, BLAKE3 family): k0 = the first such layer, num_layers = none
    uint32_t k0 = num_layers;
    bool tail_remainder = false;
#ifndef WF_NO_FRI_TAIL
    {
        uint32_t k = 0;
        while (k < num_layers && log_len - (k + 1) * log_nf > 10) k++;        // the first layer of at most 2^10 rows
        // wf_fri_tail_ok is the tail's own admission test: k0 is only moved when the launch WILL take the layers
        if (k < num_layers && wf_fri_tail_ok(hash, HF::Dev::ID, D, log_nf, log_len - k * log_nf, num_layers - k)) k0 = k;
    }
#endif
    const uint32_t log_len0 = log_len;
    // channel.commit_fri_layer(root_k) and alpha_k = channel.draw_fri_alpha() ride on the last launch of layer k's tree where a
    // fused kernel exists (wf_merkle_build_coin); cs.done says whether they did
    auto coin_step = [&](uint32_t k) -> CoinStep {
        return CoinStep{d_coin, (uint8_t *)d_roots + (size_t)k * 32, (uint8_t *)d_alphas + (size_t)k * D * sizeof(T), 0};
    };
    CoinStep cs = coin_step(0);
    if (num_layers && k0 > 0) WF_TRY(layer_commit<HF>(ctx, hash, D, d_evals, log_len, folding, d_transposed[0], d_leaves[0], d_nodes[0], nullptr, &cs));
    for (uint32_t k = 0; k < k0; k++) {
        void *alpha = (uint8_t *)d_alphas + (size_t)k * D * sizeof(T);
        if (!cs.done)
            WF_TRY(wf_coin_reseed_draw(ctx, hash, HF::Dev::ID, D, d_coin, (const uint8_t *)d_nodes[k] + 32, (uint8_t *)d_roots + (size_t)k * 32, alpha));
        cs = coin_step(k + 1);
        int fused = 0;
        if (k + 1 < k0)
            WF_TRY(fold_commit<HF>(ctx, hash, D, d_transposed[k], log_len, log_nf, h_domain_offset, alpha, d_folded[k], d_transposed[k + 1], d_leaves[k + 1],
                                   &fused));
        if (fused) {
            WF_TRY(tree_maybe_coin(ctx, hash, HF::Dev::ID, D, d_leaves[k + 1], 1ull << (log_len - 2 * log_nf), d_nodes[k + 1], &cs));
        } else {
            WF_TRY(apply_drp<HF>(ctx, D, d_transposed[k], log_len, folding, 0, 1ull << (log_len - log_nf), h_domain_offset, nullptr, alpha, d_folded[k]));
            if (k + 1 < k0)
                WF_TRY(layer_commit<HF>(ctx, hash, D, d_folded[k], log_len - log_nf, folding, d_transposed[k + 1], d_leaves[k + 1], d_nodes[k + 1], nullptr,
                                        &cs));
        }
        log_len -= log_nf;
    }
    if (k0 < num_layers) {
        if constexpr (sizeof(T) == 8) {
            // the tail: layers k0 .. num_layers - 1 (+ the remainder when its partial sums are small enough for one workgroup)
            const uint32_t nt = num_layers - k0;
            T off;
            WF_TRY(wf_load_offset<HF>(h_domain_offset, &off));
            SeriesTable io;
            const T g_inv = HF::invmod(HF::root_of_unity(log_len));
            WF_TRY(wf_get_series_table<HF>(ctx, g_inv, HF::invmod(off), log_len - log_nf, &io));
            void *w256, *w16;
            WF_TRY(wf_get_small_tables<HF>(ctx, &w256, &w16));
            const T inv_n = HF::to_internal(HF::invmod(HF::from_u64(1ull << log_nf)));
            const uint32_t log_rem = log_len - nt * log_nf;
            const uint64_t rem_n = 1ull << log_rem;
            uint32_t rem_size = 0;
            T w_inv = 0, off_inv = 0, n_inv = 0;
            if (d_remainder && blowup && rem_n / blowup >= 1) {
                rem_size = (uint32_t)(rem_n / blowup);
                w_inv = log_rem ? HF::to_internal(HF::invmod(HF::root_of_unity(log_rem))) : HF::to_internal(HF::from_u64(1));
                off_inv = HF::to_internal(HF::invmod(off));
                n_inv = HF::to_internal(HF::invmod(HF::from_u64(rem_n)));
            }
            // in the tail: the remainder's hash is one chunk there (<= 1024 bytes); rem_n <= 1024 always holds (the tail's layers have <= 1024 rows)
            const bool rem_in_tail = rem_size != 0 && wf_fri_tail_rem_ok(D, log_rem, rem_size);
            int done = 0;
            WF_TRY(wf_fri_tail(ctx, hash, HF::Dev::ID, D, log_nf, k0 ? d_folded[k0 - 1] : d_evals, log_len, nt, d_transposed + k0, d_leaves + k0, d_nodes + k0,
                               d_folded + k0, (uint8_t *)d_roots + (size_t)k0 * 32, (uint8_t *)d_alphas + (size_t)k0 * D * sizeof(T), d_coin, io.d_lo, io.d_hi,
                               io.log_lo, w16, (uint64_t)inv_n, rem_in_tail ? d_remainder : nullptr, rem_size, (uint64_t)w_inv, (uint64_t)off_inv,
                               (uint64_t)n_inv, &done));
            if (!done) return WF_ERR_UNSUPPORTED;      // cannot happen: k0 and rem_in_tail come from the tail's own predicates
            log_len -= nt * log_nf;
            tail_remainder = rem_in_tail;
        }
    }
    (void)log_len0;
    if (tail_remainder) return WF_OK;
    if (d_remainder) {
        // set_remainder (mod.rs:230-239) on the stream as well: interpolate the last evaluations over the coset (in place), keep
        // len / blowup coefficients in reverse order, hash them, and the commitment goes into the coin like a layer root
        // with no layers the evaluations are the CALLER's (const in the ABI): interpolate a private copy (at most blowup * (rem_deg + 1)
        // elements), not in place
        void *ev = num_layers ? d_folded[num_layers - 1] : nullptr;
        void *copy = nullptr;
        if (!ev) {
            const size_t bytes = ((size_t)D * sizeof(T)) << log_len;
            WF_TRY(wf_malloc(ctx, bytes, &copy));
            WF_HIP(hipMemcpyAsync(copy, d_evals, bytes, hipMemcpyDeviceToDevice, ctx->stream));
            ev = copy;
        }
        struct FreeCopy {
            wf_ctx *c;
            void *p;
            ~FreeCopy() { if (p) (void)wf_free(c, p); }      // stream-ordered pool: safe right after the last launch that reads it
        } free_copy{ctx, copy};
        if (log_len > 0) WF_TRY(wf_fft_interpolate_poly_with_offset(ctx, HF::Dev::ID, D, ev, log_len, h_domain_offset));
        const uint32_t size = (uint32_t)((1ull << log_len) / blowup), ew = D;
        if (size == 0) return WF_ERR_INVALID_ARG;
        hipLaunchKernelGGL(reverse_prefix_kernel<T>, dim3((size * ew + 255) / 256), dim3(256), 0, ctx->stream, (const T *)ev, (T *)d_remainder, size, ew);
        WF_HIP(hipGetLastError());
        void *com = (uint8_t *)d_roots + (size_t)num_layers * 32;
        WF_TRY(wf_hash_elements_batch(ctx, hash, HF::Dev::ID, d_remainder, 1, (uint64_t)size * D, size * D, com));
        WF_TRY(wf_coin_reseed(ctx, hash, d_coin, com, nullptr));
    }
    return WF_OK;
}

}  // namespace

extern "C" int wf_fri_build_layers(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_evals, uint32_t log_len,
                                   uint32_t folding, uint32_t num_layers, const void *h_domain_offset, void *d_coin,
                                   void *const *d_transposed, void *const *d_leaves, void *const *d_nodes, void *const *d_folded,
                                   void *d_roots, void *d_alphas, uint32_t blowup, void *d_remainder) {
    WF_ENTER(ctx);
    if (!ctx || !d_evals || !h_domain_offset || !d_coin || !d_roots || (num_layers && !d_alphas)) return WF_ERR_INVALID_ARG;
    if (num_layers == 0 && !d_remainder) return WF_OK;
    if (num_layers && (!d_transposed || !d_leaves || !d_nodes || !d_folded)) return WF_ERR_INVALID_ARG;
#define WF_BL(HF) return build_layers<HF>(ctx, hash, ext_degree, d_evals, log_len, folding, num_layers, h_domain_offset, d_coin, d_transposed, d_leaves, \
                                          d_nodes, d_folded, d_roots, d_alphas, blowup, d_remainder)
    switch (field) {
        case WF_FIELD_F64: WF_BL(HostF64);
        case WF_FIELD_F128: WF_BL(HostF128);
        case WF_FIELD_F62: WF_BL(HostF62);
        default: return WF_ERR_UNSUPPORTED;
    }
#undef WF_BL
}

extern "C" int wf_fri_layer_commit(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_evals,
                                   uint32_t log_len, uint32_t folding, void *d_transposed, void *d_leaves, void *d_nodes,
                                   void *h_root) {
    WF_ENTER(ctx);
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
    WF_ENTER(ctx);
    if (!ctx || !d_transposed_rows || !h_domain_offset || !h_alpha || !d_folded) return WF_ERR_INVALID_ARG;
    switch (field) {
        case WF_FIELD_F64: return apply_drp<HostF64>(ctx, ext_degree, d_transposed_rows, log_len, folding, row_start, num_rows, h_domain_offset, h_alpha, nullptr, d_folded);
        case WF_FIELD_F128: return apply_drp<HostF128>(ctx, ext_degree, d_transposed_rows, log_len, folding, row_start, num_rows, h_domain_offset, h_alpha, nullptr, d_folded);
        case WF_FIELD_F62: return apply_drp<HostF62>(ctx, ext_degree, d_transposed_rows, log_len, folding, row_start, num_rows, h_domain_offset, h_alpha, nullptr, d_folded);
        default: return WF_ERR_UNSUPPORTED;
    }
}

// the same with alpha where the device coin drew it (wf_coin_draw / wf_coin_reseed_draw): nothing crosses to the host
extern "C" int wf_fri_apply_drp_rows_dev(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_transposed_rows, uint32_t log_len,
                                         uint32_t folding, uint64_t row_start, uint64_t num_rows, const void *h_domain_offset,
                                         const void *d_alpha, void *d_folded) {
    WF_ENTER(ctx);
    if (!ctx || !d_transposed_rows || !h_domain_offset || !d_alpha || !d_folded) return WF_ERR_INVALID_ARG;
    switch (field) {
        case WF_FIELD_F64: return apply_drp<HostF64>(ctx, ext_degree, d_transposed_rows, log_len, folding, row_start, num_rows, h_domain_offset, nullptr, d_alpha, d_folded);
        case WF_FIELD_F128: return apply_drp<HostF128>(ctx, ext_degree, d_transposed_rows, log_len, folding, row_start, num_rows, h_domain_offset, nullptr, d_alpha, d_folded);
        case WF_FIELD_F62: return apply_drp<HostF62>(ctx, ext_degree, d_transposed_rows, log_len, folding, row_start, num_rows, h_domain_offset, nullptr, d_alpha, d_folded);
        default: return WF_ERR_UNSUPPORTED;
    }
}

extern "C" int wf_fri_apply_drp(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_transposed, uint32_t log_len,
                                uint32_t folding, const void *h_domain_offset, const void *h_alpha, void *d_folded) {
    WF_ENTER(ctx);
    uint32_t log_nf = 0;
    while ((1u << log_nf) < folding && log_nf < 5) log_nf++;
    if (log_len < log_nf) return WF_ERR_INVALID_ARG;
    return wf_fri_apply_drp_rows(ctx, field, ext_degree, d_transposed, log_len, folding, 0, 1ull << (log_len - log_nf), h_domain_offset,
                                 h_alpha, d_folded);
}
