// FRI layer commit, first half: transpose_slice + hash of every row in one pass (fri/src/prover/mod.rs:321-336).
#include "hashers.cuh"
#include "coin_state.cuh"
#include "fri_fold.cuh"
#include "merkle_stage.cuh"
#include "tables.cuh"

namespace {

// FRI layer commit, first half, in one pass (fri/src/prover/mod.rs:321-336 = transpose_slice + hash each row):
//   tr[i][j] = ev[i + j * rc]   and   leaf_i = H::hash_elements(tr[i]).
// A workgroup takes R consecutive rows: the N strided runs of R elements are read coalesced into an LDS tile
// [R][row_words + 1], every lane hashes its row straight out of the tile, and the tile is written to the transposed matrix
// as ONE contiguous block.  (The separate transpose wrote 8 or 16 bytes per lane at a row-sized stride — 1 TB/s — and the
// row hash then read the matrix back.)  EW = 64-bit words per element (ext_degree * words per base element).
template <class H, int MODE>
__global__ __launch_bounds__(256) void fri_rows_kernel(const uint64_t *ev, uint64_t rc, uint32_t N, uint32_t EW, uint32_t R, uint64_t *tr,
                                                       void *leaves) {
    extern __shared__ uint64_t fri_tile[];
    const uint32_t row_words = N * EW, pitch = row_words + 1;
    const uint64_t r0 = (uint64_t)blockIdx.x * R;
    const uint32_t nr = rc - r0 < R ? (uint32_t)(rc - r0) : R;
    const uint32_t run = nr * EW;                        // consecutive words of one strided run
    for (uint32_t idx = threadIdx.x; idx < N * run; idx += 256) {
        const uint32_t j = idx / run, k = idx - j * run;
        fri_tile[(k / EW) * pitch + j * EW + (k % EW)] = ev[(r0 + (uint64_t)j * rc) * EW + k];
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < nr; t += 256) {
        uint32_t d[8];
        H::template hash_elems<MODE, false>(fri_tile + t * pitch, row_words, d);
        store_digest(leaves, r0 + t, d);
    }
    for (uint32_t idx = threadIdx.x; idx < nr * row_words; idx += 256) {
        const uint32_t r = idx / row_words, w = idx - r * row_words;
        tr[r0 * row_words + idx] = fri_tile[r * pitch + w];
    }
}

// The same for rows of at most 128 bytes, the shapes the folding factors 2..16 give over f64 / f128 and their extensions:
// every lane owns one row.  Its N elements are N loads that are coalesced across the lanes as they stand (lane-consecutive
// rows are consecutive in each strided run), the row is hashed out of registers, and only the transposed copy goes through LDS —
// per wavefront, 64 rows = one contiguous block of the output, written back 16 bytes per lane in address order.  Against the
// tile kernel above: no index arithmetic with run-time divisors (it was ~40 % of the instructions), 16-byte accesses.
template <class H, int MODE, int N, int EW>
__global__ __launch_bounds__(256) void fri_rows_direct_kernel(const uint64_t *ev, uint64_t rc, uint64_t *tr, void *leaves) {
    constexpr int RW = N * EW, CP = RW / 2;                      // 64-bit words / 16-byte chunks per row
    constexpr bool POW2 = (CP & (CP - 1)) == 0;
    constexpr int SPREAD = POW2 && CP < 16 ? 16 / CP : 1;        // rows that share a swizzle value: 16 lanes then touch 16 banks-of-16-bytes
    __shared__ uint4 stage[4][64 * CP];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_row0 = (uint64_t)blockIdx.x * 256 + wave * 64;
    const uint64_t r = wave_row0 + lane;
    auto swz = [&](uint32_t row) -> uint32_t { return POW2 ? ((row / SPREAD) & (CP - 1)) : 0u; };
    uint64_t w[RW];
#pragma unroll
    for (int i = 0; i < RW; i++) w[i] = 0;
    if (r < rc) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            const uint64_t *e = ev + (r + (uint64_t)j * rc) * EW;
            if constexpr (EW % 2 == 0) {
#pragma unroll
                for (int k = 0; k < EW / 2; k++) {
                    const uint4 v = reinterpret_cast<const uint4 *>(e)[k];
                    w[j * EW + 2 * k] = (uint64_t)v.x | ((uint64_t)v.y << 32);
                    w[j * EW + 2 * k + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
                }
            } else {
#pragma unroll
                for (int k = 0; k < EW; k++) w[j * EW + k] = e[k];
            }
        }
        uint32_t d[8];
        H::template hash_elems<MODE, false>(w, RW, d);
        store_digest(leaves, r, d);
    }
    uint4 *st = stage[wave];
#pragma unroll
    for (int c = 0; c < CP; c++)
        st[lane * CP + (c ^ swz(lane))] = make_uint4((uint32_t)w[2 * c], (uint32_t)(w[2 * c] >> 32), (uint32_t)w[2 * c + 1], (uint32_t)(w[2 * c + 1] >> 32));
    __syncthreads();
    if (wave_row0 >= rc) return;
    const uint32_t nv = rc - wave_row0 < 64 ? (uint32_t)(rc - wave_row0) : 64u;
    uint4 *out = reinterpret_cast<uint4 *>(tr + wave_row0 * RW);
#pragma unroll
    for (int i = 0; i < CP; i++) {
        const uint32_t L = i * 64 + lane, row = L / CP, cc = L % CP;
        if (row < nv) out[L] = st[row * CP + (cc ^ swz(row))];
    }
}

template <class H, int MODE, int N, int EW>
void launch_fri_rows_direct(wf_ctx *ctx, const uint64_t *ev, uint64_t rc, uint64_t *tr, void *leaves) {
    hipLaunchKernelGGL((fri_rows_direct_kernel<H, MODE, N, EW>), dim3((uint32_t)((rc + 255) / 256)), dim3(256), 0, ctx->stream, ev, rc, tr, leaves);
}

// the (mode, N, EW) shapes with a direct kernel; false = take the tile kernel
template <class H>
bool try_fri_rows_direct(wf_ctx *ctx, int mode, const uint64_t *ev, uint64_t rc, uint32_t N, uint32_t EW, uint64_t *tr, void *leaves) {
    if constexpr (!H::WAVE_TREE) {
        return false;     // the BLAKE3 family only: for the others the hash, not the data movement, is the time
    } else {
        if ((rc + 255) / 256 > 0x7fffffffull) return false;
#define WF_FD(M, NN, E) if (mode == M && N == NN && EW == E) { launch_fri_rows_direct<H, M, NN, E>(ctx, ev, rc, tr, leaves); return true; }
        WF_FD(MODE_F64_CANON, 2, 1) WF_FD(MODE_F64_CANON, 4, 1) WF_FD(MODE_F64_CANON, 8, 1) WF_FD(MODE_F64_CANON, 16, 1)
        WF_FD(MODE_F64_CANON, 2, 2) WF_FD(MODE_F64_CANON, 4, 2) WF_FD(MODE_F64_CANON, 8, 2)
        WF_FD(MODE_F64_CANON, 2, 3) WF_FD(MODE_F64_CANON, 4, 3)
        WF_FD(MODE_RAW, 2, 2) WF_FD(MODE_RAW, 4, 2) WF_FD(MODE_RAW, 8, 2) WF_FD(MODE_RAW, 2, 4) WF_FD(MODE_RAW, 4, 4)
#undef WF_FD
        return false;
    }
}

template <class H>
int launch_fri_rows(wf_ctx *ctx, int mode, const uint64_t *ev, uint64_t rc, uint32_t N, uint32_t EW, uint64_t *tr, void *leaves) {
    wf_prof_begin(ctx, "fri_transpose_hash");
    if (!try_fri_rows_direct<H>(ctx, mode, ev, rc, N, EW, tr, leaves)) {
        const uint32_t pitch = N * EW + 1;
        uint32_t R = 256;
        while (R > 32 && (size_t)R * pitch * 8 > 40960) R >>= 1;
        const uint64_t blocks = (rc + R - 1) / R;
        if (blocks > 0x7fffffffull) {
            wf_prof_end(ctx);
            return WF_ERR_DOMAIN_TOO_LARGE;
        }
        const size_t lds = (size_t)R * pitch * 8;
#define WF_FR(MODE) hipLaunchKernelGGL((fri_rows_kernel<H, MODE>), dim3((uint32_t)blocks), dim3(256), lds, ctx->stream, ev, rc, N, EW, R, tr, leaves)
        switch (mode) {
            case MODE_F64_CANON: WF_FR(MODE_F64_CANON); break;
            case MODE_F62_CANON: WF_FR(MODE_F62_CANON); break;
            default: WF_FR(MODE_RAW); break;
        }
#undef WF_FR
    }
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

// apply_drp of layer k fused with the first half of layer k + 1's commit (f64, BLAKE3 family, rows of <= 128 bytes): lane i2 owns
// row i2 of the NEXT layer, i.e. the folded values of the current layer's rows i2 + j * rc2 (j < N).  It reads those N rows
// (64-byte runs, contiguous across the lanes), folds each at alpha (fri_fold_row), writes the N folded values to their places in the
// natural-order vector (the layer's evaluations: the API hands them out, the remainder needs the last one), hashes the new row out of
// registers and writes the transposed copy through the per-wavefront LDS stage of fri_rows_direct_kernel.  Against fold, then
// transpose + hash: one launch less per layer, and the folded vector is not read back.
template <class H, int LOG_NF, int D>
__global__ __launch_bounds__(256) void fri_fold_commit_kernel(const uint64_t *t, uint64_t rc, const uint64_t *io_lo, const uint64_t *io_hi, uint32_t io_log_lo,
                                                              const uint64_t *w16, uint64_t inv_n, const uint64_t *d_alpha, uint64_t g_step,
                                                              uint64_t *folded, uint64_t *tr_next, void *leaves_next) {
    typedef F64 F;
    constexpr int N = 1 << LOG_NF, RW = N * D, CP = RW / 2;
    constexpr bool POW2 = (CP & (CP - 1)) == 0;
    constexpr int SPREAD = POW2 && CP < 16 ? 16 / CP : 1;
    __shared__ uint4 stage[4][64 * (CP > 0 ? CP : 1)];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t rc2 = rc >> LOG_NF;
    const uint64_t wave_row0 = (uint64_t)blockIdx.x * 256 + wave * 64;
    const uint64_t i2 = wave_row0 + lane;
    auto swz = [&](uint32_t row) -> uint32_t { return POW2 ? ((row / SPREAD) & (CP - 1)) : 0u; };
    uint64_t w[RW];
#pragma unroll
    for (int i = 0; i < RW; i++) w[i] = 0;
    if (i2 < rc2) {
        uint64_t al[D];
#pragma unroll
        for (int d = 0; d < D; d++) al[d] = d_alpha[d];
        uint64_t io = series_at<F>(io_lo, io_hi, io_log_lo, i2);          // offset^-1 * g^-i2; row i2 + j * rc2 has io * g_step^j
#pragma unroll
        for (int j = 0; j < N; j++) {
            const uint64_t r = i2 + (uint64_t)j * rc2;
            uint64_t comp[D][N];
            const uint64_t *row = t + r * RW;
#pragma unroll
            for (int e = 0; e < N; e++)
#pragma unroll
                for (int d = 0; d < D; d++) comp[d][e] = row[e * D + d];
            uint64_t acc[D];
            fri_fold_row<F, LOG_NF, D>(comp, io, inv_n, al, w16, acc);
#pragma unroll
            for (int d = 0; d < D; d++) {
                w[j * D + d] = acc[d];
                folded[r * D + d] = acc[d];
            }
            if (j + 1 < N) io = F::mul(io, g_step);
        }
        uint32_t dg[8];
        H::template hash_elems<MODE_F64_CANON, false>(w, RW, dg);
        store_digest(leaves_next, i2, dg);
    }
    if constexpr (CP == 1) {
        if (i2 < rc2) reinterpret_cast<uint4 *>(tr_next)[i2] = make_uint4((uint32_t)w[0], (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32));
    } else {
        uint4 *st = stage[wave];
#pragma unroll
        for (int c = 0; c < CP; c++)
            st[lane * CP + (c ^ swz(lane))] = make_uint4((uint32_t)w[2 * c], (uint32_t)(w[2 * c] >> 32), (uint32_t)w[2 * c + 1], (uint32_t)(w[2 * c + 1] >> 32));
        __syncthreads();
        if (wave_row0 >= rc2) return;
        const uint32_t nv = rc2 - wave_row0 < 64 ? (uint32_t)(rc2 - wave_row0) : 64u;
        uint4 *out = reinterpret_cast<uint4 *>(tr_next + wave_row0 * RW);
#pragma unroll
        for (int i = 0; i < CP; i++) {
            const uint32_t L = i * 64 + lane, rr = L / CP, cc = L % CP;
            if (rr < nv) out[L] = st[rr * CP + (cc ^ swz(rr))];
        }
    }
}

template <class H>
bool try_fri_fold_commit(wf_ctx *ctx, uint32_t D, uint32_t log_nf, const uint64_t *t, uint64_t rc, const uint64_t *io_lo, const uint64_t *io_hi,
                         uint32_t io_log_lo, const uint64_t *w16, uint64_t inv_n, const uint64_t *d_alpha, uint64_t g_step, uint64_t *folded,
                         uint64_t *tr_next, void *leaves_next) {
    if constexpr (!H::WAVE_TREE) {
        return false;
    } else {
        const uint64_t rc2 = rc >> log_nf;
        if (rc2 == 0 || (rc2 + 255) / 256 > 0x7fffffffull) return false;
        const dim3 grid((uint32_t)((rc2 + 255) / 256));
#define WF_FC(LN, DD) if (log_nf == LN && D == DD) { hipLaunchKernelGGL((fri_fold_commit_kernel<H, LN, DD>), grid, dim3(256), 0, ctx->stream, t, rc, io_lo, io_hi, io_log_lo, w16, inv_n, d_alpha, g_step, folded, tr_next, leaves_next); return true; }
        WF_FC(1, 1) WF_FC(1, 2) WF_FC(1, 3) WF_FC(2, 1) WF_FC(2, 2) WF_FC(2, 3) WF_FC(3, 1) WF_FC(3, 2) WF_FC(4, 1)
#undef WF_FC
        return false;
    }
}


// ---- the tail of the commit phase in ONE launch ---------------------------------------------------------------------------------
// From the first layer with at most FRI_TAIL_MAX_ROWS rows on, a layer is a chain of dependent small steps — transpose + leaf
// hashes, a tree of log2(rows) levels, reseed + draw, fold — and as separate launches each step paid a launch and its event
// bracket (round 2: ~42 us per small layer, ~60 us for the remainder's four launches; fri_fold_commit of ONE workgroup took 11.7 us).
// Here one 1024-thread workgroup walks all the remaining layers and the remainder (set_remainder, fri/src/prover/mod.rs:230-239)
// with barriers in between; everything it hands from step to step goes through global memory it wrote itself (L2) or LDS.
//   per layer (build_layer, mod.rs:202-222): rows + leaves -> tree (levels of > 512 merges global to global, then merkle_stage_wg)
//   -> coin.reseed(root), alpha = coin.draw() on lane 0 -> apply_drp of every row with offset^-1 g^-i taken from the FIRST tail
//   layer's series table at index i * N^m (g of layer m is g_0^(N^m); the reference uses the same offset at every layer, mod.rs:216).
//   remainder: the len / blowup low coefficients of the coset interpolation, c_k = (1/n) offset^-k sum_i e_i w^-ik — only those are
//   kept (reversed), so they are computed as plain sums over 1024 lanes (powers of w^-1 from an LDS table, groups of lanes added up
//   with shuffles) —, their hash (one chunk, chained on four lanes out of LDS) and coin.reseed with it.  Measured by leaving the kernel
//   early (-DFRI_TAIL_STOP, HIP events; the device clock read from inside proved misleading): launch 8, rows + leaves 4, tree 15,
//   coin 5, fold 4.5 us; the remainder was 57 us of the 93 when its sums ran on running products with global loads in the loop and its
//   hash on ONE lane through the general multi-chunk path (private memory).
#ifndef FRI_TAIL_MAX_ROWS
#define FRI_TAIL_MAX_ROWS 1024
#endif
// base^e in the f64 field, Montgomery form (one = 2^64 mod p)
__device__ __forceinline__ uint64_t pow_u64(uint64_t base, uint32_t e) {
    uint64_t r = 0xffffffffull;
    while (e) {
        if (e & 1u) r = gl::mul(r, base);
        base = gl::mul(base, base);
        e >>= 1;
    }
    return r;
}
constexpr int FRI_TAIL_MAX_LAYERS = 12;
struct FriTailParams {
    const uint64_t *ev;                 // natural-order evaluations feeding the first tail layer: (rows0 << LOG_NF) elements of D words
    uint32_t log_rows0, num_layers;
    uint64_t *tr[FRI_TAIL_MAX_LAYERS];
    void *leaves[FRI_TAIL_MAX_LAYERS];
    void *nodes[FRI_TAIL_MAX_LAYERS];
    uint64_t *folded[FRI_TAIL_MAX_LAYERS];
    uint32_t *roots;                    // (num_layers + 1) digests: this call's layer roots, then the remainder commitment
    uint64_t *alphas;                   // num_layers x D words
    CoinState *coin;
    const uint64_t *io_lo, *io_hi;      // offset^-1 * g^-i of the first tail layer
    uint32_t io_log_lo;
    const uint64_t *w16;
    uint64_t inv_n;                     // 1 / N
    uint64_t *remainder;                // nullptr: no remainder step
    uint32_t rem_size, log_rem_n;       // coefficients kept, log2 of the last vector's length
    uint64_t rem_w_inv, rem_off_inv, rem_n_inv;   // w_n^-1, offset^-1, 1/n (internal form)
};

#ifdef FRI_TAIL_STOP
#define TAIL_STEP(i) do { if (FRI_TAIL_STOP == (i)) return; } while (0)      // timing experiment: leave the kernel at step boundary i
#else
#define TAIL_STEP(i) do { } while (0)
#endif
template <class H, int LOG_NF, int D>
__global__ __launch_bounds__(1024) void fri_tail_kernel(FriTailParams p) {
    typedef F64 F;
    constexpr int N = 1 << LOG_NF, RW = N * D, T = 1024;
    __shared__ uint4 bufA[512 * 2];
    __shared__ uint4 bufB[256 * 2];
    __shared__ uint64_t s_alpha[4];
    __shared__ uint64_t rem_tw[1024];
    const int tid = threadIdx.x;
    const uint64_t *ev = p.ev;
    uint32_t rows = 1u << p.log_rows0, log_mult = 0;
    TAIL_STEP(0);
    for (uint32_t k = 0; k < p.num_layers; k++) {
        uint64_t *tr = p.tr[k];
        // ---- rows (transpose_slice) and leaves
        for (uint32_t i = tid; i < rows; i += T) {
            uint64_t w[RW];
#pragma unroll
            for (int j = 0; j < N; j++)
#pragma unroll
                for (int d = 0; d < D; d++) w[j * D + d] = ev[((uint64_t)i + (uint64_t)j * rows) * D + d];
#pragma unroll
            for (int c = 0; c < RW; c++) tr[(uint64_t)i * RW + c] = w[c];
            uint32_t dg[8];
            H::template hash_elems<MODE_F64_CANON, false>(w, RW, dg);
            store_digest(p.leaves[k], i, dg);
        }
        __syncthreads();
        TAIL_STEP(1);
        // ---- the tree
        {
            const void *in = p.leaves[k];
            uint32_t c = rows;
            while (c > 1024) {                       // levels of more than 512 merges: global to global
                const uint32_t cnt = c >> 1;
                for (uint32_t i = tid; i < cnt; i += T) {
                    uint32_t m[16], dgst[8];
                    load_pair(in, i, m);
                    H::merge(m, dgst);
                    store_digest(p.nodes[k], (uint64_t)cnt + i, dgst);
                }
                __syncthreads();
                in = reinterpret_cast<const uint8_t *>(p.nodes[k]) + (size_t)cnt * 32;
                c = cnt;
            }
            uint32_t lc = 0;
            while ((1u << lc) < c) lc++;
            merkle_stage_wg<H, T>(in, p.nodes[k], c, lc | 0x80000000u, 0, tid, bufA, bufB);
        }
        __syncthreads();
        TAIL_STEP(2);
        // ---- channel.commit_fri_layer(root), alpha = channel.draw_fri_alpha()
        {
            const uint32_t *root = reinterpret_cast<const uint32_t *>(p.nodes[k]) + 8;
            uint64_t *al = p.alphas + (uint64_t)k * D;
            if constexpr (H::QUAD_MERGE) {                       // Blake3_256: one compression across four lanes
                uint32_t *scratch = reinterpret_cast<uint32_t *>(bufA);
                coin_reseed_draw_quad_wg<WF_FIELD_F64, D>(p.coin, root, p.roots + 8 * k, al, tid, scratch, scratch + 16, reinterpret_cast<int *>(scratch + 24));
            } else {
                if (tid == 0) coin_reseed_draw_lane<H, WF_FIELD_F64, D>(p.coin, root, p.roots + 8 * k, al);
            }
            __syncthreads();
            if (tid == 0) {
#pragma unroll
                for (int d = 0; d < D; d++) s_alpha[d] = al[d];
            }
        }
        __syncthreads();
        TAIL_STEP(3);
        // ---- apply_drp
        {
            uint64_t al[D];
#pragma unroll
            for (int d = 0; d < D; d++) al[d] = s_alpha[d];
            uint64_t *fo = p.folded[k];
            for (uint32_t i = tid; i < rows; i += T) {
                uint64_t comp[D][N];
#pragma unroll
                for (int e = 0; e < N; e++)
#pragma unroll
                    for (int d = 0; d < D; d++) comp[d][e] = tr[((uint64_t)i * N + e) * D + d];
                const uint64_t io = series_at<F>(p.io_lo, p.io_hi, p.io_log_lo, (uint64_t)i << log_mult);
                uint64_t acc[D];
                fri_fold_row<F, LOG_NF, D>(comp, io, p.inv_n, al, p.w16, acc);
#pragma unroll
                for (int d = 0; d < D; d++) fo[(uint64_t)i * D + d] = acc[d];
            }
        }
        __syncthreads();
        TAIL_STEP(4);
        ev = p.folded[k];
        log_mult += LOG_NF;
        rows >>= LOG_NF;
    }
    if (p.remainder == nullptr) return;
    // ---- set_remainder: n = the length of the last vector (<= 1024), coefficient k < rem_size of its coset interpolation
    {
        const uint32_t n = 1u << p.log_rem_n, size = p.rem_size;
        uint32_t *msg = reinterpret_cast<uint32_t *>(bufA);          // the remainder as message words (<= 256), then 16 words for the reseed
        // w^-j for every j < n: one square-and-multiply per lane
        if ((uint32_t)tid < n) rem_tw[tid] = pow_u64(p.rem_w_inv, (uint32_t)tid);
        if (tid < 256 + 16) msg[tid] = 0;
        __syncthreads();
        TAIL_STEP(6);
        // work split: `parts` adjacent lanes per coefficient (a power of two <= 64, so a group never straddles a wavefront), each
        // summing n / parts terms e_i w^-(i k); the group adds up with lane shuffles
        uint32_t parts = 1;
        while (size * parts * 2 <= (uint32_t)T && parts * 2 <= n && parts < 64) parts *= 2;
        const uint32_t kk = (uint32_t)tid / parts, part = (uint32_t)tid % parts;
        uint64_t acc[D];
#pragma unroll
        for (int d = 0; d < D; d++) acc[d] = F::zero();
        if (kk < size) {
            const uint32_t per = n / parts, i0 = part * per;
#pragma unroll 4
            for (uint32_t i = i0; i < i0 + per; i++) {
                const uint64_t tw = rem_tw[(i * kk) & (n - 1)];
#pragma unroll
                for (int d = 0; d < D; d++) acc[d] = F::add(acc[d], F::mul(ev[(uint64_t)i * D + d], tw));
            }
        }
        TAIL_STEP(7);
        for (uint32_t o = parts >> 1; o >= 1; o >>= 1) {
#pragma unroll
            for (int d = 0; d < D; d++) acc[d] = F::add(acc[d], (uint64_t)__shfl_xor((unsigned long long)acc[d], (int)o));
        }
        TAIL_STEP(8);
        if (kk < size && part == 0) {
            const uint64_t sc = F::mul(p.rem_n_inv, pow_u64(p.rem_off_inv, kk));
#pragma unroll
            for (int d = 0; d < D; d++) {
                const uint64_t v = F::mul(acc[d], sc);
                const uint64_t at = (uint64_t)(size - 1 - kk) * D + d;                                            // reversed (mod.rs:236)
                p.remainder[at] = v;
                const uint64_t c = gl::to_int(v);                                                                 // what is hashed: as_int()
                msg[2 * at] = (uint32_t)c;
                msg[2 * at + 1] = (uint32_t)(c >> 32);
            }
        }
        __syncthreads();
        TAIL_STEP(5);
        // ---- commitment = hash_elements(remainder) (one chunk: size * D * 8 <= 1024 bytes), channel.commit_fri_layer(commitment)
        uint32_t *com = p.roots + 8 * p.num_layers;
        if constexpr (H::QUAD_MERGE) {                            // Blake3_256: the chain of <= 17 compressions on four lanes
            uint32_t *mm = msg + 256;
            if (tid < 4) {
                uint32_t lo, hi;
                b3::quad_hash_chunk((uint32_t)tid, msg, size * D * 8, lo, hi);
                com[tid] = lo;
                com[4 + tid] = hi;
                mm[tid] = p.coin->seed[tid];
                mm[4 + tid] = p.coin->seed[4 + tid];
                mm[8 + tid] = lo;
                mm[12 + tid] = hi;
            }
            __syncthreads();
            if (tid < 4) {
                uint32_t lo, hi;
                b3::quad_hash_block(b3::quad_init((uint32_t)tid, 64, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT), mm, lo, hi);
                p.coin->seed[tid] = lo;
                p.coin->seed[4 + tid] = hi;
                if (tid == 0) p.coin->counter = 0;
            }
        } else if (tid == 0) {
            uint32_t dg[8], m[16], sd[8];
            H::template hash_elems<MODE_F64_CANON, false>(p.remainder, size * D, dg);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                com[i] = dg[i];
                m[i] = p.coin->seed[i];
                m[8 + i] = dg[i];
            }
            H::merge(m, sd);                         // channel.commit_fri_layer(commitment): coin.reseed
#pragma unroll
            for (int i = 0; i < 8; i++) p.coin->seed[i] = sd[i];
            p.coin->counter = 0;
        }
    }
}

// the (log2 folding factor, extension degree) pairs fri_tail_kernel is instantiated for: the WF_FT list below and nothing else
constexpr bool fri_tail_has_kernel(uint32_t log_nf, uint32_t D) {
    return (log_nf == 1 && D >= 1 && D <= 3) || (log_nf == 2 && D >= 1 && D <= 3) || (log_nf == 3 && (D == 1 || D == 2)) || (log_nf == 4 && D == 1);
}

template <class H>
bool try_fri_tail(wf_ctx *ctx, uint32_t D, uint32_t log_nf, const FriTailParams &p) {
    if constexpr (!H::WAVE_TREE) {
        return false;
    } else {
#define WF_FT(LN, DD) if (log_nf == LN && D == DD) { hipLaunchKernelGGL((fri_tail_kernel<H, LN, DD>), dim3(1), dim3(1024), 0, ctx->stream, p); return true; }
        WF_FT(1, 1) WF_FT(1, 2) WF_FT(1, 3) WF_FT(2, 1) WF_FT(2, 2) WF_FT(2, 3) WF_FT(3, 1) WF_FT(3, 2) WF_FT(4, 1)
#undef WF_FT
        return false;
    }
}
}  // namespace

// used by wf_fri_layer_commit (fri.hip): *done = 0 when the caller should take the unfused path (small Rescue layers, where the
// lane-cooperative row hash wins)
int wf_fri_transpose_hash(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_evals, uint32_t log_rc, uint32_t log_nf,
                          void *d_transposed, void *d_leaves, int *done) {
    *done = 0;
    WF_TRY(check_hash(hash));
    if (field != WF_FIELD_F64 && (hash == WF_HASH_RP64_256 || hash == WF_HASH_RPJIVE64_256)) return WF_ERR_UNSUPPORTED;
    if (field != WF_FIELD_F62 && hash == WF_HASH_RP62_248) return WF_ERR_UNSUPPORTED;
    const uint64_t rc = 1ull << log_rc;
    const bool rescue = hash == WF_HASH_RP64_256 || hash == WF_HASH_RPJIVE64_256 || hash == WF_HASH_RP62_248;
    if (rescue && rc <= rcoop::COOP_MAX) return WF_OK;
    const int mode = field == WF_FIELD_F64 ? MODE_F64_CANON : (field == WF_FIELD_F62 ? MODE_F62_CANON : MODE_RAW);
    const uint32_t EW = ext_degree * (field == WF_FIELD_F128 ? 2 : 1);
    *done = 1;
    return with_hasher(hash, [&](auto h) {
        return launch_fri_rows<decltype(h)>(ctx, mode, (const uint64_t *)d_evals, rc, 1u << log_nf, EW, (uint64_t *)d_transposed, d_leaves);
    });
}

// FriProver::build_layers, layer k's fold + layer k + 1's transpose and leaf hashes in one launch (f64; *done = 0: not this shape,
// the caller runs the two steps separately).  io_*: the series offset^-1 * g^-i of the CURRENT layer, g_step = g^-(rows of the next
// layer), all in internal form; d_alpha on the device.
int wf_fri_fold_commit(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, uint32_t log_nf, const void *d_transposed, uint64_t rc, const void *io_lo,
                       const void *io_hi, uint32_t io_log_lo, const void *w16, uint64_t inv_n, const void *d_alpha, uint64_t g_step, void *d_folded,
                       void *d_transposed_next, void *d_leaves_next, int *done) {
    *done = 0;
    if (field != WF_FIELD_F64) return WF_OK;
    bool ok = false;
    wf_prof_begin(ctx, "fri_fold_commit");
    WF_TRY(with_hasher(hash, [&](auto h) {
        ok = try_fri_fold_commit<decltype(h)>(ctx, ext_degree, log_nf, (const uint64_t *)d_transposed, rc, (const uint64_t *)io_lo, (const uint64_t *)io_hi,
                                              io_log_lo, (const uint64_t *)w16, inv_n, (const uint64_t *)d_alpha, g_step, (uint64_t *)d_folded,
                                              (uint64_t *)d_transposed_next, d_leaves_next);
        return (int)WF_OK;
    }));
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    *done = ok ? 1 : 0;
    return WF_OK;
}

// FriProver::build_layers' last layers and set_remainder in one launch (fri_tail_kernel; f64, BLAKE3 family, rows of <= 128 bytes,
// first layer of at most FRI_TAIL_MAX_ROWS rows).  d_evals: the natural-order vector feeding layer 0 of this call (2^log_len
// elements); per layer k of the call d_transposed[k] / d_leaves[k] / d_nodes[k] / d_folded[k] as in wf_fri_build_layers; d_roots:
// num_layers (+ 1 with a remainder) digests, d_alphas: num_layers elements; io_*: the series offset^-1 g^-i of layer 0 of the call.
// *done = 0: not this shape, nothing was launched.
// ONE predicate for "the tail launch covers these layers", used by wf_fri_tail itself and by build_layers (fri.hip) when it decides
// BEFORE anything is queued which layers go to the tail (round-3 advice: two copies of these conditions that disagree would lose the
// transcript — layers 0 .. k0-1 have reseeded the device coin by the time wf_fri_tail says no).  log_len = log2 of the evaluations
// entering the first tail layer, nt = number of tail layers.
int wf_fri_tail_ok(int hash, int field, uint32_t ext_degree, uint32_t log_nf, uint32_t log_len, uint32_t nt) {
    if (field != WF_FIELD_F64 || nt == 0 || nt > (uint32_t)FRI_TAIL_MAX_LAYERS) return 0;
    if (log_len < (uint64_t)nt * log_nf || log_len < log_nf + 1) return 0;
    if ((1ull << (log_len - log_nf)) > FRI_TAIL_MAX_ROWS || ((ext_degree << log_nf) * 8) > 128) return 0;
    if (log_len - nt * log_nf < 1) return 0;     // every layer needs at least two rows (MerkleTree::new: TooFewLeaves)
    bool kernel = false;
    const int st = with_hasher(hash, [&](auto h) {
        kernel = decltype(h)::WAVE_TREE && fri_tail_has_kernel(log_nf, ext_degree);
        return (int)WF_OK;
    });
    return st == WF_OK && kernel ? 1 : 0;
}

// ... and for "the remainder (rem_size coefficients out of 2^log_rem_n evaluations) is computed and hashed inside the tail launch"
int wf_fri_tail_rem_ok(uint32_t ext_degree, uint32_t log_rem_n, uint32_t rem_size) {
    if (rem_size == 0 || (rem_size & (rem_size - 1)) || rem_size > 1024 || log_rem_n > 10) return 0;
    if ((uint64_t)rem_size * ext_degree * 8 > 1024) return 0;                       // its hash is one chunk in the kernel
    return ((uint64_t)rem_size << log_rem_n) * ext_degree <= (1u << 17) ? 1 : 0;     // partial sums of one workgroup
}

int wf_fri_tail(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, uint32_t log_nf, const void *d_evals, uint32_t log_len, uint32_t num_layers,
                void *const *d_transposed, void *const *d_leaves, void *const *d_nodes, void *const *d_folded, void *d_roots, void *d_alphas, void *d_coin,
                const void *io_lo, const void *io_hi, uint32_t io_log_lo, const void *w16, uint64_t inv_n, void *d_remainder, uint32_t rem_size,
                uint64_t rem_w_inv, uint64_t rem_off_inv, uint64_t rem_n_inv, int *done) {
    *done = 0;
    if (!wf_fri_tail_ok(hash, field, ext_degree, log_nf, log_len, num_layers)) return WF_OK;
    const uint32_t log_rows0 = log_len - log_nf;
    const uint32_t log_rem_n = log_len - num_layers * log_nf;
    if (d_remainder && !wf_fri_tail_rem_ok(ext_degree, log_rem_n, rem_size)) return WF_OK;
    FriTailParams p{};
    p.ev = (const uint64_t *)d_evals;
    p.log_rows0 = log_rows0;
    p.num_layers = num_layers;
    for (uint32_t k = 0; k < num_layers; k++) {
        p.tr[k] = (uint64_t *)d_transposed[k];
        p.leaves[k] = d_leaves[k];
        p.nodes[k] = d_nodes[k];
        p.folded[k] = (uint64_t *)d_folded[k];
    }
    p.roots = (uint32_t *)d_roots;
    p.alphas = (uint64_t *)d_alphas;
    p.coin = (CoinState *)d_coin;
    p.io_lo = (const uint64_t *)io_lo;
    p.io_hi = (const uint64_t *)io_hi;
    p.io_log_lo = io_log_lo;
    p.w16 = (const uint64_t *)w16;
    p.inv_n = inv_n;
    p.remainder = (uint64_t *)d_remainder;
    p.rem_size = rem_size;
    p.log_rem_n = log_rem_n;
    p.rem_w_inv = rem_w_inv;
    p.rem_off_inv = rem_off_inv;
    p.rem_n_inv = rem_n_inv;
    bool ok = false;
    wf_prof_begin(ctx, "fri_tail");
    WF_TRY(with_hasher(hash, [&](auto h) {
        ok = try_fri_tail<decltype(h)>(ctx, ext_degree, log_nf, p);
        return (int)WF_OK;
    }));
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    *done = ok ? 1 : 0;
    return WF_OK;
}
