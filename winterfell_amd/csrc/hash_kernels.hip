// Row hashing (ElementHasher::hash_elements / merge_many over matrix rows) and Merkle tree construction.
//
// Reference behaviour reproduced here:
//   RowMatrix::commit_to_rows          prover/src/matrix/row_matrix.rs:184-228 (+ PartitionOptions, air/src/options.rs:428-444)
//   Blake3_256::{hash_elements,merge}  crypto/src/hash/blake/mod.rs:33-65  (f64: canonical LE bytes of as_int())
//   Rp64_256::{hash_elements,merge}    crypto/src/hash/rescue/rp64_256/mod.rs:181-257
//   build_merkle_nodes                 crypto/src/merkle/mod.rs:344-368, concurrent.rs:26-75
//
// Kernels: one row (or one row partition) per lane for leaf hashing; one workgroup per 1024-input subtree for the
// tree (10 levels per launch, intermediate levels staged in LDS, every node written to the reference's heap layout).
#include "hashers.cuh"

namespace {

// leaf[r * parts + k] = H(words [k*part_words, min((k+1)*part_words, words_per_row)) of row r).  The partition index is
// blockIdx.y, so the message length is uniform across a workgroup (scalar branches in the block loop).
template <class H, int MODE, bool MULTI>
__global__ __launch_bounds__(256) void hash_rows_kernel(const uint64_t *rows, uint64_t num_rows, uint64_t row_width,
                                                        uint32_t elems_per_row, uint32_t part_elems, uint32_t parts, void *out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= num_rows) return;
    const uint32_t k = blockIdx.y;
    const uint32_t e0 = k * part_elems;
    const uint32_t e1 = (e0 + part_elems < elems_per_row) ? e0 + part_elems : elems_per_row;
    uint32_t d[8];
    H::template hash_elems<MODE, MULTI>(rows + r * row_width + e0, e1 - e0, d);
    store_digest(out, r * parts + k, d);
}


struct Seed {
    uint32_t w[8];
};

// digest[i] = merge_with_int(seed, first + i)
template <class H>
__global__ __launch_bounds__(256) void merge_with_int_kernel(Seed seed, uint64_t first, uint64_t count, void *out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    uint32_t d[8];
    H::merge_with_int(seed.w, first + gid, d);
    store_digest(out, gid, d);
}

// Proof-of-work search (prover/src/channel.rs:169-185): lanes test nonces first + gid; every nonce whose new seed has
// >= `factor` trailing zero bits in its 8-byte head competes for the minimum, which is what the reference's serial
// `find` returns.
template <class H>
__global__ __launch_bounds__(256) void grind_kernel(Seed seed, uint64_t first, uint64_t count, uint32_t factor,
                                                    unsigned long long *best) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    uint32_t d[8];
    H::merge_with_int(seed.w, first + gid, d);
    const uint64_t h = H::head(d);
    const uint32_t tz = h ? (uint32_t)__builtin_ctzll(h) : 64u;
    if (tz >= factor) atomicMin(best, (unsigned long long)(first + gid));
}

__global__ void gather_rows_kernel(const uint8_t *rows, uint64_t row_bytes, uint32_t take_bytes, const uint64_t *pos,
                                   uint32_t count, uint8_t *out) {
    const uint32_t words = take_bytes / 8;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)count * words) return;
    const uint32_t r = (uint32_t)(gid / words), w = (uint32_t)(gid % words);
    reinterpret_cast<uint64_t *>(out)[gid] = reinterpret_cast<const uint64_t *>(rows + pos[r] * row_bytes)[w];
}

#ifndef WF_HASH_WAVES
#define WF_HASH_WAVES 4      // waves per SIMD the wide-row hash kernels are compiled for (occupancy experiment: tools/build_variant.sh)
#endif
// Row hashes for rows of >= 64 bytes, BLAKE3 family.  One row per lane as in hash_rows_kernel, but a lane walking its own row
// reads 8 bytes per load at a row-sized stride (measured 0.26 TB/s on 256-byte rows: 32 columns x 2^23 rows took 8.4 ms for
// 1.3 ms of compressions).  Here a wavefront owns 64 consecutive rows and brings the message in one 64-byte block at a time:
// eight lanes load one row's block as 8 x u64 (64-byte contiguous segments, eight rows per load instruction), the values are
// canonicalised once by the loading lane, staged through 4.5 KiB of LDS per wavefront (wave-synchronous: DS operations of
// one wavefront execute in order), and every lane reads back its own row's 16 message words.
template <class H, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WF_HASH_WAVES, WF_HASH_WAVES))) void hash_rows_wide_kernel(const uint64_t *rows, uint64_t num_rows, uint64_t row_width, uint32_t elems_per_row,
                                                             uint32_t part_elems, uint32_t parts, void *out) {
    constexpr int BW = H::WIDE_BW;                                    // 64-bit words per message block (BLAKE3 8, SHA3 17)
    constexpr int PITCH = BW | 1;                                     // odd: lanes reading their own rows hit distinct banks
    __shared__ uint64_t stage_all[4][64 * PITCH];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    volatile uint64_t *st = stage_all[wave];
    const uint64_t r_base = ((uint64_t)blockIdx.x * 4 + wave) * 64;
    if (r_base >= num_rows) return;                                   // the whole wavefront leaves together
    const uint32_t k = blockIdx.y;
    const uint32_t e0 = k * part_elems;
    const uint32_t e1 = (e0 + part_elems < elems_per_row) ? e0 + part_elems : elems_per_row;
    const uint32_t nelem = e1 - e0;
    // Software pipeline (round 3): the loads of block blk + 1 are issued BEFORE block blk is compressed, so a wavefront's trip to
    // memory (the row-sized stride makes every load instruction touch eight lines) is in flight behind ~700 instructions of
    // compression instead of in front of them; hash_wide asks for the blocks in increasing order.  raw[] = the wavefront's share of
    // the NEXT block, as loaded (canonicalised when it is staged).
    uint64_t raw[BW];
    auto issue = [&](uint32_t blk) {
#pragma unroll
        for (uint32_t it = 0; it < BW; it++) {                        // 64 * BW words, 64 per step: runs of BW words per row
            const uint32_t idx = it * 64 + lane;
            const uint32_t rl = idx / BW, wq = idx - rl * BW;
            const uint32_t wi = blk * BW + wq;                        // word of the row part
            uint64_t row = r_base + rl;
            if (row >= num_rows) row = num_rows - 1;
            raw[it] = wi < nelem ? rows[row * row_width + e0 + wi] : 0ull;
        }
    };
    issue(0);
    auto fetch64 = [&](uint32_t blk, uint64_t (&m)[BW]) {
#pragma unroll
        for (uint32_t it = 0; it < BW; it++) {
            const uint32_t idx = it * 64 + lane;
            const uint32_t rl = idx / BW, wq = idx - rl * BW;
            uint64_t v = raw[it];
            if (MODE == MODE_F64_CANON) v = gl::to_int(v);            // zero stays zero
            else if (MODE == MODE_F62_CANON) v = f62::mul(f62::norm(v), 1);
            st[rl * PITCH + wq] = v;
        }
        issue(blk + 1);                                               // past the message: every lane's guard fails, raw = 0 (SHA3's pad-only block)
#pragma unroll
        for (int i = 0; i < BW; i++) m[i] = st[lane * PITCH + i];
    };
    uint32_t d[8];
    H::hash_wide(fetch64, nelem, d);
    if (r_base + lane < num_rows) store_digest(out, (r_base + lane) * parts + k, d);
}

// PartitionOptions rows (RowMatrix::commit_to_rows with num_partitions > 1, prover/src/matrix/row_matrix.rs:204-223), Blake3_256,
// partitions of >= 64 bytes: leaf = merge_many([hash_elements(partition k of the row)]) in ONE launch.  A wavefront owns 64 rows and
// walks their partitions in order with the block pipeline of hash_rows_wide_kernel running ACROSS partitions (the loads of the next
// block — of this partition or of the next one — are in flight behind the current compression); merge_many is a BLAKE3 hash of the
// concatenated digests, i.e. one more chunk whose 64-byte blocks are PAIRS of partition digests, so it is folded in as the pairs
// complete: the lane keeps one chaining value and at most one pending digest instead of all of them.  Round 2 wrote the
// 32 parts N bytes of partition digests to HBM and read them back in a second launch (configs[3]: 8 GiB each way).
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WF_HASH_WAVES, WF_HASH_WAVES))) void hash_rows_parts_blake3_kernel(const uint64_t *rows, uint64_t num_rows,
                                                                                                              uint64_t row_width, uint32_t elems_per_row,
                                                                                                              uint32_t part_elems, uint32_t parts, void *out) {
    constexpr int BW = 8, PITCH = BW | 1;
    __shared__ uint64_t stage_all[4][64 * PITCH];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    volatile uint64_t *st = stage_all[wave];
    const uint64_t r_base = ((uint64_t)blockIdx.x * 4 + wave) * 64;
    if (r_base >= num_rows) return;
    auto part_len = [&](uint32_t k) -> uint32_t {
        const uint32_t e0 = k * part_elems;
        return (e0 + part_elems < elems_per_row ? e0 + part_elems : elems_per_row) - e0;
    };
    // cursor of the load pipeline: (partition, block) of what `raw` will hold after the next issue()
    uint32_t nk = 0, nb = 0;
    uint64_t raw[BW];
    auto issue = [&]() {
        const bool live = nk < parts;
        const uint32_t e0 = nk * part_elems, nelem = live ? part_len(nk) : 0u;
#pragma unroll
        for (uint32_t it = 0; it < BW; it++) {
            const uint32_t idx = it * 64 + lane;
            const uint32_t rl = idx / BW, wq = idx - rl * BW;
            const uint32_t wi = nb * BW + wq;
            uint64_t row = r_base + rl;
            if (row >= num_rows) row = num_rows - 1;
            raw[it] = wi < nelem ? rows[row * row_width + e0 + wi] : 0ull;
        }
        if (live) {
            nb++;
            if (nb * BW >= nelem) {
                nb = 0;
                nk++;
            }
        }
    };
    issue();
    auto fetch64 = [&](uint32_t, uint64_t (&m)[BW]) {
#pragma unroll
        for (uint32_t it = 0; it < BW; it++) {
            const uint32_t idx = it * 64 + lane;
            const uint32_t rl = idx / BW, wq = idx - rl * BW;
            uint64_t v = raw[it];
            if (MODE == MODE_F64_CANON) v = gl::to_int(v);
            else if (MODE == MODE_F62_CANON) v = f62::mul(f62::norm(v), 1);
            st[rl * PITCH + wq] = v;
        }
        issue();
#pragma unroll
        for (int i = 0; i < BW; i++) m[i] = st[lane * PITCH + i];
    };
    // merge_many as a running BLAKE3 chunk over pairs of digests
    uint32_t cv[8], pend[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = b3::iv(i);
    const uint32_t nblocks = (parts + 1) / 2;
    for (uint32_t k = 0; k < parts; k++) {
        uint32_t d[8];
        HBlake3::hash_wide(fetch64, part_len(k), d);
        if ((k & 1u) == 0 && k + 1 < parts) {
#pragma unroll
            for (int i = 0; i < 8; i++) pend[i] = d[i];
            continue;
        }
        uint32_t m[16], o[8];
        const bool pair = (k & 1u) != 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            m[i] = pair ? pend[i] : d[i];
            m[8 + i] = pair ? d[i] : 0u;
        }
        const uint32_t blk = k >> 1;
        uint32_t flags = (blk == 0 ? b3::CHUNK_START : 0u) | (blk + 1 == nblocks ? (b3::CHUNK_END | b3::ROOT) : 0u);
        b3::compress(cv, m, 0, pair ? 64u : 32u, flags, o);
#pragma unroll
        for (int i = 0; i < 8; i++) cv[i] = o[i];
    }
    if (r_base + lane < num_rows) store_digest(out, r_base + lane, cv);
}

// rows[r][c * W + w] = cols[c * col_words + r * W + w]  (W = 64-bit words per matrix element): column-major -> row-major
// through an LDS tile of R rows so that both the column reads (R * W consecutive words) and the row writes are coalesced
__global__ __launch_bounds__(256) void cols_to_rows_kernel(const uint64_t *cols, uint64_t col_words, uint32_t num_cols, uint32_t W,
                                                           uint64_t num_rows, uint32_t R, uint64_t *rows) {
    extern __shared__ uint64_t tile[];                 // [R][num_cols * W + 1]
    const uint32_t rw = num_cols * W, pitch = rw + 1;
    const uint64_t r0 = (uint64_t)blockIdx.x * R;
    const uint32_t nr = num_rows - r0 < R ? (uint32_t)(num_rows - r0) : R;
    const uint32_t run = nr * W;                       // consecutive words of one column in this tile
    for (uint32_t idx = threadIdx.x; idx < num_cols * run; idx += 256) {
        const uint32_t c = idx / run, k = idx % run;
        tile[(k / W) * pitch + c * W + (k % W)] = cols[(uint64_t)c * col_words + r0 * W + k];
    }
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < nr * rw; idx += 256) {
        const uint32_t r = idx / rw, j = idx % rw;
        rows[(r0 + r) * rw + j] = tile[r * pitch + j];
    }
}

template <class H, int MODE, bool MULTI>
int launch_hash_rows_t(wf_ctx *ctx, const uint64_t *rows, uint64_t num_rows, uint64_t row_width, uint32_t elems_per_row,
                       uint32_t part_elems, uint32_t parts, void *out) {
    const uint64_t blocks = (num_rows + 255) / 256;
    if (blocks > 0x7fffffffull || parts > 65535) return WF_ERR_DOMAIN_TOO_LARGE;
    if constexpr (H::WIDE) {
        if (MODE != MODE_DIGESTS && part_elems >= (uint32_t)H::WIDE_BW && elems_per_row >= (uint32_t)H::WIDE_BW) {
            wf_prof_begin(ctx, H::row_name());
            hipLaunchKernelGGL((hash_rows_wide_kernel<H, MODE>), dim3((uint32_t)blocks, parts), dim3(256), 0, ctx->stream, rows, num_rows, row_width,
                               elems_per_row, part_elems, parts, out);
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            return WF_OK;
        }
    }
    if constexpr (H::COOP) {
        if (num_rows * parts <= rcoop::COOP_MAX) {      // few rows: latency-bound, spread each state over 16 lanes
            wf_prof_begin(ctx, H::row_name());
            hipLaunchKernelGGL((rcoop::hash_rows_kernel<typename H::Coop>), dim3((uint32_t)((num_rows + 15) / 16), parts), dim3(256), 0,
                               ctx->stream, rows, num_rows, row_width, elems_per_row, part_elems, parts, (uint64_t *)out);
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            return WF_OK;
        }
    }
    wf_prof_begin(ctx, H::row_name());
    hipLaunchKernelGGL((hash_rows_kernel<H, MODE, MULTI>), dim3((uint32_t)blocks, parts), dim3(256), 0, ctx->stream, rows,
                       num_rows, row_width, elems_per_row, part_elems, parts, out);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

// all sizes in 64-bit words
template <class H>
int launch_hash_rows(wf_ctx *ctx, const uint64_t *rows, uint64_t num_rows, uint64_t row_width, uint32_t elems_per_row,
                     uint32_t part_elems, uint32_t parts, int mode, void *out) {
    const bool multi = part_elems > 128;   // more than one 1024-byte BLAKE3 chunk (ignored by Rescue)
#define WF_HR(MODE)                                                                                                              \
    return multi ? launch_hash_rows_t<H, MODE, true>(ctx, rows, num_rows, row_width, elems_per_row, part_elems, parts, out)       \
                 : launch_hash_rows_t<H, MODE, false>(ctx, rows, num_rows, row_width, elems_per_row, part_elems, parts, out)
    switch (mode) {
        case MODE_F64_CANON: WF_HR(MODE_F64_CANON);
        case MODE_F62_CANON: WF_HR(MODE_F62_CANON);
        case MODE_DIGESTS: WF_HR(MODE_DIGESTS);
        default: WF_HR(MODE_RAW);
    }
#undef WF_HR
}

// Narrow traces (one 8-column group per row): the LDE transpose of fft_api.hip (coset-major tmp[bc][u][m] -> row-major
// lde[(u + b*m)][8]) with the leaf hash folded in.  A tile holds b * TM <= 256 COMPLETE rows in LDS, so after the coalesced
// row-major store every lane hashes one row straight from the tile: the separate row-hash kernel, and its read of the
// whole matrix, disappear.  W = 64-bit words per base element.
template <class H, int MODE, class T>
__global__ __launch_bounds__(256) void lde_transpose_hash_kernel(const T *tmp, T *lde, uint32_t base_cols, uint32_t log_n, uint32_t log_b,
                                                                uint32_t log_tm, void *leaves) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lde_smem[];
    T *tile = reinterpret_cast<T *>(lde_smem);
    constexpr uint32_t W = sizeof(T) / 8;
    const uint64_t n = 1ull << log_n;
    const uint32_t b = 1u << log_b, TM = 1u << log_tm;
    const uint32_t row = b * 8 + 1;
    const uint64_t mt = blockIdx.x;
    const uint32_t total = b * 8 * TM;
    for (uint32_t idx = threadIdx.x; idx < total; idx += 256) {
        const uint32_t ml = idx & (TM - 1), uc = idx >> log_tm;
        const uint32_t u = uc & (b - 1), cl = uc >> log_b;
        const uint64_t m = (mt << log_tm) + ml;
        T v = 0;
        if (cl < base_cols && m < n) v = tmp[(((uint64_t)cl << log_b) + u) * n + m];
        tile[ml * row + u * 8 + cl] = v;
    }
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < total; idx += 256) {
        const uint32_t cl = idx & 7, u = (idx >> 3) & (b - 1), ml = idx >> (3 + log_b);
        const uint64_t m = (mt << log_tm) + ml;
        if (m < n) lde[(u + ((uint64_t)m << log_b)) * 8 + cl] = tile[ml * row + u * 8 + cl];
    }
    for (uint32_t r = threadIdx.x; r < b * TM; r += 256) {
        const uint32_t u = r & (b - 1), ml = r >> log_b;
        const uint64_t m = (mt << log_tm) + ml;
        if (m >= n) continue;
        uint32_t d[8];
        H::template hash_elems<MODE, false>(reinterpret_cast<const uint64_t *>(tile + ml * row + u * 8), base_cols * W, d);
        store_digest(leaves, u + (m << log_b), d);
    }
}

// The same with one row per lane (BLAKE3 family; blowup <= 256): lane (u, ml) loads its row's base_cols values from the coset-major
// columns (lanes of a coset run along m: contiguous), hashes them out of registers, and only the row-major copy goes through
// LDS: the workgroup's 256 rows (TM = 256 / b positions x b cosets) are one contiguous 16 KiB (f128: 32 KiB) block of the matrix,
// staged with 16-byte chunks XOR-swizzled by row and written back 16 bytes per lane in address order.  Against the tile kernel
// above: no index arithmetic per element, no 8-byte LDS traffic for the hash, 16-byte stores.
template <class H, int MODE, class T>
__global__ __launch_bounds__(256) void lde_rows_direct_kernel(const T *tmp, T *lde, uint32_t base_cols, uint32_t log_n, uint32_t log_b, void *leaves) {
    constexpr int W = sizeof(T) / 8, RW = 8 * W, CP = RW / 2;   // words / 16-byte chunks of a padded row
    constexpr uint32_t Q = 16 / CP;                              // rows per 256 bytes of LDS (one pass over the banks)
    __shared__ uint4 stage[256 * CP];
    const uint64_t n = 1ull << log_n;
    const uint32_t log_tm = 8 - log_b, TM = 1u << log_tm;
    const uint32_t tid = threadIdx.x, ml = tid & (TM - 1), u = tid >> log_tm;
    const uint64_t m0 = (uint64_t)blockIdx.x << log_tm, m = m0 + ml;
    // Writers are lanes with consecutive ml (rows b apart), readers lanes with consecutive rows: place row r at
    // r ^ (ml & (Q - 1)) and rotate its chunks by (ml / Q) — sixteen writer lanes and sixteen reader lanes then both touch
    // sixteen different 16-byte bank groups (log_b >= 1 keeps the row map a bijection: it XORs bits below log2 Q with bits from log_b up)
    auto phys = [&](uint32_t row, uint32_t c) -> uint32_t {
        const uint32_t mr = row >> log_b;
        return (row ^ (mr & (Q - 1))) * CP + (c ^ ((mr / Q) & (CP - 1)));
    };
    uint64_t w[RW];
#pragma unroll
    for (int i = 0; i < RW; i++) w[i] = 0;
    if (m < n) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            if ((uint32_t)c < base_cols) {
                const T v = tmp[(((uint64_t)c << log_b) + u) * n + m];
                if constexpr (W == 1) {
                    w[c] = (uint64_t)v;
                } else {
                    w[2 * c] = (uint64_t)v;
                    w[2 * c + 1] = (uint64_t)(v >> 64);
                }
            }
        }
        uint32_t d[8];
        H::template hash_elems<MODE, false>(w, base_cols * W, d);
        store_digest(leaves, u + (m << log_b), d);
    }
    const uint32_t row = (ml << log_b) + u;                       // the lane's row within the workgroup's block
#pragma unroll
    for (int c = 0; c < CP; c++)
        stage[phys(row, c)] = make_uint4((uint32_t)w[2 * c], (uint32_t)(w[2 * c] >> 32), (uint32_t)w[2 * c + 1], (uint32_t)(w[2 * c + 1] >> 32));
    __syncthreads();
    if (m0 >= n) return;
    const uint64_t left = (n - m0) << log_b;                      // rows of the matrix from this block's first row on
    const uint32_t nv = left < 256 ? (uint32_t)left : 256u;
    uint4 *out = reinterpret_cast<uint4 *>(lde + (m0 << log_b) * 8);
#pragma unroll
    for (int i = 0; i < CP; i++) {
        const uint32_t L = i * 256 + tid, r = L / CP, cc = L % CP;
        if (r < nv) out[L] = stage[phys(r, cc)];
    }
}

template <class H, class T>
int launch_lde_transpose_hash(wf_ctx *ctx, int mode, const void *tmp, void *lde, uint32_t base_cols, uint32_t log_n, uint32_t log_b,
                              uint32_t log_tm, void *leaves) {
    if constexpr (H::WAVE_TREE) {
        const bool shape_ok = log_b >= 1 && log_b <= 8 && log_n + log_b >= 8 && base_cols <= 8 && (((uint64_t)1 << (log_n + log_b)) >> 8) <= 0x7fffffffull;
        const bool mode_ok = (sizeof(T) == 8 && mode == MODE_F64_CANON) || (sizeof(T) == 16 && mode == MODE_RAW);
        if (shape_ok && mode_ok && log_n >= 8 - log_b) {
            const uint32_t blocks = (uint32_t)(((uint64_t)1 << (log_n + log_b)) >> 8);
            wf_prof_begin(ctx, "lde_transpose_hash");
            if constexpr (sizeof(T) == 8)
                hipLaunchKernelGGL((lde_rows_direct_kernel<H, MODE_F64_CANON, T>), dim3(blocks), dim3(256), 0, ctx->stream, (const T *)tmp, (T *)lde, base_cols,
                                   log_n, log_b, leaves);
            else
                hipLaunchKernelGGL((lde_rows_direct_kernel<H, MODE_RAW, T>), dim3(blocks), dim3(256), 0, ctx->stream, (const T *)tmp, (T *)lde, base_cols,
                                   log_n, log_b, leaves);
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            return WF_OK;
        }
    }
    const uint64_t n = 1ull << log_n;
    const uint64_t blocks = (n + (1ull << log_tm) - 1) >> log_tm;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    const size_t lds = ((size_t)1 << log_tm) * ((8u << log_b) + 1) * sizeof(T);
    wf_prof_begin(ctx, "lde_transpose_hash");
#define WF_LT(MODE) hipLaunchKernelGGL((lde_transpose_hash_kernel<H, MODE, T>), dim3((uint32_t)blocks), dim3(256), lds, ctx->stream, (const T *)tmp, (T *)lde, base_cols, log_n, log_b, log_tm, leaves)
    switch (mode) {
        case MODE_F64_CANON: WF_LT(MODE_F64_CANON); break;
        case MODE_F62_CANON: WF_LT(MODE_F62_CANON); break;
        default: WF_LT(MODE_RAW); break;
    }
#undef WF_LT
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

}  // namespace


// used by wf_build_trace_commitment through wf_evaluate_polys_over_fused (fft_api.hip).  Only the byte hashers take the fused
// path: a Rescue row hash is three orders of magnitude more arithmetic than the transpose it would be fused with.
int wf_lde_transpose_hash(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_tmp, void *d_lde, uint32_t base_cols,
                          uint32_t log_n, uint32_t log_b, uint32_t log_tm, void *d_leaves, int *done) {
    *done = 0;
    if (hash != WF_HASH_BLAKE3_256 && hash != WF_HASH_BLAKE3_192 && hash != WF_HASH_SHA3_256) return WF_OK;
    if (!d_leaves) return WF_ERR_INVALID_ARG;
    (void)ext_degree;
    const int mode = field == WF_FIELD_F64 ? MODE_F64_CANON : (field == WF_FIELD_F62 ? MODE_F62_CANON : MODE_RAW);
    *done = 1;
    return with_hasher(hash, [&](auto h) {
        typedef decltype(h) H;
        if (field == WF_FIELD_F128) return launch_lde_transpose_hash<H, f128::u128>(ctx, mode, d_tmp, d_lde, base_cols, log_n, log_b, log_tm, d_leaves);
        return launch_lde_transpose_hash<H, uint64_t>(ctx, mode, d_tmp, d_lde, base_cols, log_n, log_b, log_tm, d_leaves);
    });
}

extern "C" int wf_hash_merge_batch(wf_ctx *ctx, int hash, const void *d_pairs, uint64_t count, void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_pairs || !d_out) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (count == 0) return WF_OK;
    const uint64_t blocks = (count + 255) / 256;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    WF_TRY(with_hasher(hash, [&](auto h) {
        typedef decltype(h) H;
        if constexpr (H::COOP) {
            if (count <= rcoop::COOP_MAX) {
                hipLaunchKernelGGL((rcoop::merge_kernel<typename H::Coop>), dim3((uint32_t)((count + 15) / 16)), dim3(256), 0, ctx->stream,
                                   (const uint64_t *)d_pairs, count, (uint64_t *)d_out);
                return (int)WF_OK;
            }
        }
        hipLaunchKernelGGL(merge_batch_kernel<H>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, d_pairs, count, d_out);
        return (int)WF_OK;
    }));
    WF_HIP(hipGetLastError());
    return WF_OK;
}

static int hash_rows_impl(wf_ctx *ctx, int hash, int field, uint32_t D, const void *d_rows, uint64_t num_rows,
                          uint64_t row_width, uint32_t elems_per_row, uint32_t num_partitions, uint32_t hash_rate,
                          void *d_leaves) {
    if (!ctx || !d_rows || !d_leaves || num_rows == 0 || D == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128 && field != WF_FIELD_F62) return WF_ERR_UNSUPPORTED;
    if (field != WF_FIELD_F64 && (hash == WF_HASH_RP64_256 || hash == WF_HASH_RPJIVE64_256)) return WF_ERR_UNSUPPORTED;   // Rescue over f64 only
    if (field != WF_FIELD_F62 && hash == WF_HASH_RP62_248) return WF_ERR_UNSUPPORTED;                                       // Rescue over f62 only
    if (elems_per_row > row_width || elems_per_row % D) return WF_ERR_INVALID_ARG;
    if (num_partitions < 1 || num_partitions > 16 || hash_rate < 1 || hash_rate > 256) return WF_ERR_INVALID_ARG;
    hash_rate &= 0xffu;   // PartitionOptions::new stores `hash_rate as u8` (air/src/options.rs:414-418): the permitted 256 wraps to 0
    // f64 is not IS_CANONICAL: hash the canonical LE bytes; f128 is: hash the raw element bytes (blake/mod.rs:52-65).
    // Rows are addressed in 64-bit words: an f128 element is two words.
    const int mode = field == WF_FIELD_F64 ? MODE_F64_CANON : (field == WF_FIELD_F62 ? MODE_F62_CANON : MODE_RAW);
    const uint32_t W = field == WF_FIELD_F128 ? 2 : 1;
    row_width *= W;
    elems_per_row *= W;
    D *= W;
    const uint64_t *rows = (const uint64_t *)d_rows;
    // PartitionOptions::partition_size / num_partitions — air/src/options.rs:428-444 (in columns of E)
    const uint32_t num_cols = elems_per_row / D;
    uint32_t ps = num_cols;
    if (num_partitions > 1) {
        const uint32_t min_ps = hash_rate / (D / W);
        ps = (num_cols + num_partitions - 1) / num_partitions;
        if (ps < min_ps) ps = min_ps;
    }
    // row_matrix.rs:193: the single-hash path is taken only when partition_size == num_cols; a partition size ABOVE the
    // column count (the hash_rate floor with num_partitions > 1) still goes through merge_many over its one digest
    if (ps == num_cols) {
        return with_hasher(hash, [&](auto h) {
            return launch_hash_rows<decltype(h)>(ctx, rows, num_rows, row_width, elems_per_row, elems_per_row, 1, mode, d_leaves);
        });
    }
    const uint32_t parts = (num_cols + ps - 1) / ps;
#ifndef WF_NO_FUSED_PARTITIONS
    // Blake3_256, every partition at least one 64-byte block, at most 16 digests (one BLAKE3 chunk): partition hashes and merge_many
    // in one launch, no digest round trip (hash_rows_parts_blake3_kernel)
    if (hash == WF_HASH_BLAKE3_256 && ps * D >= 8 && elems_per_row - (parts - 1) * ps * D >= 1 && parts <= 16) {
        const uint64_t blocks = (num_rows + 255) / 256;
        if (blocks <= 0x7fffffffull) {
            wf_prof_begin(ctx, HBlake3::row_name());
#define WF_HP(MODE) hipLaunchKernelGGL((hash_rows_parts_blake3_kernel<MODE>), dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, rows, num_rows, row_width, elems_per_row, ps * D, parts, d_leaves)
            switch (mode) {
                case MODE_F64_CANON: WF_HP(MODE_F64_CANON); break;
                case MODE_F62_CANON: WF_HP(MODE_F62_CANON); break;
                default: WF_HP(MODE_RAW); break;
            }
#undef WF_HP
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            return WF_OK;
        }
    }
#endif
    void *tmp;
    WF_TRY(wf_scratch(ctx, 2, (size_t)num_rows * parts * 32, &tmp));
    // partition digests, then leaf = merge_many(partition digests): Blake3 hashes the raw digest bytes
    // (blake/mod.rs:37-39), Rp64_256 hashes the digests' 4*parts elements (rp64_256/mod.rs:194-196).
    return with_hasher(hash, [&](auto h) {
        typedef decltype(h) H;
        WF_TRY(launch_hash_rows<H>(ctx, rows, num_rows, row_width, elems_per_row, ps * D, parts, mode, tmp));
        return launch_hash_rows<H>(ctx, (const uint64_t *)tmp, num_rows, 4ull * parts, 4 * parts, 4 * parts, 1, MODE_DIGESTS, d_leaves);
    });
}

extern "C" int wf_hash_rows(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_rows, uint64_t num_rows,
                            uint64_t row_width, uint32_t elems_per_row, uint32_t num_partitions, uint32_t hash_rate,
                            void *d_leaves) {
    WF_ENTER(ctx);
    return hash_rows_impl(ctx, hash, field, ext_degree, d_rows, num_rows, row_width, elems_per_row, num_partitions,
                          hash_rate, d_leaves);
}

namespace {
template <class H>
__global__ __launch_bounds__(256) void hash_bytes_kernel(const uint64_t *msgs, uint64_t count, uint64_t stride_words, uint64_t nbytes, void *out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    uint32_t d[8];
    if constexpr (H::BYTES) H::hash_bytes(msgs + gid * stride_words, nbytes, d);
    store_digest(out, gid, d);
}
}  // namespace

extern "C" int wf_hash_bytes_batch(wf_ctx *ctx, int hash, const void *d_msgs, uint64_t count, uint64_t stride_bytes, uint64_t len_bytes,
                                   void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_out || (count && len_bytes && !d_msgs)) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (hash != WF_HASH_BLAKE3_256 && hash != WF_HASH_BLAKE3_192 && hash != WF_HASH_SHA3_256) return WF_ERR_UNSUPPORTED;
    if (stride_bytes % 8 || stride_bytes < ((len_bytes + 7) / 8) * 8 || len_bytes >= (1ull << 32)) return WF_ERR_INVALID_ARG;
    if (count == 0) return WF_OK;
    const uint64_t blocks = (count + 255) / 256;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    WF_TRY(with_hasher(hash, [&](auto h) {
        hipLaunchKernelGGL(hash_bytes_kernel<decltype(h)>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, (const uint64_t *)d_msgs, count,
                           stride_bytes / 8, len_bytes, d_out);
        return (int)WF_OK;
    }));
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_hash_columns(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_cols, uint32_t num_cols,
                               uint64_t col_stride, uint64_t num_rows, void *d_leaves) {
    WF_ENTER(ctx);
    if (!ctx || !d_cols || !d_leaves || num_cols == 0 || num_rows == 0 || ext_degree == 0) return WF_ERR_INVALID_ARG;
    if (col_stride < num_rows * ext_degree) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    const uint32_t fw = field == WF_FIELD_F128 ? 2 : 1;          // 64-bit words per base element
    const uint32_t W = ext_degree * fw;
    const uint64_t rw = (uint64_t)num_cols * W;
    if (rw > 4095) return WF_ERR_UNSUPPORTED;                    // one row must fit the LDS tile
    uint32_t R = (uint32_t)(4096 / (rw + 1));                   // <= 32 KiB of LDS
    if (R > 256) R = 256;
    if (R == 0) R = 1;
    void *tmp;
    WF_TRY(wf_scratch(ctx, 1, (size_t)num_rows * rw * 8, &tmp));
    const uint64_t blocks = (num_rows + R - 1) / R;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    wf_prof_begin(ctx, "cols_to_rows");
    hipLaunchKernelGGL(cols_to_rows_kernel, dim3((uint32_t)blocks), dim3(256), (size_t)R * (rw + 1) * 8, ctx->stream, (const uint64_t *)d_cols,
                       col_stride * fw, num_cols, W, num_rows, R, (uint64_t *)tmp);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    const uint32_t bc = num_cols * ext_degree;                   // base elements per row
    return hash_rows_impl(ctx, hash, field, ext_degree, tmp, num_rows, bc, bc, 1, 1, d_leaves);
}

extern "C" int wf_hash_elements_batch(wf_ctx *ctx, int hash, int field, const void *d_elems, uint64_t count,
                                      uint64_t row_width, uint32_t elems_per_row, void *d_out) {
    WF_ENTER(ctx);
    if (count == 0) return WF_OK;
    return hash_rows_impl(ctx, hash, field, 1, d_elems, count, row_width, elems_per_row, 1, 1, d_out);
}

extern "C" int wf_hash_merge_many_batch(wf_ctx *ctx, int hash, const void *d_digests, uint64_t count, uint32_t k, void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_digests || !d_out || k == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (count == 0) return WF_OK;
    // Blake3: hash of the concatenated digest bytes (blake/mod.rs:37-39); Rp64_256: hash_elements over the 4*k digest
    // elements (rp64_256/mod.rs:194-196).  Both are "raw words" for the row kernel.
    return with_hasher(hash, [&](auto h) {
        return launch_hash_rows<decltype(h)>(ctx, (const uint64_t *)d_digests, count, 4ull * k, 4 * k, 4 * k, 1, MODE_DIGESTS, d_out);
    });
}

extern "C" int wf_hash_merge_with_int_batch(wf_ctx *ctx, int hash, const void *h_seed, uint64_t first_value, uint64_t count,
                                           void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !h_seed || !d_out) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (count == 0) return WF_OK;
    if (count - 1 > ~0ull - first_value) return WF_ERR_INVALID_ARG;   // the range must not wrap past u64::MAX
    const uint64_t blocks = (count + 255) / 256;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    Seed seed;
    memcpy(seed.w, h_seed, 32);
    WF_TRY(with_hasher(hash, [&](auto h) {
        typedef decltype(h) H;
        if constexpr (H::COOP) {
            if (count <= rcoop::COOP_MAX) {
                rcoop::SeedWords sw;
                memcpy(sw.w, seed.w, 32);
                hipLaunchKernelGGL((rcoop::merge_with_int_kernel<typename H::Coop>), dim3((uint32_t)((count + 15) / 16)), dim3(256), 0,
                                   ctx->stream, sw, first_value, count, (uint64_t *)d_out);
                return (int)WF_OK;
            }
        }
        hipLaunchKernelGGL(merge_with_int_kernel<H>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, seed, first_value, count, d_out);
        return (int)WF_OK;
    }));
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_grind(wf_ctx *ctx, int hash, const void *h_seed, uint32_t grinding_factor, uint64_t first_nonce,
                        uint64_t max_nonce, uint64_t *h_nonce) {
    WF_ENTER(ctx);
    if (!ctx || !h_seed || !h_nonce || grinding_factor > 64 || first_nonce > max_nonce) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    Seed seed;
    memcpy(seed.w, h_seed, 32);
    void *tmp;
    WF_TRY(wf_scratch(ctx, 2, 8, &tmp));
    unsigned long long *d_best = (unsigned long long *)tmp;
    // batch ~ twice the expected number of trials, within [2^16, 2^24] lanes per launch
    uint32_t lg = grinding_factor + 1;
    if (lg < 16) lg = 16;
    if (lg > 24) lg = 24;
    if ((hash == WF_HASH_RP64_256 || hash == WF_HASH_RPJIVE64_256 || hash == WF_HASH_RP62_248) && lg > 21) lg = 21;   // a Rescue permutation is ~200x a BLAKE3 block
    if (hash == WF_HASH_SHA3_256 && lg > 23) lg = 23;
    const uint64_t batch = 1ull << lg;
    uint64_t first = first_nonce;
    for (;;) {
        const uint64_t left = max_nonce - first;               // nonces first .. max_nonce inclusive = left + 1
        const uint64_t count = left < batch - 1 ? left + 1 : batch;
        WF_HIP(hipMemsetAsync(d_best, 0xff, 8, ctx->stream));
        WF_TRY(with_hasher(hash, [&](auto h) {
            typedef decltype(h) H;
            wf_prof_begin(ctx, H::grind_name());
            hipLaunchKernelGGL(grind_kernel<H>, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, seed, first, count,
                               grinding_factor, d_best);
            wf_prof_end(ctx);
            return (int)WF_OK;
        }));
        WF_HIP(hipGetLastError());
        unsigned long long best;
        WF_TRY(wf_copy_d2h(ctx, &best, d_best, 8));
        if (best != ~0ull) {
            *h_nonce = best;
            return WF_OK;
        }
        if (count == left + 1) return WF_ERR_NOT_FOUND;
        first += count;
    }
}

extern "C" int wf_rows_fetch(wf_ctx *ctx, const void *d_rows, uint64_t row_width, uint32_t elems_per_row,
                             uint32_t elem_bytes, const uint64_t *h_positions, uint32_t count, void *h_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_rows || !h_positions || !h_out || (elem_bytes != 8 && elem_bytes != 16)) return WF_ERR_INVALID_ARG;
    if (count == 0) return WF_OK;
    const uint64_t row_bytes = row_width * elem_bytes;
    const uint32_t take = elems_per_row * elem_bytes;
    void *tmp;
    WF_TRY(wf_scratch(ctx, 2, (size_t)count * (take + 8), &tmp));
    uint64_t *d_pos = (uint64_t *)tmp;
    uint8_t *d_out = (uint8_t *)tmp + (size_t)count * 8;
    WF_TRY(wf_copy_h2d(ctx, d_pos, h_positions, (size_t)count * 8));
    const uint64_t total = (uint64_t)count * (take / 8);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint8_t *)d_rows, row_bytes, take, d_pos, count, d_out);
    WF_HIP(hipGetLastError());
    WF_TRY(wf_copy_d2h(ctx, h_out, d_out, (size_t)count * take));
    return wf_check_status(ctx);
}

extern "C" int wf_build_trace_commitment(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, void *d_trace,
                                         uint32_t num_cols, uint64_t col_stride, uint32_t log_n, uint32_t log_blowup,
                                         const void *h_offset, uint32_t num_partitions, uint32_t hash_rate,
                                         int skip_interpolate, void *d_lde, void *d_leaves, void *d_nodes, void *h_root) {
    WF_ENTER(ctx);
    if (!ctx || !d_trace || !d_lde || !d_leaves || !d_nodes) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (ext_degree == 0 || num_cols == 0 || num_partitions < 1 || num_partitions > 16 || hash_rate < 1 || hash_rate > 256) return WF_ERR_INVALID_ARG;
    // extend_execution_trace (in place: every argument is validated before the trace is touched)
    if (!skip_interpolate) WF_TRY(wf_interpolate_columns(ctx, field, ext_degree, d_trace, num_cols, col_stride, log_n));
    // (for narrow traces committed with a byte hasher and no partitions the LDE transpose also produces the row hashes)
    const uint64_t N = 1ull << (log_n + log_blowup);
    const uint64_t rw = wf_row_width(num_cols, ext_degree);
    bool one_partition = num_partitions == 1;
    if (!one_partition) {   // PartitionOptions::partition_size (air/src/options.rs:428-437): one partition if it covers all columns
        const uint32_t min_ps = (hash_rate & 0xffu) / ext_degree;
        uint32_t ps = (num_cols + num_partitions - 1) / num_partitions;
        if (ps < min_ps) ps = min_ps;
        one_partition = ps == num_cols;   // row_matrix.rs:193 (ps > num_cols keeps the merge_many step)
    }
    int fused = 0;
    WF_TRY(wf_evaluate_polys_over_fused(ctx, field, ext_degree, d_trace, num_cols, col_stride, log_n, log_blowup, h_offset, d_lde,
                                        one_partition ? hash : -1, d_leaves, &fused));
    // compute_execution_trace_commitment
    if (!fused) WF_TRY(wf_hash_rows(ctx, hash, field, ext_degree, d_lde, N, rw, num_cols * ext_degree, num_partitions, hash_rate, d_leaves));
    WF_TRY(wf_merkle_build(ctx, hash, d_leaves, N, d_nodes));
    if (h_root) WF_TRY(wf_memcpy_d2h(ctx, h_root, (const uint8_t *)d_nodes + 32, 32));
    return WF_OK;
}
