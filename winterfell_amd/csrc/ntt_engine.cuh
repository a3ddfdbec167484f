// NTT engine for gfx950, generic over the field policy (fields.cuh); the design notes below are written for
// f64 (Goldilocks), where the register-DFT twiddles are shifts; other fields use table constants instead.
//
// Computes what the reference's math::fft computes (math/src/fft/mod.rs:85-386, serial.rs, concurrent.rs,
// fft_inputs.rs:215-252): natural-order in, natural-order out DFTs over the 2^k-th roots of unity of the
// f64 field, with optional coset scaling on the input (evaluate_poly_with_offset) or output
// (interpolate_poly_with_offset).  The reference's recursive radix-2 butterflies + bit-reversal permute
// are NOT mirrored: field arithmetic is exact, so any DFT algorithm yields the same canonical values.
//
// Algorithm: a transform of n = 2^L points is split into P = ceil(L/8) passes of radix R_p = 2^(r_p)
// (r_p <= 8, decimation in frequency, most significant index digit first).  One pass =
//   - each 256-thread workgroup owns a tile of R_p "rows" (stride S_p = n / (R_1..R_p)) x T "columns";
//     a thread holds A = 2^LOG_A elements in registers and runs an A-point DFT whose internal twiddles
//     are powers of omega_16 = 2^12 (shifts, no multiplies; omega_64 = 8 in this field, f64/mod.rs:258-267),
//   - multiplies by omega_{R_p}^(k_a * b), exchanges through LDS (padded, conflict-free),
//   - runs the B-point DFT, multiplies by the inter-pass twiddles omega_n^(k' * rem * mult): a lane's 16 outputs
//     k' = k0 + 16 i form a geometric progression base * step^i, so two look-ups in the two-level table and a
//     multiplication chain replace sixteen look-ups (the same trick scales the inputs of a coset transform),
//   - stores.  Passes 1..P-1 store in place (tile rows at the same addresses, coalesced along the
//     columns); the last pass runs along the contiguous axis and stores digit-reversed so that the
//     result is in natural order (coalesced along the tile's columns, which are the low output digits), or, for
//     the LDE of a wide trace, straight into the row-major matrix (NttJob::rowmajor).
// The inverse transform is the forward transform with the output index negated (k -> -k mod n) and a
// 1/n scale, so only forward tables exist.
//
// HBM traffic: P reads + P writes of the data (P = 3 at n = 2^24), see DESIGN.md.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "blake3.cuh"
#include "dft_regs.cuh"
#include "l24.cuh"
#include "ntt_params.cuh"
#include "ntt_big.cuh"
#include "tables.cuh"
#include "wf_internal.h"

namespace {

// At least four waves per SIMD (at most 128 VGPRs): the f64 radix-256 progression pass compiles to 129 on its own — three waves — and
// fits 128 without a spill when asked (round 5, same box, alternating runs: 57.1-58.4 against 58.7-60.2 us per pass at 2^24; every
// other pass kernel is below 128 already and keeps its own count).  -DNTT_WAVES_PER_EU=k pins the occupancy for experiments.
#ifndef NTT_MIN_WAVES
#define NTT_MIN_WAVES 4
#endif
#ifndef NTT_WAVES_PER_EU
#define NTT_WAVES_ATTR
#else
#define NTT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(NTT_WAVES_PER_EU, NTT_WAVES_PER_EU)))
#endif

// TWTAB (non-last passes): the inter-pass twiddles come from the pass's L2-resident table (p.tw_tab) instead of the per-lane
// progression; a compile-time switch so that neither variant carries the other's registers
//
// PF (prefetch, an experiment kept behind WF_NTT_PREFETCH=1): the launch is a grid of persistent workgroups that walk the
// tiles (tile = blockIdx.x, + gridDim.x, ...) and issue the NEXT tile's global loads right after the LDS exchange barrier, so
// that they are in flight during step 2 (the second DFT, the inter-pass twiddles, the stores) instead of being waited for at
// the top of a fresh workgroup: twice the bytes in flight per workgroup for 32 more VGPRs.  Measured result (round 2, 2^24
// f64): 304 us per transform against 227 us for the plain launches — the extra registers cost a wave per SIMD (168-VGPR cap,
// with spills in the last-pass kernel; at two waves and no spills: 333 us), and occupancy is worth more to this kernel than
// memory-level parallelism.  The vm counter retires in order, so step 2 must not wait on loads ISSUED AFTER the prefetch: the
// table kernel (32 table loads per lane in step 2) is never launched with PF.
#ifndef NTT_PF_WAVES
#define NTT_PF_WAVES 3     // waves per SIMD the prefetching variants are compiled for (<= 168 VGPRs)
#endif
// RH (rows + hash, the last pass of a NARROW trace's LDE — at most one 8-column group — when the caller wants the Blake3_256 row
// digests as well): the pass tiles over [position c][coset u][column] with the column fastest, so that a tile's outputs are, for
// every output digit k', TC / cp2 CONSECUTIVE rows of the row-major LDE matrix (row = u + b (c + ncols k') = j + b ncols k' with
// j = c b + u the tile's column index divided by the padded column count cp2).  The outputs go to LDS instead of a coset-major
// buffer, then every lane takes whole rows: hashes them out of LDS (leaf = Blake3_256::hash_elements(row), one compression),
// stores the digest and the 64-byte padded row.  The coset-major round trip (write + read of the whole LDE) and the separate
// transpose + hash launch disappear (2^20 x 4 x blowup 8: lde_transpose_hash 277 us + last pass 120 us -> one launch).
// f64 inter-pass twiddle TABLES (TWTAB kernels, WF_NTT_F64_TABLES=1): one Montgomery word per entry and one product per element — the
// progression's second product (cur *= stp, ~18 instructions of a pass's 114 per element) becomes an 8-byte load from a table that
// lives in L2 (a single transform's pass of <= 2^17 twiddles) or in the Infinity Cache (shared by the >= 8 vectors of a batch, <= 2^22).
// -DNTT_F64_TW_ROWS=1 restores round 2's four-word rows (the multiplication rides on the exit from the limb form, but 32 bytes of table
// per element through the vector memory pipeline cost what the chain cost: 75.9 against 73.6-76.6 us per pass).
#ifndef NTT_F64_TW_ROWS
#define NTT_F64_TW_ROWS 0
#endif
// VT (vector tiles, round 6; f64, the passes of a wide trace's coset LDE): the tile's TC columns are TC VECTORS (base columns of one
// coset) at ONE position instead of TC positions of one vector, on column-interleaved buffers ([position][column]: polynomials
// transposed once, the ping buffer [coset][position][column]), so that every access is still a 128 / 256-byte run — and every twiddle
// of the tile is shared by its columns: the inter-pass twiddles omega^(k' rem) are R values per tile, expanded once per workgroup
// into four-word rows in LDS and applied by the exit from the limb form (the multiplication is free there, as in step 1), instead
// of two Montgomery products per element for the per-lane progression; the coset pre-scale (offset g^u)^j of the first pass is R
// values per tile as well (one look-up per lane into LDS, one product per element instead of two).  Per element of a non-last pass:
// 114 -> ~86 VALU instructions, first pass of an LDE 150 -> ~109.  The last pass only changes its lane <-> (row, column) assignment and
// its source address (columns are the contiguous axis of every VT buffer); its outputs — row-major rows, leaves — are as before.
//
// Block tiles for a SINGLE transform of three passes (round 6, VTM = 2 for its first pass): the same sharing for the MIDDLE pass of one
// vector.  Its twiddles omega^(k1 a2) do not depend on the block k0 (the first pass's output digit), so a tile of 16 consecutive BLOCKS at
// one a2 shares them — if the blocks are the contiguous axis.  The first pass therefore stores its outputs transposed, T[rem0 R0 + k0]
// (lanes of step 2 are dealt k_a-fastest so that a lane group's 16 consecutive k0 are one 128-byte run; the exchange buffer's columns are
// rotated by (32 / B) k_a for that read pattern), the middle pass IS the vector-tile kernel with C = R0 "columns" and no coset dimension, in
// place, and the last pass reads [(k1 S1 + a2) R0 + k0] with columns fastest (p.bt) and stores in natural order as always.
template <class F, int LOG_A, int LOG_B, bool LAST, bool TWTAB, bool PF, bool RH = false, int VTM = 0>
__global__ __launch_bounds__(256) NTT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(PF ? NTT_PF_WAVES : (F::USE_L24 ? NTT_MIN_WAVES : 1)))) void ntt_pass(PassParams<typename F::T> p) {
    constexpr bool VT = VTM == 1;       // vector tiles (and the block-tile plan's middle and last pass)
    constexpr bool BT0 = VTM == 2;      // the block-tile plan's first pass: standard tile, transposed output
    static_assert(!VT || (F::USE_L24 && !TWTAB && LOG_B >= 3 && !(PF && LAST)), "vector tiles: f64 passes of radix 64 .. 256");
    static_assert(!BT0 || (F::USE_L24 && !PF && !TWTAB && !LAST && !RH && LOG_B >= 3), "block-transposing first pass: f64, radix 64 .. 256");
    static_assert(!(LAST && TWTAB), "the last pass has no inter-pass twiddles");
    static_assert(!RH || (LAST && !PF && F::USE_L24 && LOG_B >= 3), "rows mode: f64 last passes of radix 64 / 128 / 256");
    static_assert(!(PF && TWTAB), "the table kernel's step-2 loads would drain the prefetch (in-order vm counter)");
    typedef typename F::T T;
    constexpr int A = 1 << LOG_A, B = 1 << LOG_B, LOG_R = LOG_A + LOG_B;
    constexpr int TC = 256 / B;         // tile columns
    constexpr int G = A / B;            // B-point DFTs per thread in step 2
    static_assert(B == 1 || G == 1 || G == 2, "A must be B or 2B");
    // LDS exchange buffer: non-last [k_a][b][t], last [k_a][t][b].  Bank conflicts are avoided by XOR-swizzling the index
    // inside a row instead of padding the rows: a radix-256 tile of 64-bit elements is then exactly 32 KiB, so FIVE
    // workgroups fit the CU's 160 KiB (padded rows: 34 KiB, four) — one more wave per SIMD for a VALU-bound kernel.
    //   non-last: rows of 256 elements; odd k_a rows have bit 4 of the column flipped (= a 16-element shift, what the
    //             padding did modulo the bank count): writes stay lane-linear, reads at fixed b spread over k_a parity
    //   last:     rows of B elements; b is XORed with (t ^ k_a) so that lanes that differ in t (stride B elements = the
    //             same banks) or in k_a land on different banks
#ifdef NTT_LDS_PADDED
    constexpr int ROW_NL = B * TC + 16;
    constexpr int ROW_L = B + 1;
    auto idx_nl = [](int ka, int col) -> int { return ka * ROW_NL + col; };
    auto idx_l = [](int ka, int t, int b) -> int { return (ka * TC + t) * ROW_L + b; };
#else
    constexpr int ROW_NL = B * TC;
    constexpr int ROW_L = B;
    auto idx_nl = [](int ka, int col) -> int { return ka * ROW_NL + (col ^ ((ka & 1) << 4)); };
    auto idx_l = [](int ka, int t, int b) -> int { return (ka * TC + t) * ROW_L + (b ^ ((t ^ ka) & (B - 1))); };
#endif
    // BT0: step 2 reads with k_a fastest across lanes; rotating a row's columns by (32 / B) k_a puts the B k_a x 32 / B columns of a
    // half-wave on 32 different 8-byte banks (rows are 256 elements = a multiple of the 32 banks), and step 1's writes stay linear
    auto idx_bt = [](int ka, int col) -> int { return ka * (B * TC) + ((col + (32 / B) * ka) & (B * TC - 1)); };
    constexpr int LDS_ELEMS = (B == 1) ? 1 : (LAST ? A * TC * ROW_L : A * ROW_NL);
    __shared__ T lds[LDS_ELEMS];
    // f64: the intra-pass twiddle rows (four words each, l24.cuh) a workgroup needs — omega_R^(k_a b), R rows — are copied to LDS
    // once per workgroup: as global loads they were 32 bytes per element through the vector memory pipeline, twice the
    // pipeline time of the tile's own data (64 B/clk/CU into registers, however few distinct lines the lanes touch).
#ifndef NTT_W256_IN_LDS
#define NTT_W256_IN_LDS 1
#endif
    constexpr bool WLDS = F::USE_L24 && B > 1 && NTT_W256_IN_LDS != 0;
    constexpr int WROWS = WLDS ? (1 << LOG_R) : 1;
    __shared__ uint4 wlds[2 * WROWS];
    static_assert(!VT || WLDS, "vector tiles ride on the limb passes");
    // VT: the tile's inter-pass twiddle rows take wlds' place once step 1 is through with it, the coset pre-scale factors of a first pass
    // sit in the exchange buffer until step 1 has read them (one more barrier each): no LDS beyond the 40 KiB of a radix-256 pass, i.e.
    // four workgroups per CU (with arrays of their own, 50 KiB and three: measured 2^22 x 32 LDE passes 7.66 ms)
    // The persistent prefetching variant (PF) walks tiles in a loop: it keeps wlds for every tile's step 1 and holds the rows and the
    // pre-scale factors in arrays of their own (50 KiB, three workgroups per CU — the occupancy its 132 VGPRs allow anyway).
    __shared__ uint4 vt_rows_own[(VT && !LAST && PF) ? 2 * (1 << LOG_R) : 1];
    __shared__ T vt_ps_own[(VT && !LAST && PF) ? (1 << LOG_R) : 1];
    uint4 *const vt_rows = PF ? vt_rows_own : wlds;
    T *const vt_ps = PF ? vt_ps_own : lds;

    int tid = threadIdx.x;
#if defined(WF_EXPERIMENTS) && defined(NTT_EXTRA_LDS)      // occupancy experiment (tools/build_variant.sh): bytes of LDS a workgroup holds on top of what it needs
    __shared__ uint4 extra_lds[NTT_EXTRA_LDS / 16];
    if (p.log_n == 77u) extra_lds[tid] = make_uint4(1, 2, 3, 4);      // never true: keeps the allocation alive
#endif
#ifdef WF_EXPERIMENTS
    if (!PF && p.stagger_ticks && blockIdx.x < 1024u) {
        // The resident workgroups of a CU all start together and take equally long, so their load, arithmetic and store phases coincide
        // launch after launch of replacements: skew the first generation (mode 0: generation = blockIdx / 256, one workgroup per CU and
        // round of the dispatcher; mode 1: blockIdx % 4) and the replacements inherit the skew
        const uint32_t g = p.stagger_mode == 0 ? (blockIdx.x >> 8) : (blockIdx.x & 3u);
        if (g) {
            const uint64_t t0 = wall_clock64();
            while (wall_clock64() - t0 < (uint64_t)g * p.stagger_ticks) __builtin_amdgcn_s_sleep(16);
        }
    }
#endif
    const uint32_t L = p.log_n;
    const uint64_t n = 1ull << L;
    const uint64_t ncols = n >> LOG_R;                     // columns per vector
    const bool RM = LAST && p.rowmajor;
    const uint32_t log_ncols = L - LOG_R;                  // ncols is a power of two
    // joint (vector, column) space; in row-major mode: [column group][coset u][column c][column-in-group] (last fastest)
    const uint32_t rm_groups = RM ? (p.rm_base_cols + (1u << p.rm_log_i) - 1) >> p.rm_log_i : 0;
    const uint64_t total_cols = RH ? (ncols << (p.rm_log_b + p.rh_log_cp))
                                   : (RM ? ((uint64_t)rm_groups << (p.rm_log_b + log_ncols + p.rm_log_i)) : ncols * (uint64_t)p.nvec);
    // (vector v, column c) of joint index cc; in row-major mode also the base column bc and the coset u, and whether the
    // lane carries a real column (lanes past base_cols in the last group only write padding zeros)
    auto decompose = [&](uint64_t cc, uint64_t &v, uint64_t &c, uint32_t &bc, uint32_t &u) -> bool {
        if constexpr (RH) {
            // [position c][coset u][padded column]: lanes past base_cols carry zeros (the padding of the row)
            bc = (uint32_t)cc & ((1u << p.rh_log_cp) - 1);
            const uint64_t j = cc >> p.rh_log_cp;
            u = (uint32_t)j & ((1u << p.rm_log_b) - 1);
            c = j >> p.rm_log_b;
            v = ((uint64_t)bc << p.rm_log_b) + u;
            return bc < p.rm_base_cols;
        }
        if (!RM) {
            v = cc >> log_ncols;
            c = cc & (ncols - 1);
            bc = u = 0;
            return true;
        }
        const uint32_t ci = (uint32_t)cc & ((1u << p.rm_log_i) - 1);
        const uint64_t r1 = cc >> p.rm_log_i;
        c = r1 & (ncols - 1);
        const uint64_t r2 = r1 >> log_ncols;
        u = (uint32_t)r2 & ((1u << p.rm_log_b) - 1);
        bc = ((uint32_t)(r2 >> p.rm_log_b) << p.rm_log_i) + ci;
        v = ((uint64_t)bc << p.rm_log_b) + u;
        return bc < p.rm_base_cols;
    };
    // log2 of this digit's stride S_p
    uint32_t log_s0 = L;
    for (uint32_t q = 0; q <= p.pass; q++) log_s0 -= p.log_r[q];
    const uint32_t log_mult = L - log_s0 - LOG_R;          // n / n_p = R_1..R_{p-1}
    // In the persistent (PF) variants the stride is re-read through an opaque move in every iteration: otherwise the compiler
    // hoists the 2 x 16 per-register address offsets (and more) out of the tile loop and the kernel needs 250-330 VGPRs
    auto opaque = [](uint32_t v) -> uint32_t {
        if constexpr (PF) asm volatile("" : "+s"(v));
        return v;
    };
    uint32_t log_s = log_s0;

    int b1, t1;
    // (VT: the columns are the contiguous axis of every buffer, so the last pass deals its lanes like a non-last one)
    if (!LAST || VT) { t1 = tid % TC; b1 = tid / TC; } else { b1 = tid % B; t1 = tid / B; }
    // VT non-last pass: tile -> (position column c, column group g of TC columns, coset u); whole tiles only (checked on the host)
    uint64_t vt_c = 0, vtn_c = 0;                // (vtn_*: the prefetched next tile of the persistent variant)
    uint32_t vt_g = 0, vt_u = 0, vtn_g = 0, vtn_u = 0;
    uint64_t vt_tiles = 0;
    if constexpr (VT && !LAST) vt_tiles = p.vt_tiles;
    auto vt_tile = [&](uint64_t tl, uint64_t &vt_c, uint32_t &vt_g, uint32_t &vt_u) {
        const uint32_t b = 1u << p.rm_log_b, gt = p.vt_cols / TC;
        const uint64_t full = (ncols * gt * b) / 64 * 64;
        uint64_t pair;
        if (p.pass == 0 && b > 1 && b <= 8 && tl < full) {
            // the b cosets of one (c, g) source tile on ONE XCD inside a window of 64 blocks: the polynomials are fetched once (see below)
            const uint32_t sl = (uint32_t)tl & 63u, x = sl & 7u, k = sl >> 3;
            vt_u = k & (b - 1);
            pair = (tl >> 6) * (64 / b) + 8 * (k / b) + x;
        } else {
            vt_u = (uint32_t)tl & (b - 1);
            pair = tl >> p.rm_log_b;
        }
        vt_c = pair / gt;
        vt_g = (uint32_t)(pair - vt_c * gt);
    };
    // issue the loads of this lane's A inputs of tile `tile` (no wait); also returns what step 1 needs to know about them
    auto load_inputs = [&](uint64_t tile, T (&xin)[A], bool &active, uint64_t &v, uint64_t &base, bool nxt) {
        if constexpr (VT && !LAST) {
            // element (column bc, coset u, position j) lives at [(u n + j) C + bc]; the first pass reads the polynomials: [j C + bc]
            uint64_t tc;                           // this tile's coordinates: kept for step 2 (vt_*), or for the next iteration (vtn_*)
            uint32_t tg, tu;
            vt_tile(tile, tc, tg, tu);
            if (nxt) {
                vtn_c = tc;
                vtn_g = tg;
                vtn_u = tu;
            } else {
                vt_c = tc;
                vt_g = tg;
                vt_u = tu;
            }
            const uint32_t bc = tg * TC + (uint32_t)t1;
            v = ((uint64_t)bc << p.rm_log_b) + tu;
            const uint64_t rem = tc & ((1ull << log_s) - 1);
            base = ((tc >> log_s) << (log_s + LOG_R)) + rem;
            active = true;
            const T *ptr = p.src + ((p.pass ? ((uint64_t)tu << L) : 0ull) + base + ((uint64_t)b1 << log_s)) * p.vt_cols + bc;
            const uint64_t istep = ((uint64_t)B << log_s) * p.vt_cols;
#pragma unroll
            for (int a = 0; a < A; a++) {
                xin[a] = *ptr;
                ptr += istep;
            }
            return;
        }
        const uint64_t cc = tile * TC + t1;
        uint64_t c = 0;
        uint32_t bc1, u1;
        v = 0;
        active = cc < total_cols && decompose(cc, v, c, bc1, u1);
        if (!LAST) {
            const uint64_t rem = c & ((1ull << log_s) - 1);
            base = ((c >> log_s) << (log_s + LOG_R)) + rem;
        } else {
            base = 0;
            uint64_t cr = c;
            uint32_t ls = L;
            for (uint32_t q = 0; q + 1 < p.npass; q++) {
                ls -= p.log_r[q];
                base += (cr & ((1ull << p.log_r[q]) - 1)) << ls;
                cr >>= p.log_r[q];
            }
        }
        if constexpr (VT) {
            // last pass on the column-interleaved ping buffer: [(u n + position) C + bc], positions base + b1 + B a
            if (active) {
                // block-tile plan (p.bt): column index cc = k0 + R0 k1; the lane's inputs are a2 = b1 + B a of [(k1 S1 + a2) R0 + k0]
                const uint64_t pos0 = p.bt ? ((c >> p.log_r[0]) << (L - p.log_r[0] - p.log_r[1])) : (((uint64_t)u1 << L) + base);
                const uint32_t bcx = p.bt ? ((uint32_t)c & ((1u << p.log_r[0]) - 1)) : bc1;
                const T *ptr = p.src + (pos0 + (uint64_t)b1) * p.vt_cols + bcx;
                const uint64_t istep = (uint64_t)B * p.vt_cols;
#pragma unroll
                for (int a = 0; a < A; a++) {
                    xin[a] = *ptr;
                    ptr += istep;
                }
            } else {
#pragma unroll
                for (int a = 0; a < A; a++) xin[a] = F::zero();
            }
            return;
        }
        uint32_t vs, vq, vr;
        divmod_uniform((uint32_t)v, p.src_div, vs, vr);
        divmod_uniform(vs, p.src_inner, vq, vr);
        const T *src = p.src + (uint64_t)vq * p.src_vec_stride + (uint64_t)vr * p.src_inner_stride;
        if (active) {
            // one 64-bit address per lane; the lane's A inputs follow at a wave-uniform stride (scalar arithmetic), so an element
            // costs one address instruction instead of a 64-bit multiply by the element stride (ten)
            const T *ptr = src + (base + ((uint64_t)b1 << log_s)) * p.src_es;
            const uint64_t istep = ((uint64_t)B << log_s) * p.src_es;
#pragma unroll
            for (int a = 0; a < A; a++) {
                xin[a] = F::load_norm(*ptr);
                ptr += istep;          // a running pointer: one shift-add with the scalar step (a * istep becomes multiply-adds)
            }
        } else {
#pragma unroll
            for (int a = 0; a < A; a++) xin[a] = F::zero();
        }
    };

    const uint64_t ntiles = (VT && !LAST) ? vt_tiles : (total_cols + TC - 1) / TC;
    uint64_t tile = blockIdx.x;
    if constexpr (!LAST && !PF) {
        // First pass of a coset LDE (src_div = blowup b cosets per input column): the b tiles that read the SAME source tile — cosets
        // u = 0 .. b-1 of (column, column block) — are given to b workgroups that land on ONE XCD (blockIdx mod 8) within a window
        // of 64 consecutive blocks, so the source is fetched from memory once instead of b times (round 4's counters: the first LDE
        // pass fetched 9.7 GB for a 1.1 GB source, the f128 one 34 GB for 4.3).  Linear order: tile = v * tiles_per_vector + cblk with
        // v = column * b + u; here window slot s -> XCD x = s mod 8, k = s / 8 = u + b j, (column, cblk) pair = window * 64 / b + 8 j + x.
        const uint32_t b = p.src_div;
        if (p.pass == 0 && p.coset_order && b > 1 && b <= 8 && (b & (b - 1)) == 0) {
            const uint64_t tpv = ncols / TC;                                  // tiles per vector (whole tiles: checked on the host)
            const uint64_t full = ntiles / 64 * 64;
            uint64_t pair;
            uint32_t u;
            if (tile < full) {
                const uint32_t sl = (uint32_t)tile & 63u, x = sl & 7u, k = sl >> 3;
                u = k & (b - 1);
                pair = (tile >> 6) * (64 / b) + 8 * (k / b) + x;
            } else {                                                          // the last, partial window: the remaining pairs, coset fastest
                const uint64_t rest = tile - full;
                u = (uint32_t)rest & (b - 1);
                pair = full / b + rest / b;
            }
            const uint64_t col = pair / tpv, cblk = pair - col * tpv;
            tile = (col * b + u) * tpv + cblk;
        }
    }
    T x[A];
    bool active;
    uint64_t v1, base1;
    if constexpr (WLDS) {
        const uint4 *rows = reinterpret_cast<const uint4 *>(p.w256);
        // rows at their natural place, the two 16-byte halves of a row next to each other (a [half][k_a][b] layout that is free of
        // bank conflicts for the last pass's lanes measured SLOWER: 73 against 64 us for that pass)
        for (int i = threadIdx.x; i < 2 * WROWS; i += 256) wlds[i] = rows[2 * ((i >> 1) << (8 - LOG_R)) + (i & 1)];
    }
    load_inputs(tile, x, active, v1, base1, false);
    uint64_t vt_w = 0;
    if constexpr (VT && !LAST) {
        // the tile's R inter-pass twiddles omega_n^((k' rem) mult), one per lane, expanded to the four-word rows of the exit (l24.cuh:
        // w T^k mod p as plain integers — what pass_twiddle_table_kernel writes for a whole pass, here for one tile), and, in the first
        // pass of a coset LDE, its R pre-scale factors (offset g^u)^(j of row r): both shared by the tile's TC columns
        const uint32_t r32 = (uint32_t)(vt_c & ((1ull << log_s0) - 1));
        if (tid < (1 << LOG_R)) {
            // (the look-up now, while the tile's loads are in flight; the expansion to a row after step 1, see vt_rows)
            vt_w = gl::to_int(series_at32<F>(p.w_lo, p.w_hi, p.w_log_lo, ((uint32_t)tid * r32) << log_mult));
            if (p.pre_lo != nullptr && p.pass == 0) {
                const T *plo = p.pre_lo + vt_u * p.pre_lo_stride, *phi = p.pre_hi + vt_u * p.pre_hi_stride;
                const uint64_t base_t = ((vt_c >> log_s0) << (log_s0 + LOG_R)) + r32;
                vt_ps[tid] = series_at32<F>(plo, phi, p.pre_log_lo, (uint32_t)(base_t + ((uint64_t)tid << log_s0)));
            }
        }
    }
    if constexpr (WLDS) __syncthreads();
    for (;;) {
    log_s = opaque(log_s0);
    if constexpr (PF) {
        // the same for everything derived from the lane index (LDS addresses, table addresses, b1 / t1 / t2 / q2)
        asm volatile("" : "+v"(tid));
        if (!LAST || VT) { t1 = tid % TC; b1 = tid / TC; } else { b1 = tid % B; t1 = tid / B; }
    }
    const uint64_t cc0 = tile * TC;
    // ---- step 1: A-point DFT over the high half of the pass digit ---------------------------------
    {
        const uint64_t v = v1, base = base1;
        if constexpr (VT && !LAST) {
            if (p.pre_lo != nullptr && p.pass == 0) {
#pragma unroll
                for (int a = 0; a < A; a++) x[a] = F::mul(x[a], vt_ps[b1 + B * a]);       // row r = b1 + B a of the tile
                __syncthreads();       // (wave-uniform branch) every lane has its factors before step 1's outputs overwrite them
            }
        } else
        if (active) {
            if (p.pre_lo != nullptr && p.pass == 0) {
                uint32_t uq, u;
                divmod_uniform((uint32_t)v, p.pre_mod, uq, u);
                const T *plo = p.pre_lo + u * p.pre_lo_stride, *phi = p.pre_hi + u * p.pre_hi_stride;
                // the lane's A inputs sit at j0 + a * (B << log_s): their scale factors base^j are a geometric progression
                // (the tables hold base^j with no extra factor), so two look-ups and a chain replace A look-ups
                T cur = series_at32<F>(plo, phi, p.pre_log_lo, (uint32_t)(base + ((uint64_t)b1 << log_s)));
                const T stp = series_at32<F>(plo, phi, p.pre_log_lo, (uint32_t)B << log_s);
#pragma unroll
                for (int a = 0; a < A; a++) {
                    x[a] = F::mul(x[a], cur);
                    if (a + 1 < A) cur = F::mul(cur, stp);
                }
            }
        }
        if constexpr (F::USE_L24) {
            // f64: the DFT runs on four 24-bit limbs per element (l24.cuh); leaving that representation IS the intra-pass
            // twiddle multiplication: p.w256 holds rows of four words omega_256^e T^k mod p (plain integers; the data stay
            // Montgomery residues because the transform is linear), eight multiply-adds + a fold per element.  The LDS
            // exchange carries 64-bit words that need not be below p ("lazy"): step 2 splits them into limbs again.
            typedef l24::Dft<LOG_A> DA;
            int32_t v[DA::NV];
#pragma unroll
            for (int a = 0; a < A; a++) DA::load(v, a, x[a]);
#ifndef NTT_EXPERIMENT_NO_DFT     // timing experiments only (tools/build_variant.sh): results are wrong without the butterflies
            DA::run(v);
#endif
#pragma unroll
            for (int i = 0; i < A; i++) {
                const int ka = brev(i, LOG_A);
                uint32_t y[4];
#pragma unroll
                for (int q = 0; q < 4; q++) y[q] = DA::limb(v, i, q);
                if constexpr (B > 1) {
                    T val;
                    if (ka != 0 || (LAST && p.scale_in_w256)) {
                        if constexpr (WLDS) {
                            const uint4 wa = wlds[2 * (ka * b1)], wb = wlds[2 * (ka * b1) + 1];
                            val = l24::fold_lazy(l24::mul4(y, (T)wa.x | ((T)wa.y << 32), (T)wa.z | ((T)wa.w << 32), (T)wb.x | ((T)wb.y << 32),
                                                           (T)wb.z | ((T)wb.w << 32)));
                        } else {
                            const T *w = p.w256 + 4 * ((uint32_t)(ka * b1) << (8 - LOG_R));
                            val = l24::fold_lazy(l24::mul4(y, w[0], w[1], w[2], w[3]));
                        }
                    } else {
                        val = l24::fold_lazy(l24::mul4_one(y));
                    }
                    if constexpr (BT0) lds[idx_bt(ka, b1 * TC + t1)] = val;
                    else if (!LAST) lds[idx_nl(ka, b1 * TC + t1)] = val;
                    else lds[idx_l(ka, t1, b1)] = val;
                } else {
                    x[i] = l24::fold(l24::mul4_one(y));
                }
            }
        } else {
        dft_dif<F, LOG_A, true>(x, p.w16);
        if (B > 1) {
            // intra-pass twiddle omega_R^(k_a * b) = omega_256^((k_a * b) << (8 - LOG_R)), then LDS exchange
#pragma unroll
            for (int i = 0; i < A; i++) {
                const int ka = brev(i, LOG_A);
                T val = x[i];
                if (ka != 0 || (LAST && p.scale_in_w256)) val = F::mul_tab(val, p.w256, (uint32_t)(ka * b1) << (8 - LOG_R));
                if (!LAST) lds[idx_nl(ka, b1 * TC + t1)] = val;
                else lds[idx_l(ka, t1, b1)] = val;
            }
        }
        }
    }

    if (B > 1) __syncthreads();
    if constexpr (VT && !LAST) {
        // step 1 is through with wlds: the tile's inter-pass twiddle rows (w T^k mod p, k = 0 .. 3, plain integers) take its place
        if (tid < (1 << LOG_R)) {
            const uint64_t c0 = vt_w, c1 = gl::mul_pow2<24>(c0), c2 = gl::mul_pow2<48>(c0), c3 = gl::mul_pow2<72>(c0);
            vt_rows[2 * tid] = make_uint4((uint32_t)c0, (uint32_t)(c0 >> 32), (uint32_t)c1, (uint32_t)(c1 >> 32));
            vt_rows[2 * tid + 1] = make_uint4((uint32_t)c2, (uint32_t)(c2 >> 32), (uint32_t)c3, (uint32_t)(c3 >> 32));
        }
        __syncthreads();
    }

    // ---- step 2: B-point DFT(s) over the low half, inter-pass twiddle, store -----------------------
    const int t2 = BT0 ? tid / B : ((B > 1) ? tid % TC : t1);      // BT0: k_a fastest across lanes (transposed, coalesced stores)
    const int q2 = BT0 ? tid % B : ((B > 1) ? tid / TC : 0);
    const uint64_t cc = cc0 + t2;
    const uint64_t next_tile = tile + gridDim.x;
    const bool more = PF && next_tile < ntiles;
    T xn[A];
    bool active_n = false;
    uint64_t vn = 0, basen = 0;
    // The prefetching variants run on whole tiles only (cc < total_cols for every lane).  The memory counter of a wavefront
    // retires in order: any load that step 2 issued AFTER the prefetch would make its wait drain the prefetch too.  So the only
    // loads of step 2 — the (base, step) look-ups of the twiddle progressions — are taken first, then the next tile's loads go
    // out and stay in flight through the whole of step 2 (which from here on touches registers and LDS only).
    T pf_cur[G > 0 ? G : 1], pf_stp[G > 0 ? G : 1];
    if constexpr (PF && !LAST && B > 1) {
        uint64_t v0, c0;
        uint32_t bq, uq;
        decompose(cc, v0, c0, bq, uq);
        const uint32_t r32 = (uint32_t)(c0 & ((1ull << log_s) - 1));
#pragma unroll
        for (int g = 0; g < G; g++) {
            pf_cur[g] = series_at32<F>(p.w_lo, p.w_hi, p.w_log_lo, ((uint32_t)(q2 + B * g) * r32) << log_mult);
            pf_stp[g] = series_at32<F>(p.w_lo, p.w_hi, p.w_log_lo, ((uint32_t)A * r32) << log_mult);
        }
    }
    uint64_t vt_w_n = 0;
    if constexpr (PF && VT && !LAST) {
        if (more) {
            // the NEXT tile's twiddle and pre-scale look-ups go out before its data loads (the vm counter retires in order: whoever
            // waits for these does not wait for the sixteen loads behind them); step 1 of this tile is through with vt_ps
            uint64_t nc;
            uint32_t ng, nu;
            vt_tile(next_tile, nc, ng, nu);
            const uint32_t r32n = (uint32_t)(nc & ((1ull << log_s0) - 1));
            if (tid < (1 << LOG_R)) {
                vt_w_n = gl::to_int(series_at32<F>(p.w_lo, p.w_hi, p.w_log_lo, ((uint32_t)tid * r32n) << log_mult));
                if (p.pre_lo != nullptr && p.pass == 0) {
                    const T *plo = p.pre_lo + nu * p.pre_lo_stride, *phi = p.pre_hi + nu * p.pre_hi_stride;
                    const uint64_t base_t = ((nc >> log_s0) << (log_s0 + LOG_R)) + r32n;
                    vt_ps[tid] = series_at32<F>(plo, phi, p.pre_log_lo, (uint32_t)(base_t + ((uint64_t)tid << log_s0)));
                }
            }
        }
    }
    if (more) load_inputs(next_tile, xn, active_n, vn, basen, true);
    if ((VT && !LAST) || cc < total_cols) {
    uint64_t v, c;
    uint32_t bc2, u2;
    bool real_col_;
    if constexpr (VT && !LAST) {
        bc2 = vt_g * TC + (uint32_t)t2;
        u2 = vt_u;
        c = vt_c;
        v = ((uint64_t)bc2 << p.rm_log_b) + u2;
        real_col_ = true;
    } else {
        real_col_ = decompose(cc, v, c, bc2, u2);
    }
    const bool real_col = real_col_;
    uint32_t dq, dr;
    divmod_uniform((uint32_t)v, p.dst_inner, dq, dr);
    T *dst = p.dst + (uint64_t)dq * p.dst_vec_stride + (uint64_t)dr * p.dst_inner_stride;
    const uint64_t rem = c & ((1ull << log_s) - 1);
    const uint64_t base_nl = ((c >> log_s) << (log_s + LOG_R)) + rem;

    // Output addressing: every output of a lane is  kp = kbase + krel  with kbase per lane (the k_a of its B-point DFT, or 0) and
    // krel a compile-time multiple of the unrolled loops, so its address is  o_ptr + krel * o_step  with ONE 64-bit pointer per lane
    // and a wave-uniform step (scalar arithmetic): one address instruction per element.  (The form  dst[(base + (kp << log_s)) *
    // dst_es]  cost a 64-bit multiply per element — two multiply-adds, moves, two shift-adds — and kept sixteen pointers live.)
    T *o_ptr = nullptr, *o_ptr0 = nullptr;      // o_ptr0: the krel = 0 output (differs for the inverse transform's k = 0 wrap)
    int64_t o_step = 0;                         // elements of T per unit of krel (negative when the output index runs backwards)
    const T *tw0 = nullptr;                     // TWTAB: the table entry of krel = 0 for this lane's column
    uint32_t o_k32 = 0, o_kstep32 = 0;          // natural output index of krel = 0 and its step (last pass: post-scale look-ups)
    auto set_out_base = [&](uint32_t kbase) {
        if constexpr (VT && !LAST) {
            // the column-interleaved ping buffer: [(u n + position) C + bc]
            o_ptr = o_ptr0 = p.dst + (((uint64_t)u2 << L) + base_nl + ((uint64_t)kbase << log_s)) * p.vt_cols + bc2;
            o_step = (int64_t)(((uint64_t)p.vt_cols) << log_s);
        } else if constexpr (BT0) {
            // transposed: output digit k0 of column rem0 (= c: the first pass has one block) at [rem0 R0 + k0]
            o_ptr = o_ptr0 = p.dst + ((c << LOG_R) + kbase);
            o_step = 1;
        } else if constexpr (!LAST) {
            o_ptr = o_ptr0 = dst + (base_nl + ((uint64_t)kbase << log_s)) * p.dst_es;
            o_step = (int64_t)(((uint64_t)p.dst_es) << log_s);
            if constexpr (TWTAB) tw0 = p.tw_tab + (F::USE_L24 && NTT_F64_TW_ROWS != 0 ? 4 : (p.tw_pair ? F::TAB_WORDS : 1)) * ((((uint64_t)kbase) << log_s) + (uint32_t)rem);
        } else if (RM || (RH && p.rh_log_cp > 3)) {
            // LDE row u + b * m, column bc
            o_ptr = o_ptr0 = p.dst + (u2 + ((c + ncols * (uint64_t)kbase) << p.rm_log_b)) * p.rm_row_width + bc2;
            o_step = (int64_t)((ncols << p.rm_log_b) * p.rm_row_width);
        } else {
            const uint64_t k0 = c + ncols * (uint64_t)kbase;           // natural output index
            if (p.inverse) {
                o_ptr = dst + (n - k0) * p.dst_es;                      // k = n - k0 - krel ncols: never wraps for krel > 0
                o_ptr0 = dst + ((n - k0) & (n - 1)) * p.dst_es;
                o_step = -(int64_t)(ncols * (uint64_t)p.dst_es);
                o_k32 = (uint32_t)(n - k0);
                o_kstep32 = 0u - (uint32_t)ncols;
            } else {
                o_ptr = o_ptr0 = dst + k0 * p.dst_es;
                o_step = (int64_t)(ncols * (uint64_t)p.dst_es);
                o_k32 = (uint32_t)k0;
                o_kstep32 = (uint32_t)ncols;
            }
        }
    };
    // the unrolled loops walk krel = 0, s, 2 s, ... in order: a running pointer, one shift-add with a scalar operand per output
    auto out_next = [&](bool first, int64_t stride) -> T * {
        if (first) return o_ptr0;
        o_ptr += stride;
        return o_ptr;
    };
    auto emit = [&](T val, uint32_t kp, uint32_t krel, int64_t stride) {
        if constexpr (RH) {
            // staged: [output digit k'][tile column], rows are taken from here below
            const uint32_t log_cp = p.rh_log_cp;
            if (log_cp <= 3) {
                lds[kp * TC + t2] = val;    // rows of one 8-column group: stored by the lanes that hash them
                return;
            }
            // wide rows (16 / 32 padded columns, 128 / 256 bytes): the lanes of step 2 store the row segment themselves (consecutive
            // lanes = consecutive columns), the staged copy is only read by the hashing lanes: word c of row r sits at (c + r) mod cp2
            // of the row, so that lanes walking their own rows hit different banks
            const uint32_t cp2m = (1u << log_cp) - 1;
            const uint32_t rowid = (kp << ((8 - LOG_B) - log_cp)) + ((uint32_t)t2 >> log_cp);
            lds[kp * TC + ((uint32_t)t2 & ~cp2m) + ((((uint32_t)t2 & cp2m) + rowid) & cp2m)] = val;
            T *cell = out_next(krel == 0, stride);
            if (real_col) *cell = val;
            else if (bc2 < p.rm_row_width) *cell = F::zero();
            return;
        }
        if (RM) {
            // the last group's lanes also zero the padding columns of that row
            T *cell = out_next(krel == 0, stride);
            if (real_col) *cell = val;
            if ((bc2 >> p.rm_log_i) + 1 == rm_groups) {
                T *row = cell - bc2;
                for (uint64_t pc = p.rm_base_cols + (bc2 & ((1u << p.rm_log_i) - 1)); pc < p.rm_row_width; pc += 1u << p.rm_log_i) row[pc] = F::zero();
            }
        } else {
            if (p.post_lo != nullptr) {
                const uint32_t k = (o_k32 + krel * o_kstep32) & (uint32_t)(n - 1);
                val = F::mul(val, series_at32<F>(p.post_lo, p.post_hi, p.post_log_lo, k));
            } else if (p.has_post_const && !p.scale_in_w256) val = F::mul(val, p.post_const);
            *out_next(krel == 0, stride) = val;
        }
    };

    // Non-last passes: the inter-pass twiddles of one lane's outputs k' = k0 + STEP * i form a geometric progression
    // omega^((k0 + STEP i) * rem * mult) = base * step^i, so two table look-ups (base, step) and a multiplication chain
    // replace a two-level look-up per output: 2 + (CNT - 1) + CNT multiplications instead of 2 (CNT - 1), but 4 loads
    // instead of 2 CNT and none of the per-output index arithmetic.
    // `val(i)` yields the value in register i (for f64 it leaves the limb representation on demand, one element at a time)
    auto emit_progression = [&](auto &&val, uint32_t k0, uint32_t step_k, auto log_cnt_tag, int g = 0) {
        constexpr int LOG_CNT = decltype(log_cnt_tag)::value;
        constexpr int CNT = 1 << LOG_CNT;
        const uint32_t r32 = (uint32_t)rem;
        set_out_base(k0);
        if constexpr (TWTAB) {
            // small strides: the pass's 2^(LOG_R + log_s) twiddles sit in one L2-resident table, rows of 2^log_s consecutive
            // `rem` (a tile's 16 columns = one 128-byte run): 16 coalesced loads replace the 15-multiplication chain
#pragma unroll
            for (int ip = 0; ip < CNT; ip++) {
                const int i = brev(ip, LOG_CNT);                 // register holding output digit k0 + step_k * ip
                const uint64_t ti = ((uint64_t)(step_k * (uint32_t)ip)) << log_s;
                // small tables are in the field's table layout (f128: pairs, F::mul_tab), the big shared ones one word per entry
                *out_next(ip == 0, (int64_t)step_k * o_step) = (F::TAB_WORDS > 1 && p.tw_pair) ? F::mul_tab(val(i), tw0, ti) : F::mul(val(i), tw0[ti]);
            }
            return;
        }
        T cur, stp;
        if constexpr (PF && B > 1) {
            cur = pf_cur[g];
            stp = pf_stp[g];
        } else {
            cur = series_at32<F>(p.w_lo, p.w_hi, p.w_log_lo, (k0 * r32) << log_mult);
            stp = series_at32<F>(p.w_lo, p.w_hi, p.w_log_lo, (step_k * r32) << log_mult);
        }
#pragma unroll
        for (int ip = 0; ip < CNT; ip++) {
            const int i = brev(ip, LOG_CNT);                 // register holding output digit k0 + step_k * ip
#ifdef NTT_EXPERIMENT_NO_CHAIN    // timing experiments only
            *out_next(ip == 0, (int64_t)step_k * o_step) = val(i) ^ cur ^ stp;
#else
            *out_next(ip == 0, (int64_t)step_k * o_step) = F::mul(val(i), cur);
            if (ip + 1 < CNT) cur = F::mul(cur, stp);
#endif
        }
    };

    if constexpr (B == 1) {
        if constexpr (!LAST) {
            emit_progression([&](int i) { return x[i]; }, 0u, 1u, std::integral_constant<int, LOG_A>{});
        } else {
            set_out_base(0u);
#pragma unroll
            for (int ip = 0; ip < A; ip++) emit(x[brev(ip, LOG_A)], (uint32_t)ip, (uint32_t)ip, o_step);
        }
    } else {
        // RH: the exchange buffer becomes the row staging area, so every lane takes ALL its inputs (both B-point DFTs of a radix-128
        // pass) before the first output is staged
        T yall[RH ? G : 1][B];
        if constexpr (RH) {
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int bb = 0; bb < B; bb++) yall[g][bb] = lds[idx_l(q2 + B * g, t2, bb)];
            __syncthreads();
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int ka = q2 + B * g;
            T y[B];
#pragma unroll
            for (int bb = 0; bb < B; bb++) {
                if constexpr (RH) y[bb] = yall[g][bb];
                else if constexpr (BT0) y[bb] = lds[idx_bt(ka, bb * TC + t2)];
                else if (!LAST) y[bb] = lds[idx_nl(ka, bb * TC + t2)];
                else y[bb] = lds[idx_l(ka, t2, bb)];
            }
            if constexpr (F::USE_L24) {
                typedef l24::Dft<LOG_B> DB;
                int32_t v[DB::NV];
#pragma unroll
                for (int bb = 0; bb < B; bb++) DB::load(v, bb, y[bb]);
#ifndef NTT_EXPERIMENT_NO_DFT
                DB::run(v);
#endif
                if constexpr (VT && !LAST) {
                    // the tile's twiddle rows (LDS, generated above): the exit from the limb form IS the inter-pass multiplication
                    set_out_base((uint32_t)ka);
#pragma unroll
                    for (int ip = 0; ip < B; ip++) {
                        const int i = brev(ip, LOG_B);           // output digit k_a + A * ip
                        uint32_t yl[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) yl[q] = DB::limb(v, i, q);
                        const uint4 wa = vt_rows[2 * (ka + A * ip)], wb = vt_rows[2 * (ka + A * ip) + 1];
                        *out_next(ip == 0, (int64_t)A * o_step) = l24::fold_lazy(l24::mul4(yl, (T)wa.x | ((T)wa.y << 32), (T)wa.z | ((T)wa.w << 32),
                                                                                           (T)wb.x | ((T)wb.y << 32), (T)wb.z | ((T)wb.w << 32)));
                    }
                } else if constexpr (TWTAB && NTT_F64_TW_ROWS != 0) {
                    // inter-pass twiddles from the L2-resident table, kept in the same four-word form: multiply, fold, store
                    set_out_base((uint32_t)ka);
#pragma unroll
                    for (int ip = 0; ip < B; ip++) {
                        const int i = brev(ip, LOG_B);           // output digit k_a + A * ip
                        uint32_t yl[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) yl[q] = DB::limb(v, i, q);
                        const T *w = tw0 + 4 * (((uint64_t)A * (uint32_t)ip) << log_s);
                        *out_next(ip == 0, (int64_t)A * o_step) = l24::fold(l24::mul4(yl, w[0], w[1], w[2], w[3]));
                    }
                } else {
                    // leave the limb form one element at a time, right where the value is consumed.  The last pass stores these
                    // words (or multiplies them by a canonical factor): below p.  The progression multiplies them by canonical
                    // twiddles, and the Montgomery reduction accepts any 64-bit word: lazy is enough.
                    auto out = [&](int i) -> T {
                        uint32_t yl[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) yl[q] = DB::limb(v, i, q);
                        return LAST ? l24::fold(l24::mul4_one(yl)) : l24::fold_lazy(l24::mul4_one(yl));
                    };
                    if constexpr (!LAST) {
                        emit_progression(out, (uint32_t)ka, (uint32_t)A, std::integral_constant<int, LOG_B>{}, g);
                    } else {
                        set_out_base((uint32_t)ka);
#pragma unroll
                        for (int ip = 0; ip < B; ip++) emit(out(brev(ip, LOG_B)), (uint32_t)(ka + A * ip), (uint32_t)(A * ip), (int64_t)A * o_step);
                    }
                }
            } else {
                dft_dif<F, LOG_B, true>(y, p.w16);
                if constexpr (!LAST) {
                    emit_progression([&](int i) { return y[i]; }, (uint32_t)ka, (uint32_t)A, std::integral_constant<int, LOG_B>{});
                } else {
                    set_out_base((uint32_t)ka);
#pragma unroll
                    for (int ip = 0; ip < B; ip++) emit(y[brev(ip, LOG_B)], (uint32_t)(ka + A * ip), (uint32_t)(A * ip), (int64_t)A * o_step);
                }
            }
        }
    }
    }   // cc < total_cols
    if constexpr (RH) {
        // whole tiles only (checked on the host): every lane went through step 2 and its barrier
        __syncthreads();
        constexpr int LOG_TC = 8 - LOG_B;
        const uint32_t log_cp = p.rh_log_cp, cp2 = 1u << log_cp;
        const uint32_t log_rpk = LOG_TC - log_cp;                        // rows per output digit
        const uint32_t nrows = (1u << LOG_R) << log_rpk;
        const uint64_t j0 = cc0 >> log_cp, kp_stride = ncols << p.rm_log_b;
        if (log_cp > 3) {
            // wide rows: one BLAKE3 chunk of up to four blocks per row, message words straight from the staged tile
            const uint32_t cp2m = cp2 - 1, base_cols = p.rm_base_cols;
            for (uint32_t r = (uint32_t)tid; r < nrows; r += 256) {
                const uint32_t kp = r >> log_rpk, jl = r & ((1u << log_rpk) - 1);
                const uint64_t row = j0 + jl + kp_stride * kp;
                const T *rw = lds + kp * TC + (jl << log_cp);
                auto fetch = [&](uint32_t blk, uint32_t, uint32_t (&m)[16]) {
#pragma unroll
                    for (uint32_t i = 0; i < 8; i++) {
                        const uint32_t col = 8 * blk + i;
                        const uint64_t vi = col < base_cols ? gl::to_int((uint64_t)rw[(col + r) & cp2m]) : 0ull;
                        m[2 * i] = (uint32_t)vi;
                        m[2 * i + 1] = (uint32_t)(vi >> 32);
                    }
                };
                uint32_t d[8];
                b3::chunk_blocks(fetch, 0, base_cols * 2, 0, true, d);
                uint4 *q = reinterpret_cast<uint4 *>(p.rh_leaves) + row * 2;
                q[0] = make_uint4(d[0], d[1], d[2], d[3]);
                q[1] = make_uint4(d[4], d[5], d[6], d[7]);
            }
        } else
        for (uint32_t r = (uint32_t)tid; r < nrows; r += 256) {
            const uint32_t kp = r >> log_rpk, jl = r & ((1u << log_rpk) - 1);
            const uint64_t row = j0 + jl + kp_stride * kp;
            uint64_t w[8];
#pragma unroll
            for (int c = 0; c < 8; c++) w[c] = (uint32_t)c < cp2 ? (uint64_t)lds[kp * TC + (jl << log_cp) + c] : 0ull;
            // leaf = Blake3_256::hash_elements(row) over the canonical little-endian bytes of the real columns: one block
            uint32_t cv[8], m[16], d[8];
#pragma unroll
            for (int i = 0; i < 8; i++) cv[i] = b3::iv(i);
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const uint64_t vi = gl::to_int(w[c]);
                m[2 * c] = (uint32_t)vi;
                m[2 * c + 1] = (uint32_t)(vi >> 32);
            }
            b3::compress(cv, m, 0, p.rm_base_cols * 8, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT, d);
            uint4 *q = reinterpret_cast<uint4 *>(p.rh_leaves) + row * 2;
            q[0] = make_uint4(d[0], d[1], d[2], d[3]);
            q[1] = make_uint4(d[4], d[5], d[6], d[7]);
            uint4 *o = reinterpret_cast<uint4 *>(p.dst + row * 8);
#pragma unroll
            for (int i = 0; i < 4; i++)
                o[i] = make_uint4((uint32_t)w[2 * i], (uint32_t)(w[2 * i] >> 32), (uint32_t)w[2 * i + 1], (uint32_t)(w[2 * i + 1] >> 32));
        }
    }
    if (!more) break;
    if (B > 1) __syncthreads();          // every lane is done reading the exchange buffer before the next tile overwrites it
#pragma unroll
    for (int a = 0; a < A; a++) x[a] = xn[a];
    active = active_n;
    v1 = vn;
    base1 = basen;
    tile = next_tile;
    if constexpr (PF && VT && !LAST) {
        vt_c = vtn_c;
        vt_g = vtn_g;
        vt_u = vtn_u;
        vt_w = vt_w_n;
    }
    }   // tiles
}

// the prefetching (persistent) variants exist for the radices whose tiles are big enough to be latency bound, f64 only; they lost in
// round 2 (see PF above) and are compiled only into experiment builds (-DWF_EXPERIMENTS, tools/build_variant.sh)
template <class F, int LA, int LB>
constexpr bool has_prefetch_variant() {
#ifdef WF_EXPERIMENTS
    return F::USE_L24 && LA + LB >= 6;
#else
    return false;
#endif
}

template <class F, int LA, int LB>
constexpr bool has_rows_variant() { return F::USE_L24 && (LA == LB || LA == LB + 1) && LB >= 3; }

template <class F, int LA, int LB>
auto pick(bool last, bool twtab, bool pf, bool rh = false, int vt = 0) -> void (*)(PassParams<typename F::T>) {
    typedef void (*fn)(PassParams<typename F::T>);
    if constexpr (has_rows_variant<F, LA, LB>()) {
#ifndef WF_NO_VECTOR_TILES
        if (vt == 1 && rh && last) return (fn)ntt_pass<F, LA, LB, true, false, false, true, 1>;
        if (vt == 1 && last) return (fn)ntt_pass<F, LA, LB, true, false, false, false, 1>;
#ifdef WF_EXPERIMENTS      // the persistent prefetching variant of the vector-tile passes: measured slower (see ntt_run), experiment builds only
        if (vt == 1 && pf) return (fn)ntt_pass<F, LA, LB, false, false, true, false, 1>;
#endif
        if (vt == 1) return (fn)ntt_pass<F, LA, LB, false, false, false, false, 1>;
        if (vt == 2 && !last) return (fn)ntt_pass<F, LA, LB, false, false, false, false, 2>;
#endif
        if (rh && last) return (fn)ntt_pass<F, LA, LB, true, false, false, true>;
    }
    if constexpr (has_prefetch_variant<F, LA, LB>()) {
        if (pf && !twtab) return last ? (fn)ntt_pass<F, LA, LB, true, false, true> : (fn)ntt_pass<F, LA, LB, false, false, true>;
    }
    if (last) return (fn)ntt_pass<F, LA, LB, true, false, false>;
    return twtab ? (fn)ntt_pass<F, LA, LB, false, true, false> : (fn)ntt_pass<F, LA, LB, false, false, false>;
}

template <class F>
auto kernel_for(uint32_t r, bool last, bool twtab, bool pf, bool rh = false, int vt = 0) -> void (*)(PassParams<typename F::T>) {
    switch (r) {
        case 1: return pick<F, 1, 0>(last, twtab, pf);
        case 2: return pick<F, 1, 1>(last, twtab, pf);
        case 3: return pick<F, 2, 1>(last, twtab, pf);
        case 4: return pick<F, 2, 2>(last, twtab, pf);
        case 5: return pick<F, 3, 2>(last, twtab, pf);
        case 6: return pick<F, 3, 3>(last, twtab, pf, rh, vt);
        case 7: return pick<F, 4, 3>(last, twtab, pf, rh, vt);
        default: return pick<F, 4, 4>(last, twtab, pf, rh, vt);
    }
}

template <class F>
static bool prefetch_variant_exists(uint32_t r) {
#ifdef WF_EXPERIMENTS
    return F::USE_L24 && r >= 6;
#else
    return false;
#endif
}

inline uint32_t log_b_for(uint32_t r) { return r == 1 ? 0 : r / 2; }

}  // namespace

// Split L bits into passes of at most max_bits bits, as evenly as possible (largest first).  max_bits is 8 for the
// 64-bit fields; f128 uses 6: a radix-256 pass of 16-byte elements needs 207 VGPRs and 70 KB of LDS per workgroup
// (2 waves per SIMD), a radix-64 pass 116 VGPRs and 35 KB (4 waves per SIMD), which more than pays for the extra pass.
#ifndef NTT_RH32_PLAN
#define NTT_RH32_PLAN 1
#endif
static inline void plan_passes(uint32_t L, uint32_t max_bits, uint32_t &npass, uint32_t log_r[6], bool rows32 = false) {
    npass = (L + max_bits - 1) / max_bits;
    if (npass == 0) npass = 1;
    if (NTT_RH32_PLAN && rows32 && L == 22 && max_bits == 8) {
        // rows + leaves for rows of 17 .. 32 columns: the radix-64 last pass (68 VGPRs, 18 KiB of LDS: many workgroups per CU to
        // run beside the lanes that hash) instead of the radix-128 one of the 7, 8, 7 plan
        npass = 3;
        log_r[0] = 8;
        log_r[1] = 8;
        log_r[2] = 6;
        for (uint32_t q = 3; q < 6; q++) log_r[q] = 0;
        return;
    }
    if (npass == 3 && max_bits == 8 && L >= 18) {
        // three radix <= 256 passes: the widest in the middle, the rest split evenly with the first pass the larger one — measured over
        // every split of 2^18 .. 2^23 points (tools/time_batch_ntt.py with WF_NTT_PLAN): 2^20: 6,8,6 417 us against 7,7,6 436 and
        // 7,6,7 474 for 32 vectors; 2^19: 6,7,6 403 against 7,6,6 426; 2^23: 8,8,7 437 against 7,8,8 449
        const uint32_t mid = L - 12 < 8 ? L - 12 : 8, rest = L - mid;
        log_r[0] = (rest + 1) / 2;
        log_r[1] = mid;
        log_r[2] = rest / 2;
        for (uint32_t q = 3; q < 6; q++) log_r[q] = 0;
        return;
    }
    uint32_t rem = L;
    for (uint32_t q = 0; q < npass; q++) {
        uint32_t left = npass - q;
        uint32_t r = (rem + left - 1) / left;
        log_r[q] = r;
        rem -= r;
    }
    for (uint32_t q = npass; q < 6; q++) log_r[q] = 0;
}

// the rows + leaves mode exists for f64 when the last pass of the plan has radix 64 .. 256, a (padded) row fits the tile's columns
// (32 for radix 64 / 128, 16 for radix 256) and tiles are whole
template <class F>
static bool rows_mode_ok(uint32_t L, uint32_t log_b, uint32_t base_cols) {
    if (!F::USE_L24 || base_cols == 0 || base_cols > 32) return false;
    uint32_t log_cp = 0;
    while ((1u << log_cp) < base_cols) log_cp++;
    uint32_t npass, log_r[6];
    plan_passes(L, F::MAX_LOG_RADIX, npass, log_r, log_cp == 5);
    const uint32_t r = log_r[npass - 1];
    if (r < 6 || r > 8) return false;
    // rows of 17 .. 32 columns of 2^20 points and more: only beside a radix-64 last pass (2^20 .. 2^22-point columns).  With the radix-128
    // last pass of 2^23-point columns (plan 8, 8, 7) the fused pass loses to the row-major store + row-hash kernel by 2 .. 4 % (round-6
    // sweep; round 5 had measured the same for a 7, 8, 7 plan at 2^22)
    if (log_cp == 5 && r != 6 && L >= 20) return false;
    const uint32_t log_tc = 8 - r / 2;
    return (L - r) + log_b + log_cp >= log_tc && log_cp <= log_tc;
}

namespace {
// columns -> column-interleaved: out[pos * C + bc] = column bc at position pos (column bc of a ColMatrix of extension degree D lives
// at (bc / D) * vec_stride + (bc % D) * inner_stride, element stride es).  A workgroup moves 16 columns x 64 positions through LDS: reads
// are 512-byte runs along a column, writes 128-byte runs along a row.  The source of every VT pass chain (ntt_pass<..., VT>).
template <class T>
__global__ __launch_bounds__(256) void vt_interleave_kernel(const T *src, T *out, uint32_t C, uint64_t n, uint32_t inner, uint64_t vec_stride,
                                                            uint64_t inner_stride, uint32_t es) {
    __shared__ T tile[16][65];
    const uint32_t cg = blockIdx.x % (C / 16);
    const uint64_t p0 = (uint64_t)(blockIdx.x / (C / 16)) * 64;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t idx = threadIdx.x + 256 * k, cl = idx >> 6, pl = idx & 63;
        const uint32_t bc = cg * 16 + cl;
        uint32_t vq, vr;
        divmod_uniform(bc, inner, vq, vr);
        const uint64_t pos = p0 + pl;
        tile[cl][pl] = pos < n ? src[(uint64_t)vq * vec_stride + (uint64_t)vr * inner_stride + pos * es] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t idx = threadIdx.x + 256 * k, pl = idx >> 4, cl = idx & 15;
        const uint64_t pos = p0 + pl;
        if (pos < n) out[pos * C + cg * 16 + cl] = tile[cl][pl];
    }
}
}  // namespace

#ifndef NTT_TW_TABLE_MAX_LOG
#define NTT_TW_TABLE_MAX_LOG 17   // largest inter-pass twiddle table kept (entries): 1 MiB of f64, L2 resident
#endif

namespace {
template <class F>
__global__ __launch_bounds__(256) void pass_twiddle_table_kernel(const typename F::T *w_lo, const typename F::T *w_hi, uint32_t w_log_lo,
                                                                 uint32_t log_s, uint32_t log_mult, uint32_t log_total, typename F::T *out, uint32_t pair,
                                                                 typename F::T two64) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >> log_total) return;
    const uint32_t kp = idx >> log_s, rem = idx & ((1u << log_s) - 1);
    const typename F::T w = series_at32<F>(w_lo, w_hi, w_log_lo, (kp * rem) << log_mult);
    if constexpr (F::USE_L24 && NTT_F64_TW_ROWS != 0) {
        // rows of four plain-integer words w T^k mod p (l24.cuh); the series tables hold Montgomery residues
        const uint64_t c = gl::to_int(w);
        out[4 * (uint64_t)idx + 0] = c;
        out[4 * (uint64_t)idx + 1] = gl::mul_pow2<24>(c);
        out[4 * (uint64_t)idx + 2] = gl::mul_pow2<48>(c);
        out[4 * (uint64_t)idx + 3] = gl::mul_pow2<72>(c);
    } else {
        if (F::TAB_WORDS > 1 && pair) {      // (w, w * 2^64): F::mul_tab
            out[2 * (uint64_t)idx] = w;
            out[2 * (uint64_t)idx + 1] = F::mul(w, two64);
        } else {
            out[idx] = w;
        }
    }
}
}  // namespace

#ifndef NTT_TW_TABLE_BATCH_MAX_LOG
#define NTT_TW_TABLE_BATCH_MAX_LOG 22   // the same for transforms of >= 8 vectors at a time (f128: 64 MiB): the table is shared by all of them
#endif
#ifndef NTT_TW_PAIR_MAX_LOG
#define NTT_TW_PAIR_MAX_LOG 17          // f128: tables of up to 2^17 entries are kept as pairs (w, w 2^64) for F::mul_tab: 4 MiB, L2 resident
#endif
template <class HF>
static int get_pass_twiddles(wf_ctx *ctx, const SeriesTable &om, uint32_t L, uint32_t r, uint32_t log_s, uint32_t log_mult, uint32_t nvec, const void **out,
                             uint32_t *pair_out) {
    typedef typename HF::Dev F;
    typedef typename F::T T;
    *out = nullptr;
    *pair_out = 0;
    const uint32_t log_total = r + log_s;
    // f64 keeps four words per twiddle (l24.cuh): the same byte budget is one entry-doubling earlier; single-step passes
    // (radix 2) have no limb form of the table multiplication
    // A pass over many vectors (the LDE of a wide trace: columns x cosets) shares one table: a generic multiplication by a table entry
    // instead of the progression's two, at 16 bytes of table per element from the L2 / Infinity Cache (f128, configs[3]: the first
    // LDE pass 31.7 -> 30.0 ms — that pass also carries the coset pre-scale, two more products per element — LDE + commit 116.9 ->
    // 113.6 ms; walking the vectors fastest so that a table tile is reused 512 times in a row measured the same)
    const uint32_t lim = nvec >= 8 ? NTT_TW_TABLE_BATCH_MAX_LOG : NTT_TW_TABLE_MAX_LOG;
    const uint32_t max_log = (F::USE_L24 && NTT_F64_TW_ROWS != 0) ? lim - 1 : lim;
    if (NTT_TW_TABLE_MAX_LOG == 0 || log_total > max_log || (F::USE_L24 && log_b_for(r) == 0)) return WF_OK;
    // f64: 32 bytes of table per element through the vector memory pipeline cost what the 15-multiplication chain costs in issue
    // slots (passes of a 2^24 transform: 75.9 against 73.6-76.6 us): the per-lane progression is used throughout unless a build asks
    // for the tables (-DNTT_F64_TW_TABLES); f128 / f62 keep them (-7 % on their passes)
    // f64 (one-word tables, see NTT_F64_TW_ROWS): measured over the round-6 sweep (profiles/r06/plan_sweep.csv, variant f64-tables): -3 .. -5 %
    // for batches of 2^18 / 2^19-point vectors (the pass's table, <= 4 MiB, stays in the XCD's L2), nothing at 2^20 / 2^21, +4 .. +6 % at
    // 2^22 (a 32 MiB table streams from the Infinity Cache beside the data) and for single transforms (+3 .. +6 %: the table is read once
    // per transform, like the data).  WF_NTT_F64_TABLES=0 / 1: never / wherever the generic limits above allow.
    if (F::USE_L24 && (ctx->f64_tw_tables == 0 || (ctx->f64_tw_tables < 0 && !(nvec >= 8 && log_total <= 19)))) return WF_OK;
    const uint32_t pair = (F::TAB_WORDS > 1 && log_total <= NTT_TW_PAIR_MAX_LOG) ? 1u : 0u;      // a function of the key
    auto key = std::make_tuple((int)F::ID, L, r, log_s, log_mult);
    auto it = ctx->pass_twiddles.find(key);
    if (it == ctx->pass_twiddles.end()) {
        void *d;
        WF_TRY(wf_dev_malloc(ctx, &d, (sizeof(T) * ((F::USE_L24 && NTT_F64_TW_ROWS != 0) ? 4 : (pair ? F::TAB_WORDS : 1))) << log_total));
        ctx->owned.push_back(d);
        const uint32_t total = 1u << log_total;
        const T two64 = HF::to_internal(HF::mulmod(HF::from_u64(1ull << 32), HF::from_u64(1ull << 32)));
        hipLaunchKernelGGL(pass_twiddle_table_kernel<F>, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, (const T *)om.d_lo, (const T *)om.d_hi,
                           om.log_lo, log_s, log_mult, log_total, (T *)d, pair, two64);
        WF_HIP(hipGetLastError());
        it = ctx->pass_twiddles.emplace(key, d).first;
    }
    *out = it->second;
    *pair_out = pair;
    return WF_OK;
}

// one three-step pass (ntt_big.cuh): radix 2^10 (tile 1024 x 8), 2^11 (2048 x 4) or 2^12 (4096 x 4, one 1024-lane workgroup per CU)
template <int LB, int LC, int LTC, bool HALF>
static int launch_big_pass_t(wf_ctx *ctx, const PassParams<uint64_t> &p, bool last, uint64_t total_cols) {
    typedef nttbig::Geo<LB, LC, LTC> G;
    constexpr size_t smem = nttbig::smem_bytes<LB, LC, LTC, HALF>();
    const uint64_t blocks = (total_cols + G::TC - 1) / G::TC;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    typedef void (*fn)(PassParams<uint64_t>);
    const fn k = last ? (fn)nttbig::ntt_pass3<LB, LC, LTC, true, HALF> : (fn)nttbig::ntt_pass3<LB, LC, LTC, false, HALF>;
    if (!ctx->big_lds_opt_in.count((const void *)k)) {
        WF_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        ctx->big_lds_opt_in.insert((const void *)k);
    }
    wf_prof_begin(ctx, last ? "ntt_pass3_last" : "ntt_pass3");
    hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(G::NT), smem, ctx->stream, p);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}
template <class HF>      // a template so that only the f64 translation unit instantiates the kernels
static int launch_big_pass(wf_ctx *ctx, const PassParams<uint64_t> &p, uint32_t r, bool last, uint64_t total_cols) {
    switch (r) {
        case 10: return launch_big_pass_t<3, 3, 3, false>(ctx, p, last, total_cols);
        case 11: return launch_big_pass_t<3, 4, 2, true>(ctx, p, last, total_cols);
        case 12: return launch_big_pass_t<4, 4, 2, true>(ctx, p, last, total_cols);
        default: return WF_ERR_UNSUPPORTED;
    }
}

static inline bool vt_pf_enabled(const wf_ctx *ctx) {
#ifdef WF_EXPERIMENTS
    return ctx->vt_prefetch;
#else
    (void)ctx;
    return false;
#endif
}

template <class HF>
static int ntt_run(wf_ctx *ctx, const NttJob &job) {
    typedef typename HF::Dev F;
    typedef typename F::T T;
    const uint32_t L = job.log_n;
    if (L == 0) return WF_ERR_INVALID_ARG;
    if (L > HF::TWO_ADICITY || L > 32) return WF_ERR_DOMAIN_TOO_LARGE;
    PassParams<T> p{};
    p.log_n = L;
    plan_passes(L, F::MAX_LOG_RADIX, p.npass, p.log_r);
    if (ctx->plan_log_n == L && ctx->plan_npass && job.rh_leaves == nullptr) {   // WF_NTT_PLAN (measurements), read once by wf_ctx_create
        bool fits = true;
        for (uint32_t q = 0; q < ctx->plan_npass; q++) fits = fits && ctx->plan_log_r[q] <= F::MAX_LOG_RADIX;
        if (fits) {
            p.npass = ctx->plan_npass;
            for (uint32_t q = 0; q < 6; q++) p.log_r[q] = ctx->plan_log_r[q];
        }
    }
    // Two passes of radix 2^10 .. 2^12 (ntt_big.cuh) instead of three of radix <= 256: a third less traffic, the same number of
    // register steps and general multiplications, one LDS exchange more.  Measured (tools/time_two_pass.py):
    // it wins for single 2^21 / 2^22-point transforms (-14 % / -9 %) and for batches of 2^20-point vectors (radix 1024, tiles of
    // eight columns = 64-byte segments: -8 %); the radix-2048 tiles of four columns (32-byte segments) lose on batches that stream
    // from HBM (+14 % at 2^22 x 288), the radix-4096 pass (one 1024-lane workgroup per CU) loses everywhere (2^24: 229 against
    // 193 us).  WF_NTT_BIG=1 forces the plan for every eligible transform, 0 switches it off.
    // vector tiles (see below) for this job with the plan in p?  Measured (tools/time_vt.py, profiles/r06/vector_tiles_ab.txt): the LDE
    // passes of 2^22 x 32 columns, blowup 8, 8.37 -> 7.36 ms (the call 17.37 -> 16.66 ms), 2^19 x 96 3.12 -> 2.68 ms; what they cannot beat
    // is the two-pass plan where that is the default (2^20 x 64: 7.70 against 8.02 ms), and with a blowup of 2 the interleaving of the
    // polynomials (one read + one write of them) costs more than the shared twiddles save (2^22 x 64, blowup 2: 11.11 against 11.69 ms)
    auto vt_possible = [&]() -> bool {
#ifdef WF_NO_VECTOR_TILES
        return false;
#else
        if (!F::USE_L24) return false;
        const uint32_t C = job.rm_base_cols, b = 1u << job.rm_log_b;
        bool ok = ctx->lde_vt && !job.inverse && (job.rowmajor || job.rh_leaves != nullptr) && job.pre_lo != nullptr && job.pre_mod == b && job.src_div == b &&
                  job.post_lo == nullptr && !job.has_post_const && C >= 16 && job.nvec == C * b && p.npass >= 2 && job.rm_log_b >= 2 && job.rm_log_b <= 6;
        for (uint32_t q = 0; ok && q < p.npass; q++) {
            const uint32_t r = p.log_r[q], tc = 256u >> log_b_for(r);
            ok = r >= 6 && r <= 8 && (q + 1 == p.npass || (C % tc == 0 && ((((uint64_t)1 << (L - r)) * (C / tc)) << job.rm_log_b) < 0x7fffffffull));
        }
        return ok;
#endif
    };
    bool big = false;
    if constexpr (F::USE_L24) {
        const bool plan_forced = ctx->plan_log_n == L && ctx->plan_npass;
        const uint32_t r_last = L / 2, log_tc_last = r_last == 10 ? 3 : 2;
        // where the two-pass plan is the default: the rule of the round-6 sweep (tools/plan_sweep.py -> profiles/r06/plan_sweep.csv; every
        // other (size, batch) cell is within the run-to-run noise of the three-pass plan or loses: 2^22 batches +5 .. +11 %, 2^23 / 2^24 +2 .. +30 %)
        const bool wins = (L == 20 && job.nvec >= 8) || ((L == 21 || L == 22) && job.nvec < 8) || (L == 21 && job.nvec >= 256);
        const bool wanted = ctx->ntt_big == 1 || (ctx->ntt_big < 0 && wins);
        big = wanted && !plan_forced && L >= 20 && L <= 24 && job.rh_leaves == nullptr && (!job.rowmajor || job.rm_log_i >= log_tc_last);
        if (big) {
            p.npass = 2;
            p.log_r[0] = (L + 1) / 2;
            p.log_r[1] = L / 2;
            for (uint32_t q = 2; q < 6; q++) p.log_r[q] = 0;
        }
    }
    p.nvec = job.nvec;
    p.inverse = job.inverse ? 1 : 0;
    SeriesTable om;
    WF_TRY(wf_get_omega_table<HF>(ctx, L, &om));
    p.w_lo = (const T *)om.d_lo;
    p.w_hi = (const T *)om.d_hi;
    p.w_log_lo = om.log_lo;
    void *w256, *w16;
    if constexpr (F::TAB_WORDS > 1) WF_TRY(wf_get_pair_tables<HF>(ctx, HF::from_u64(1), &w256, &w16));
    else WF_TRY(wf_get_small_tables<HF>(ctx, &w256, &w16));
    if constexpr (F::USE_L24) WF_TRY(wf_get_w256_4form<HF>(ctx, HF::from_u64(1), &w256));
    p.w256 = (const T *)w256;
    p.w16 = (const T *)w16;
    p.pre_lo = (const T *)job.pre_lo;
    p.pre_hi = (const T *)job.pre_hi;
    p.pre_log_lo = job.pre_log_lo;
    p.pre_mod = job.pre_mod ? job.pre_mod : 1;
    p.pre_lo_stride = job.pre_lo_stride;
    p.pre_hi_stride = job.pre_hi_stride;
    p.post_lo = (const T *)job.post_lo;
    p.post_hi = (const T *)job.post_hi;
    p.post_log_lo = job.post_log_lo;
    p.has_post_const = job.has_post_const ? 1 : 0;
    memcpy((void *)&p.post_const, (const void *)job.post_const, sizeof(T));
    p.rowmajor = 0;
    p.coset_order = 0;
    p.rm_log_b = job.rm_log_b;
    p.rm_log_i = job.rm_log_i;
    p.rm_base_cols = job.rm_base_cols;
    p.rm_row_width = job.rm_row_width;
    if (job.rowmajor && (job.nvec != (job.rm_base_cols << job.rm_log_b) || job.rm_log_i > 5)) return WF_ERR_INVALID_ARG;
    // rows + leaves mode of the last pass (see ntt_pass): the caller asked wf_ntt_rows_mode_ok first
    const bool rh = job.rh_leaves != nullptr;
    uint32_t rh_log_cp = 0;
    if (rh) {
        if (job.rowmajor || job.rm_row_width != 8 * ((job.rm_base_cols + 7) / 8) || job.nvec != (job.rm_base_cols << job.rm_log_b) ||
            !rows_mode_ok<F>(L, job.rm_log_b, job.rm_base_cols))
            return WF_ERR_INVALID_ARG;
        while ((1u << rh_log_cp) < job.rm_base_cols) rh_log_cp++;
    }
    p.rh_log_cp = rh_log_cp;
    p.rh_leaves = job.rh_leaves;
    if (rh && rh_log_cp == 5) plan_passes(L, F::MAX_LOG_RADIX, p.npass, p.log_r, true);
    // only the radix 2^6 .. 2^8 last passes have a rows + leaves variant: a plan that ends otherwise must not silently drop the leaves
    if (rh && (p.npass == 0 || p.log_r[p.npass - 1] < 6 || p.log_r[p.npass - 1] > 8)) return WF_ERR_INVALID_ARG;
#ifdef WF_EXPERIMENTS
    {
        static const std::pair<uint32_t, uint32_t> stagger = [] {
            const char *e = getenv("WF_NTT_STAGGER");
            uint32_t t = 0, m = 0;
            if (e) sscanf(e, "%u,%u", &t, &m);
            return std::make_pair(t, m);
        }();
        p.stagger_ticks = stagger.first;
        p.stagger_mode = stagger.second;
    }
#endif

    const uint64_t n = 1ull << L;
    // Vector tiles (ntt_pass<..., VT>): the coset LDE of a wide f64 trace — row-major output or rows + leaves — whose column count is a
    // multiple of every non-last pass's tile width runs on column-interleaved buffers with tile-shared twiddles.  WF_LDE_VT=0: off.
    bool vt = false;
    const T *vt_src = nullptr;
    if constexpr (F::USE_L24) {
#ifndef WF_NO_VECTOR_TILES
        const uint32_t C = job.rm_base_cols;
        vt = !big && vt_possible();
        if (vt) {
            void *il;
            WF_TRY(wf_scratch(ctx, 1, (size_t)n * C * sizeof(T), &il));
            const uint64_t blocks = (uint64_t)(C / 16) * ((n + 63) / 64);
            if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
            wf_prof_begin(ctx, "vt_interleave");
            hipLaunchKernelGGL(vt_interleave_kernel<T>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, (const T *)job.src, (T *)il, C, n,
                               job.src_inner ? job.src_inner : 1, job.src_vec_stride, job.src_inner_stride, job.src_es);
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            vt_src = (const T *)il;
            p.vt_cols = C;
        }
#endif
    }
    // Block tiles for a single three-pass f64 transform (ntt_pass, BT0): first pass stores transposed, the middle pass is the vector-tile
    // kernel over C = R0 blocks with its twiddles shared per tile, the last pass reads columns-fastest.  Measured (tools/time_bt.py, three
    // boxes, profiles/r06/block_tiles_ab.txt): the middle pass 55.2 against 59.0 us at 2^24 — a third fewer instructions, but latency
    // bound at four waves per SIMD — while the transposing first pass and the strided last pass each give ~1 us back: 2^24 166.0 against
    // 163.8 us (+1 %), 2^23 95.9-98.1 against 101.1-102.5 us (-3 .. -5 %), nothing below.  Default: 2^23 only.  WF_NTT_BT=0 / 1: never /
    // wherever eligible.
    bool bt = false;
    if constexpr (F::USE_L24) {
#ifndef WF_NO_VECTOR_TILES
        bt = !vt && !big && ctx->ntt_bt != 0 && (ctx->ntt_bt > 0 || L == 23) && job.nvec == 1 && p.npass == 3 && job.src_div <= 1 && job.src_inner <= 1 &&
             job.dst_inner <= 1 && job.src_es == 1 && job.dst_es == 1 && !job.rowmajor && !rh;
        for (uint32_t q = 0; bt && q < 3; q++) {
            const uint32_t r = p.log_r[q], tc = 256u >> log_b_for(r);
            bt = r >= 6 && r <= 8 && (n >> r) % tc == 0 && (q == 0 || (1u << p.log_r[0]) % tc == 0);
        }
#endif
    }
    T *tmp = nullptr;
    if (p.npass > 1) {
        void *t;
        WF_TRY(wf_scratch(ctx, 0, (size_t)n * job.nvec * sizeof(T), &t));
        tmp = (T *)t;
    }
    for (uint32_t q = 0; q < p.npass; q++) {
        const bool first = q == 0, last = q + 1 == p.npass;
        p.pass = q;
        if (vt) {
            // every VT pass addresses its buffers itself: polynomials [position][C] -> ping buffer [coset][position][C] (in place
            // from the second pass on) -> the last pass's row-major rows (+ leaves), as without vector tiles
            const uint32_t r = p.log_r[q];
            p.src = first ? vt_src : tmp;
            p.dst = last ? (T *)job.dst : tmp;
            p.src_div = 1;
            p.src_inner = 1;
            p.dst_inner = 1;
            p.src_vec_stride = p.dst_vec_stride = 0;
            p.src_inner_stride = p.dst_inner_stride = 1;
            p.src_es = p.dst_es = 1;
            p.scale_in_w256 = 0;
            p.w256 = (const T *)w256;
            p.tw_tab = nullptr;
            p.tw_pair = 0;
            p.coset_order = 0;
            p.rowmajor = (last && job.rowmajor) ? 1 : 0;
            const uint32_t Tc = 256u >> log_b_for(r);
            uint64_t blocks;
            const bool rh_pass = rh && last;
            bool pfv = false;
            if (!last) {
                blocks = ((n >> r) * (p.vt_cols / Tc)) << job.rm_log_b;
                p.vt_tiles = blocks;
                // The persistent prefetching variant (the next tile's loads in flight behind this tile's arithmetic; -DWF_EXPERIMENTS builds with
                // WF_VT_PREFETCH=1).  With a third fewer instructions the vector-tile passes are latency bound at four waves per SIMD (a wave
                // idles ~3.5 us per tile on its loads: 3.3 ps per element and pass measured against 2.3 ps of issue time), which is what the
                // prefetch was built to hide — but its 134 VGPRs and 50 KiB of LDS leave three waves per SIMD, and it LOSES: LDE passes of
                // 2^22 x 32 columns 10.25 against 8.26 ms, the block-tile middle pass of a 2^24 transform 61.0 against 55.2 us (round 6,
                // profiles/r06/vector_tiles_prefetch_ab.txt) — the same verdict as round 2's prefetching variant of the plain pass.
                if (vt_pf_enabled(ctx)) {
                    uint32_t resident = 0;
                    WF_TRY(wf_resident_blocks(ctx, (const void *)kernel_for<F>(r, false, false, true, false, 1), &resident));
                    if (resident > 0 && blocks >= 4ull * resident) {
                        pfv = true;
                        blocks = resident;
                    }
                }
            } else {
                uint64_t total_cols;
                if (rh_pass) total_cols = (n >> r) << (job.rm_log_b + rh_log_cp);
                else {
                    const uint64_t groups = (job.rm_base_cols + (1u << job.rm_log_i) - 1) >> job.rm_log_i;
                    total_cols = (groups << (job.rm_log_b + job.rm_log_i)) * (n >> r);
                }
                blocks = (total_cols + Tc - 1) / Tc;
            }
            if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
            auto k = kernel_for<F>(r, last, false, pfv, rh_pass, 1);
            wf_prof_begin(ctx, rh_pass ? "ntt_pass_last_rows_hash" : (last ? "ntt_pass_last" : "ntt_pass"));
            hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, p);
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            continue;
        }
        if (first) {
            p.src = (const T *)job.src;
            p.src_div = job.src_div ? job.src_div : 1;
            p.src_inner = job.src_inner ? job.src_inner : 1;
            p.src_vec_stride = job.src_vec_stride;
            p.src_inner_stride = job.src_inner_stride;
            p.src_es = job.src_es;
        } else {
            p.src = tmp;
            p.src_div = 1;
            p.src_inner = 1;
            p.src_vec_stride = n;
            p.src_inner_stride = 1;
            p.src_es = 1;
        }
        if (last) {
            p.dst = (T *)job.dst;
            p.dst_inner = job.dst_inner ? job.dst_inner : 1;
            p.dst_vec_stride = job.dst_vec_stride;
            p.dst_inner_stride = job.dst_inner_stride;
            p.dst_es = job.dst_es;
        } else {
            p.dst = tmp;
            p.dst_inner = 1;
            p.dst_vec_stride = n;
            p.dst_inner_stride = 1;
            p.dst_es = 1;
        }
        const uint32_t r = p.log_r[q];
        // a constant output scale (the 1/n of interpolate_poly) rides on the last pass's intra-pass twiddles when that
        // pass has any (B > 1): 16 multiplications per lane instead of 15 + 16
        p.scale_in_w256 = 0;
        p.w256 = (const T *)w256;
        if (last && job.has_post_const && log_b_for(r) > 0) {
            void *ws;
            if constexpr (F::USE_L24) WF_TRY(wf_get_w256_4form<HF>(ctx, HF::from_internal(p.post_const), &ws));
            else if constexpr (F::TAB_WORDS > 1) {
                void *unused;
                WF_TRY(wf_get_pair_tables<HF>(ctx, HF::from_internal(p.post_const), &ws, &unused));
            } else WF_TRY(wf_get_scaled_w256<HF>(ctx, HF::from_internal(p.post_const), &ws));
            p.w256 = (const T *)ws;
            p.scale_in_w256 = 1;
        }
        if constexpr (F::USE_L24) {
            if (big) {
                void *bt;
                WF_TRY(wf_get_big_table<HF>(ctx, r, &bt));
                p.big_tab = (const T *)bt;
                p.tw_tab = nullptr;
                p.rowmajor = (last && job.rowmajor) ? 1 : 0;
                uint64_t cols = (n >> r) * (uint64_t)job.nvec;
                if (p.rowmajor) {
                    const uint64_t groups = (job.rm_base_cols + (1u << job.rm_log_i) - 1) >> job.rm_log_i;
                    cols = (groups << (job.rm_log_b + job.rm_log_i)) * (n >> r);
                }
                WF_TRY(launch_big_pass<HF>(ctx, p, r, last, cols));
                continue;
            }
        }
        if (bt) {
            const uint32_t Tc = 256u >> log_b_for(r), R0 = 1u << p.log_r[0];
            p.rowmajor = 0;
            p.coset_order = 0;
            p.tw_tab = nullptr;
            p.tw_pair = 0;
            p.bt = last ? 1 : 0;
            p.vt_cols = q == 0 ? 0 : R0;
            uint64_t blocks = (n >> r) / Tc;                                              // first and last pass: the standard tiling
            bool pfv = false;
            if (q == 1) {
                p.src = tmp;                                                               // in place on the transposed buffer
                p.dst = tmp;
                blocks = (n >> (p.log_r[0] + r)) * (R0 / Tc);                              // a2 positions x groups of Tc blocks
                p.vt_tiles = blocks;
                if (vt_pf_enabled(ctx)) {
                    uint32_t resident = 0;
                    WF_TRY(wf_resident_blocks(ctx, (const void *)kernel_for<F>(r, false, false, true, false, 1), &resident));
                    if (resident > 0 && blocks >= 4ull * resident) {
                        pfv = true;
                        blocks = resident;
                    }
                }
            }
            if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
            auto k = kernel_for<F>(r, last, false, pfv, false, q == 0 ? 2 : 1);
            wf_prof_begin(ctx, last ? "ntt_pass_last" : "ntt_pass");
            hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, p);
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            continue;
        }
        const uint32_t Tc = 256u >> log_b_for(r);
        uint64_t total_cols = (n >> r) * (uint64_t)job.nvec;
        p.rowmajor = (last && job.rowmajor) ? 1 : 0;
        if (p.rowmajor) {
            const uint64_t groups = (job.rm_base_cols + (1u << job.rm_log_i) - 1) >> job.rm_log_i;
            total_cols = (groups << (job.rm_log_b + job.rm_log_i)) * (n >> r);
        }
        const bool rh_pass = rh && last;
        if (rh_pass) total_cols = (n >> r) << (job.rm_log_b + rh_log_cp);
        // coset-sharing tile order of the first pass of a coset LDE (see ntt_pass): whole tiles per vector only
        p.coset_order = (ctx->coset_order && first && !last && job.src_div > 1 && job.src_div <= 8 && (n >> r) % Tc == 0 &&
                         job.nvec % job.src_div == 0) ? 1 : 0;
        const uint64_t blocks = (total_cols + Tc - 1) / Tc;
        if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
        // Persistent prefetching launch (see ntt_pass): whole tiles only, and enough of them that every resident workgroup gets
        // several — otherwise one workgroup per tile as before.  Such a pass takes its inter-pass twiddles from the per-lane
        // progression, never from the table (whose loads in step 2 would drain the prefetch).
        uint32_t resident = 0;
        bool pf = false;
        if (ctx->ntt_prefetch && !rh_pass && prefetch_variant_exists<F>(r) && total_cols % Tc == 0) {
            auto kp = kernel_for<F>(r, last, false, true);
            WF_TRY(wf_resident_blocks(ctx, (const void *)kp, &resident));
            pf = resident > 0 && blocks >= 4ull * resident;
        }
        p.tw_tab = nullptr;
        p.tw_pair = 0;
        if (!last && !pf) {
            uint32_t log_s = L;
            for (uint32_t qq = 0; qq <= q; qq++) log_s -= p.log_r[qq];
            const void *tab;
            WF_TRY(get_pass_twiddles<HF>(ctx, om, L, r, log_s, L - log_s - r, job.nvec, &tab, &p.tw_pair));
            p.tw_tab = (const T *)tab;
        }
        auto k = kernel_for<F>(r, last, p.tw_tab != nullptr, pf, rh_pass);
        wf_prof_begin(ctx, rh_pass ? "ntt_pass_last_rows_hash" : (last ? "ntt_pass_last" : "ntt_pass"));
        hipLaunchKernelGGL(k, dim3(pf ? resident : (uint32_t)blocks), dim3(256), 0, ctx->stream, p);
        wf_prof_end(ctx);
        WF_HIP(hipGetLastError());
    }
    return WF_OK;
}
