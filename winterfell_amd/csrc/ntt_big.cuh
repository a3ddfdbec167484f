// Three-step NTT passes of radix 2^10 .. 2^12 for the f64 field: a transform of 2^20 .. 2^24 points in TWO passes over memory
// instead of three (same results as math::fft, math/src/fft/mod.rs:85-386 — any DFT algorithm over an exact field does).
//
// Why: the batched transforms behind RowMatrix::evaluate_polys_over (prover/src/matrix/row_matrix.rs:84-100)
// stream from HBM, and the radix-256 plan of ntt_engine.cuh reads and writes the data three times.  Here a workgroup owns a tile of
// R = 2^(4 + LOG_B + LOG_C) "rows" (the pass digit) x TC columns — 64 KiB of LDS for 2048 x 4 — and runs the R-point DFT in three
// register steps A = 16, B, C with two LDS exchanges:
//   step 1  lane (t, j_bc): 16-point DFT over the top digit j_a; leaves the limb form (l24.cuh), multiplies by omega_R^(k_a j_bc)
//           — the one twiddle layer of a pass with R distinct values: an 8-byte table in LDS and a Montgomery product —, LDS
//   step 2  unit (k_a, j_c, t): B-point DFT over j_b; leaving the limb form IS the multiplication by omega_BC^(k_b j_c) (four-word
//           rows, B C of them, in LDS); written back IN PLACE (a unit owns its B slots: no barrier between read and write)
//   step 3  unit (k_a, k_b, t): C-point DFT over j_c; inter-pass twiddles (per-lane geometric progression) or the final store
// Output digit k = k_a + 16 k_b + 16 B k_c.  Everything around the DFT (vector / column addressing, coset pre-scale, inverse index
// negation, output scaling, the row-major LDE store) follows ntt_pass of ntt_engine.cuh.
//
// LDS layout: element (q = k_a B + j_b, g = j_c, t) at ((q C) + (g ^ (q mod SW))) TC + t.  Step 1 writes and step 2 reads / writes
// 64 consecutive words per wavefront (q constant per wavefront: the XOR permutes inside the run); step 3 reads a fixed g of 32 / TC
// consecutive q per lane group, which the XOR spreads over all 64 banks.
//
// Memory: a non-last pass touches TC consecutive elements (32 bytes for TC = 4) of R rows; the workgroups that share a 128-byte
// line are placed on the same XCD (one L2) next to each other in dispatch order, see tile_of_block.
#pragma once
#include "dft_regs.cuh"
#include "l24.cuh"
#include "ntt_params.cuh"

#ifndef NB_HD
#define NB_HD __device__ __forceinline__
#endif

namespace nttbig {

struct alignas(16) Row4 {
    uint64_t w[4];
};

template <int LOG_B_, int LOG_C_, int LOG_TC_>
struct Geo {
    static constexpr int LOG_A = 4, LOG_B = LOG_B_, LOG_C = LOG_C_, LOG_TC = LOG_TC_;
    static constexpr int A = 16, B = 1 << LOG_B, C = 1 << LOG_C, TC = 1 << LOG_TC;
    static constexpr int LOG_BC = LOG_B + LOG_C, BC = 1 << LOG_BC;
    static constexpr int LOG_R = LOG_A + LOG_BC, R = 1 << LOG_R;
    static constexpr int NT = (R * TC) / 16;          // lanes per workgroup: sixteen elements each in every step
    static constexpr int NB = 16 / B, NC = 16 / C;    // B-point / C-point DFTs per lane in steps 2 / 3
    static constexpr int SW = (32 / TC) < C ? (32 / TC) : C;
    static_assert(LOG_B >= 3 && LOG_B <= 4 && LOG_C >= 3 && LOG_C <= 4 && LOG_BC <= 8, "steps of 8 or 16 points");
    static_assert(TC <= 16 && NT <= 1024, "tile shape");
    static constexpr int LDS_WORDS = R * TC;
    static NB_HD int elem(int q, int g, int t) { return (((q << LOG_C) + (g ^ (q & (SW - 1)))) << LOG_TC) + t; }
};

// wave-uniform geometry of a pass
template <class G, bool LAST>
struct Pass {
    uint32_t L, log_ncols, log_s, log_mult, rm_groups;
    uint64_t n, ncols, total_cols;
    bool RM;
    NB_HD explicit Pass(const PassParams<uint64_t> &p) {
        L = p.log_n;
        n = 1ull << L;
        ncols = n >> G::LOG_R;
        log_ncols = L - G::LOG_R;
        RM = LAST && p.rowmajor;
        rm_groups = RM ? (p.rm_base_cols + (1u << p.rm_log_i) - 1) >> p.rm_log_i : 0;
        total_cols = RM ? ((uint64_t)rm_groups << (p.rm_log_b + log_ncols + p.rm_log_i)) : ncols * (uint64_t)p.nvec;
        uint32_t ls = L;
        for (uint32_t q = 0; q <= p.pass; q++) ls -= p.log_r[q];
        log_mult = L - ls - G::LOG_R;          // n / n_p = R_1 .. R_{p-1}
        log_s = LAST ? 0 : ls;                 // the last pass runs along the contiguous axis
    }
    // (vector v, column c) of the joint index cc; row-major mode: [column group][coset u][column c][column-in-group] (last fastest),
    // returns whether the lane carries a real column (lanes past base_cols in the last group only write padding zeros)
    NB_HD bool decompose(const PassParams<uint64_t> &p, uint64_t cc, uint64_t &v, uint64_t &c, uint32_t &bc, uint32_t &u) const {
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
    }
    // element offset of the pass digit 0 of column c
    NB_HD uint64_t base_of(const PassParams<uint64_t> &p, uint64_t c) const {
        if (!LAST) {
            const uint64_t rem = c & ((1ull << log_s) - 1);
            return ((c >> log_s) << (log_s + G::LOG_R)) + rem;
        }
        uint64_t base = 0, cr = c;
        uint32_t ls = L;
        for (uint32_t q = 0; q + 1 < p.npass; q++) {
            ls -= p.log_r[q];
            base += (cr & ((1ull << p.log_r[q]) - 1)) << ls;
            cr >>= p.log_r[q];
        }
        return base;
    }
};

// ---- step 1: loads, coset pre-scale, 16-point DFT over the top digit, omega_R^(k_a j_bc), LDS --------------------------------
template <class G, bool LAST>
NB_HD void step1_load(const PassParams<uint64_t> &p, uint64_t tile, int tid, uint64_t (&x)[16]) {
    typedef uint64_t T;
    const Pass<G, LAST> g(p);
    const int t1 = tid & (G::TC - 1), jbc = tid >> G::LOG_TC;
    const uint64_t cc = tile * G::TC + t1;
    uint64_t v = 0, c = 0;
    uint32_t bc, u;
    const bool active = cc < g.total_cols && g.decompose(p, cc, v, c, bc, u);
    if (!active) {
#pragma unroll
        for (int a = 0; a < 16; a++) x[a] = 0;
        return;
    }
    const uint64_t base = g.base_of(p, c);
    uint32_t vs, vq, vr;
    divmod_uniform((uint32_t)v, p.src_div, vs, vr);
    divmod_uniform(vs, p.src_inner, vq, vr);
    const T *src = p.src + (uint64_t)vq * p.src_vec_stride + (uint64_t)vr * p.src_inner_stride;
    // the lane's sixteen inputs j = j_a BC + j_bc sit at a wave-uniform stride: one running pointer
    const T *ptr = src + (base + ((uint64_t)jbc << g.log_s)) * p.src_es;
    const uint64_t istep = ((uint64_t)G::BC << g.log_s) * p.src_es;
#pragma unroll
    for (int a = 0; a < 16; a++) {
        x[a] = *ptr;
        ptr += istep;
    }
    if (p.pre_lo != nullptr && p.pass == 0) {
        // coset pre-scale base^j (evaluate_poly_with_offset): the lane's inputs form a geometric progression
        uint32_t uq, cu;
        divmod_uniform((uint32_t)v, p.pre_mod, uq, cu);
        const T *plo = p.pre_lo + cu * p.pre_lo_stride, *phi = p.pre_hi + cu * p.pre_hi_stride;
        T cur = series_at32<F64>(plo, phi, p.pre_log_lo, (uint32_t)(base + ((uint64_t)jbc << g.log_s)));
        const T stp = series_at32<F64>(plo, phi, p.pre_log_lo, (uint32_t)G::BC << g.log_s);
#pragma unroll
        for (int a = 0; a < 16; a++) {
            x[a] = gl::mul(x[a], cur);
            if (a + 1 < 16) cur = gl::mul(cur, stp);
        }
    }
}

template <class G, bool HALF>
NB_HD void step1_compute(int tid, const uint64_t (&x)[16], uint64_t *lds, const uint64_t *bigtab) {
    typedef l24::Dft<4> DA;
    const int t1 = tid & (G::TC - 1), jbc = tid >> G::LOG_TC;
    const int be = jbc >> G::LOG_C, ga = jbc & (G::C - 1);
    int32_t v[DA::NV];
#pragma unroll
    for (int a = 0; a < 16; a++) DA::load(v, a, x[a]);
    DA::run(v);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int ka = brev(i, 4);
        uint32_t y[4];
#pragma unroll
        for (int q = 0; q < 4; q++) y[q] = DA::limb(v, i, q);
        uint64_t val = l24::fold_lazy(l24::mul4_one(y));
        if (ka != 0) {
            // omega_R^(k_a j_bc): e < R, no wrap.  HALF: the table holds e < R / 2, omega^(e + R/2) = -omega^e (the twiddle is never 0)
            const uint32_t e = (uint32_t)ka * (uint32_t)jbc;
            uint64_t w;
            if constexpr (HALF) {
                w = bigtab[e & (G::R / 2 - 1)];
                if (ka > 1 && (e >> (G::LOG_R - 1))) w = gl::P - w;      // k_a = 1: e = j_bc < R / 16
            } else {
                w = bigtab[e];
            }
            val = gl::mul(val, w);       // the Montgomery reduction accepts any 64-bit word times a canonical one
        }
        lds[G::elem(ka * G::B + be, ga, t1)] = val;
    }
}

// ---- step 2: B-point DFTs over the middle digit, omega_BC^(k_b j_c) on the way out of the limb form, in place -------------------
template <class G, bool LAST>
NB_HD void step2(const PassParams<uint64_t> &p, int tid, uint64_t *lds, const Row4 *w4) {
    typedef l24::Dft<G::LOG_B> DB;
#pragma unroll
    for (int h = 0; h < G::NB; h++) {
        const int un = tid + G::NT * h;
        const int t = un & (G::TC - 1), ga = (un >> G::LOG_TC) & (G::C - 1), al = un >> (G::LOG_TC + G::LOG_C);
        uint64_t y[G::B];
#pragma unroll
        for (int bb = 0; bb < G::B; bb++) y[bb] = lds[G::elem(al * G::B + bb, ga, t)];
        int32_t v[DB::NV];
#pragma unroll
        for (int bb = 0; bb < G::B; bb++) DB::load(v, bb, y[bb]);
        DB::run(v);
#pragma unroll
        for (int i = 0; i < G::B; i++) {
            const int kb = brev(i, G::LOG_B);
            uint32_t yl[4];
#pragma unroll
            for (int q = 0; q < 4; q++) yl[q] = DB::limb(v, i, q);
            uint64_t val;
            if (kb != 0 || (LAST && p.scale_in_w256)) {
                const Row4 r = w4[kb * ga];            // e = k_b j_c < BC
                val = l24::fold_lazy(l24::mul4(yl, r.w[0], r.w[1], r.w[2], r.w[3]));
            } else {
                val = l24::fold_lazy(l24::mul4_one(yl));
            }
            lds[G::elem(al * G::B + kb, ga, t)] = val;
        }
    }
}

// ---- step 3: C-point DFTs over the low digit, inter-pass twiddles / output scaling, stores -------------------------------------
template <class G, bool LAST>
NB_HD void step3(const PassParams<uint64_t> &p, uint64_t tile, int tid, const uint64_t *lds) {
    typedef uint64_t T;
    typedef l24::Dft<G::LOG_C> DC;
    const Pass<G, LAST> g(p);
    const uint32_t log_s = g.log_s;
    const uint64_t n = g.n, ncols = g.ncols;
#pragma unroll
    for (int h = 0; h < G::NC; h++) {
        const int un = tid + G::NT * h;
        const int t = un & (G::TC - 1), q = un >> G::LOG_TC;
        const uint64_t cc = tile * G::TC + t;
        if (cc >= g.total_cols) continue;
        uint64_t y[G::C];
#pragma unroll
        for (int gg = 0; gg < G::C; gg++) y[gg] = lds[G::elem(q, gg, t)];
        uint64_t v, c;
        uint32_t bc2, u2;
        const bool real_col = g.decompose(p, cc, v, c, bc2, u2);
        uint32_t dq, dr;
        divmod_uniform((uint32_t)v, p.dst_inner, dq, dr);
        T *dst = p.dst + (uint64_t)dq * p.dst_vec_stride + (uint64_t)dr * p.dst_inner_stride;
        const uint64_t rem = c & ((1ull << log_s) - 1);
        const uint64_t base_nl = ((c >> log_s) << (log_s + G::LOG_R)) + rem;

        int32_t lv[DC::NV];
#pragma unroll
        for (int gg = 0; gg < G::C; gg++) DC::load(lv, gg, y[gg]);
        DC::run(lv);
        // output digit k = kbase + AB k_c, k_c = brev(register index)
        const uint32_t kbase = (uint32_t)(q >> G::LOG_B) + 16u * (uint32_t)(q & (G::B - 1));
        constexpr uint32_t KSTEP = 16u * G::B;
        auto out = [&](int i, bool lazy) -> T {
            uint32_t yl[4];
#pragma unroll
            for (int k = 0; k < 4; k++) yl[k] = DC::limb(lv, i, k);
            return lazy ? l24::fold_lazy(l24::mul4_one(yl)) : l24::fold(l24::mul4_one(yl));
        };
        if constexpr (!LAST) {
            // inter-pass twiddles omega_n^(k rem mult): a geometric progression along k_c (two look-ups, a multiplication chain)
            T *o_ptr = dst + (base_nl + ((uint64_t)kbase << log_s)) * p.dst_es;
            const uint64_t o_step = ((uint64_t)KSTEP * p.dst_es) << log_s;
            const uint32_t r32 = (uint32_t)rem;
            T cur = series_at32<F64>(p.w_lo, p.w_hi, p.w_log_lo, (kbase * r32) << g.log_mult);
            const T stp = series_at32<F64>(p.w_lo, p.w_hi, p.w_log_lo, (KSTEP * r32) << g.log_mult);
#pragma unroll
            for (int ip = 0; ip < G::C; ip++) {
                const int i = brev(ip, G::LOG_C);
                *o_ptr = gl::mul(out(i, true), cur);
                o_ptr += o_step;
                if (ip + 1 < G::C) cur = gl::mul(cur, stp);
            }
        } else if (g.RM) {
            // LDE row u + b m (m = c + ncols k), column bc; the last group's lanes also zero the padding columns of the row
            T *cell = p.dst + (u2 + ((c + ncols * (uint64_t)kbase) << p.rm_log_b)) * p.rm_row_width + bc2;
            const uint64_t o_step = ((ncols * KSTEP) << p.rm_log_b) * p.rm_row_width;
            const bool pads = (bc2 >> p.rm_log_i) + 1 == g.rm_groups;
#pragma unroll
            for (int ip = 0; ip < G::C; ip++) {
                const T val = out(brev(ip, G::LOG_C), false);
                if (real_col) *cell = val;
                if (pads) {
                    T *row = cell - bc2;
                    for (uint64_t pc = p.rm_base_cols + (bc2 & ((1u << p.rm_log_i) - 1)); pc < p.rm_row_width; pc += 1u << p.rm_log_i) row[pc] = 0;
                }
                cell += o_step;
            }
        } else {
            // natural order: index k0 + krel ncols (inverse transform: negated, k = 0 stays)
            const uint64_t k0 = c + ncols * (uint64_t)kbase;
            T *o_ptr, *o_ptr0;
            int64_t o_step;
            uint32_t o_k32, o_kstep32;
            if (p.inverse) {
                o_ptr = dst + (n - k0) * p.dst_es;
                o_ptr0 = dst + ((n - k0) & (n - 1)) * p.dst_es;
                o_step = -(int64_t)(ncols * KSTEP * (uint64_t)p.dst_es);
                o_k32 = (uint32_t)(n - k0);
                o_kstep32 = 0u - (uint32_t)(ncols * KSTEP);
            } else {
                o_ptr = o_ptr0 = dst + k0 * p.dst_es;
                o_step = (int64_t)(ncols * KSTEP * (uint64_t)p.dst_es);
                o_k32 = (uint32_t)k0;
                o_kstep32 = (uint32_t)(ncols * KSTEP);
            }
#pragma unroll
            for (int ip = 0; ip < G::C; ip++) {
                T val = out(brev(ip, G::LOG_C), false);
                if (p.post_lo != nullptr) {
                    const uint32_t k = (o_k32 + (uint32_t)ip * o_kstep32) & (uint32_t)(n - 1);
                    val = gl::mul(val, series_at32<F64>(p.post_lo, p.post_hi, p.post_log_lo, k));
                } else if (p.has_post_const && !p.scale_in_w256) {
                    val = gl::mul(val, p.post_const);
                }
                if (ip == 0) {
                    *o_ptr0 = val;
                } else {
                    o_ptr += o_step;
                    *o_ptr = val;
                }
            }
        }
    }
}

// Workgroups that share 128-byte lines (16 / TC consecutive tiles) go to the same XCD — blockIdx mod 8 — as neighbours in dispatch
// order, so the partial-line reads and writes of a non-last pass meet in one L2.
template <class G>
NB_HD uint64_t tile_of_block(uint32_t block, uint32_t nblocks) {
    constexpr uint32_t LPT = 16 / G::TC > 0 ? 16 / G::TC : 1;
    constexpr uint32_t GRP = 8 * LPT;
    const uint32_t full = nblocks / GRP * GRP;
    if (LPT == 1 || block >= full) return block;
    const uint32_t grp = block / GRP, x = block & 7u, i = (block >> 3) % LPT;
    return ((uint64_t)grp * 8 + x) * LPT + i;
}

#if defined(__HIPCC__)
template <int LOG_B, int LOG_C, int LOG_TC, bool LAST, bool HALF>
__global__ __launch_bounds__((Geo<LOG_B, LOG_C, LOG_TC>::NT)) void ntt_pass3(PassParams<uint64_t> p) {
    typedef Geo<LOG_B, LOG_C, LOG_TC> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char nttbig_smem[];
    uint64_t *lds = reinterpret_cast<uint64_t *>(nttbig_smem);
    Row4 *w4 = reinterpret_cast<Row4 *>(lds + G::LDS_WORDS);
    uint64_t *bt = reinterpret_cast<uint64_t *>(w4 + G::BC);
    const int tid = threadIdx.x;
    const uint64_t tile = tile_of_block<G>(blockIdx.x, gridDim.x);
    uint64_t x[16];
    step1_load<G, LAST>(p, tile, tid, x);
    {
        // tables -> LDS (once per workgroup): omega_R^e words, and the four-word rows c omega_BC^e T^k (rows of the 256-row table)
        constexpr int NBT = HALF ? G::R / 2 : G::R;
        for (int i = tid; i < NBT; i += G::NT) bt[i] = p.big_tab[i];
        const uint4 *rows = reinterpret_cast<const uint4 *>(p.w256);
        uint4 *wl = reinterpret_cast<uint4 *>(w4);
        for (int i = tid; i < 2 * G::BC; i += G::NT) wl[i] = rows[2 * ((i >> 1) << (8 - G::LOG_BC)) + (i & 1)];
    }
    __syncthreads();
    step1_compute<G, HALF>(tid, x, lds, bt);
    __syncthreads();
    step2<G, LAST>(p, tid, lds, w4);
    __syncthreads();
    step3<G, LAST>(p, tile, tid, lds);
}
#endif

template <int LOG_B, int LOG_C, int LOG_TC, bool HALF>
constexpr size_t smem_bytes() {
    typedef Geo<LOG_B, LOG_C, LOG_TC> G;
    return (size_t)G::LDS_WORDS * 8 + (size_t)G::BC * 32 + (size_t)(HALF ? G::R / 2 : G::R) * 8;
}

}  // namespace nttbig
