// Constraint evaluation over the constraint-evaluation domain for the reference's example AIRs.
//
// Reference behaviour reproduced (single-value assertions; one optional auxiliary segment — evaluate_fragment_full,
// evaluate_aux_transition, BoundaryConstraints::evaluate_all: default.rs:214-271,301-332, boundary.rs:103-115):
//   DefaultConstraintEvaluator::evaluate / evaluate_fragment_main / evaluate_main_transition
//                                         prover/src/constraints/evaluator/default.rs:52-106,165-210,277-299
//   BoundaryConstraints::evaluate_main    prover/src/constraints/evaluator/boundary.rs:86-97,213-232,318-327
//   PeriodicValueTable                    prover/src/constraints/evaluator/periodic_table.rs:24-88
//   ConstraintEvaluationTable::combine    prover/src/constraints/evaluation_table.rs:163-176,317-407
//   ConstraintDivisor                     air/src/air/divisor.rs:43-71,120-123
//   read_main_trace_frame_into            prover/src/trace/trace_lde/default/mod.rs:171-185
//   AIRs: FibSmall (examples/src/fibonacci/fib_small/air.rs:41-59), RescueAir (examples/src/rescue/air.rs:88-138,
//         examples/src/rescue/rescue.rs:60-123; f128 only, constants generated into rescue_f128_constants.h)
//
//         Fib8 / MulFib2 / MulFib8 (examples/src/fibonacci/{fib8,mulfib2,mulfib8}/air.rs), Vdf regular and exempt
//         (examples/src/vdf/{regular,exempt}/air.rs; the exempt variant has two transition exemptions)
//
// AIR transition functions are user Rust closures in the reference, so they cannot cross a C ABI generically: seven of
// the reference's example AIRs are hand-written device functions selected by `air`.  Everything around them is generic: one lane per
// constraint-evaluation step reads its frame straight from the device-resident row-major LDE (no D2H copy of the
// trace), folds the transition evaluations with the composition coefficients, multiplies by the inverse divisor
// 1 / (x^n - 1) * (x - g^(n-1)), adds the boundary groups divided by (x - g^step), and writes the combined value.
// The per-step inverses 1 / (x_i - g^step) come from a batch-inversion kernel (16 steps per lane, one field inversion).
#include <string.h>

#include <type_traits>
#include <vector>

#include "batch_inv.cuh"
#include "dft_regs.cuh"
#include "tables.cuh"
#include "wf_internal.h"

#define RESCUE_CONST static const
#include "rescue_f128_constants.h"

namespace {

constexpr int MAX_GROUPS = 8;
constexpr int MAX_ASSERT = 64;
constexpr int INV_CHUNK = 16;

// ---- AIRs -------------------------------------------------------------------------------------------------------
struct AirFibSmall {
    static constexpr int WIDTH = 2, NT = 2, NP = 0, CYCLE = 0, LOG_CE = 1, EXEMPT = 1, AUX_WIDTH = 0, NTA = 0, NR = 0;
    struct Consts {};
    template <class F>
    static __device__ __forceinline__ void transition(const typename F::T *cur, const typename F::T *next, const typename F::T *,
                                                      const Consts &, typename F::T *res) {
        res[0] = F::sub(next[0], F::add(cur[0], cur[1]));     // s_{0,i+1} = s_{0,i} + s_{1,i}
        res[1] = F::sub(next[1], F::add(cur[1], next[0]));    // s_{1,i+1} = s_{1,i} + s_{0,i+1}
    }
};

struct AirFib8 {      // examples/src/fibonacci/fib8/air.rs:40-65
    static constexpr int WIDTH = 2, NT = 2, NP = 0, CYCLE = 0, LOG_CE = 1, EXEMPT = 1, AUX_WIDTH = 0, NTA = 0, NR = 0;
    struct Consts {};
    template <class F>
    static __device__ __forceinline__ void transition(const typename F::T *cur, const typename F::T *next, const typename F::T *,
                                                      const Consts &, typename F::T *res) {
        typename F::T a = F::add(cur[0], cur[1]), b = F::add(cur[1], a);      // n0, n1
#pragma unroll
        for (int k = 0; k < 3; k++) {                                          // (n2, n3), (n4, n5), (n6, n7)
            a = F::add(a, b);
            b = F::add(b, a);
        }
        res[0] = F::sub(next[0], a);
        res[1] = F::sub(next[1], b);
    }
};

struct AirMulFib2 {   // examples/src/fibonacci/mulfib2/air.rs:41-60
    static constexpr int WIDTH = 2, NT = 2, NP = 0, CYCLE = 0, LOG_CE = 1, EXEMPT = 1, AUX_WIDTH = 0, NTA = 0, NR = 0;
    struct Consts {};
    template <class F>
    static __device__ __forceinline__ void transition(const typename F::T *cur, const typename F::T *next, const typename F::T *,
                                                      const Consts &, typename F::T *res) {
        res[0] = F::sub(next[0], F::mul(cur[0], cur[1]));
        res[1] = F::sub(next[1], F::mul(cur[1], next[0]));
    }
};

struct AirMulFib8 {   // examples/src/fibonacci/mulfib8/air.rs:52-82
    static constexpr int WIDTH = 8, NT = 8, NP = 0, CYCLE = 0, LOG_CE = 1, EXEMPT = 1, AUX_WIDTH = 0, NTA = 0, NR = 0;
    struct Consts {};
    template <class F>
    static __device__ __forceinline__ void transition(const typename F::T *cur, const typename F::T *next, const typename F::T *,
                                                      const Consts &, typename F::T *res) {
        res[0] = F::sub(next[0], F::mul(cur[6], cur[7]));
        res[1] = F::sub(next[1], F::mul(cur[7], next[0]));
#pragma unroll
        for (int k = 2; k < 8; k++) res[k] = F::sub(next[k], F::mul(next[k - 2], next[k - 1]));
    }
};

// examples/src/vdf/regular/air.rs:49-61 (EX = 1) and vdf/exempt/air.rs (EX = 2: the last TWO steps are exempt)
template <int EX>
struct AirVdf {
    static constexpr int WIDTH = 1, NT = 1, NP = 0, CYCLE = 0, LOG_CE = 1, EXEMPT = EX, AUX_WIDTH = 0, NTA = 0, NR = 0;   // degree 3: ce_blowup = npo2(3 - 1) = 2 (degree.rs:96-99)
    struct Consts {
        f128::u128 forty_two;    // BaseElement::new(42) in the field's internal form (low word for the 64-bit fields)
    };
    template <class F>
    static __device__ __forceinline__ void transition(const typename F::T *cur, const typename F::T *next, const typename F::T *,
                                                      const Consts &c, typename F::T *res) {
        typedef typename F::T T;
        const T cube = F::mul(F::mul(next[0], next[0]), next[0]);
        res[0] = F::sub(cur[0], F::add(cube, (T)c.forty_two));
    }
};

struct AirRescue {   // F128 only
    static constexpr int WIDTH = 4, NT = 4, NP = 9, CYCLE = 16, LOG_CE = 2, EXEMPT = 1, AUX_WIDTH = 0, NTA = 0, NR = 0;
    struct Consts {
        f128::u128 mds[16], inv_mds[16];
    };
    template <class F>
    static __device__ __forceinline__ void mds(typename F::T (&st)[4], const f128::u128 *m) {
        typename F::T r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            r[i] = F::mul(m[i * 4], st[0]);
#pragma unroll
            for (int j = 1; j < 4; j++) r[i] = F::add(r[i], F::mul(m[i * 4 + j], st[j]));
        }
#pragma unroll
        for (int i = 0; i < 4; i++) st[i] = r[i];
    }
    // rescue::enforce_round (examples/src/rescue/rescue.rs:60-89) without the flag: d[i] = step2[i] - step1[i]
    template <class F>
    static __device__ __forceinline__ void round_diff(const typename F::T *cur, const typename F::T *next, const typename F::T *ark,
                                                      const Consts &c, typename F::T (&d)[4]) {
        typedef typename F::T T;
        T s1[4], s2[4];
#pragma unroll
        for (int i = 0; i < 4; i++) s1[i] = F::mul(F::mul(cur[i], cur[i]), cur[i]);           // apply_sbox
        mds<F>(s1, c.mds);
#pragma unroll
        for (int i = 0; i < 4; i++) s1[i] = F::add(s1[i], ark[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) s2[i] = F::sub(next[i], ark[4 + i]);
        mds<F>(s2, c.inv_mds);
#pragma unroll
        for (int i = 0; i < 4; i++) s2[i] = F::mul(F::mul(s2[i], s2[i]), s2[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) d[i] = F::sub(s2[i], s1[i]);
    }
    template <class F>
    static __device__ __forceinline__ void transition(const typename F::T *cur, const typename F::T *next, const typename F::T *per,
                                                      const Consts &c, typename F::T *res) {
        typedef typename F::T T;
        const T flag = per[0];
        T d[4];
        round_diff<F>(cur, next, per + 1, c, d);
        const T copy_flag = F::sub((T)1, flag);                                               // not(hash_flag); f128: ONE = 1
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const T round = F::mul(flag, d[i]);
            const T copy = F::mul(copy_flag, i < 2 ? F::sub(cur[i], next[i]) : next[i]);
            res[i] = F::add(round, copy);
        }
    }
};

// RescueRapsAir (examples/src/rescue_raps/air.rs:60-253), F128 only: two Rescue chains side by side and an auxiliary segment of
// three columns over E that carries the permutation argument between the values the chains absorb.
struct AirRescueRaps {
    static constexpr int WIDTH = 8, NT = 8, NP = 10, CYCLE = 16, LOG_CE = 2, EXEMPT = 1, AUX_WIDTH = 3, NTA = 3, NR = 3;
    typedef AirRescue::Consts Consts;
    template <class F>
    static __device__ __forceinline__ void transition(const typename F::T *cur, const typename F::T *next, const typename F::T *per,
                                                      const Consts &c, typename F::T *res) {   // air.rs:91-153
        typedef typename F::T T;
        const T hash_flag = per[0], absorption_flag = per[1];
        const T copy_flag = F::sub((T)1, F::add(hash_flag, absorption_flag));
#pragma unroll
        for (int h = 0; h < 2; h++) {
            T d[4];
            AirRescue::round_diff<F>(cur + 4 * h, next + 4 * h, per + 2, c, d);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const T same = F::sub(cur[4 * h + i], next[4 * h + i]);
                T r = F::add(F::mul(hash_flag, d[i]), F::mul(copy_flag, same));               // enforce_round + enforce_hash_copy
                if (i >= 2) r = F::add(r, F::mul(absorption_flag, same));                     // capacity registers stay put while absorbing
                res[4 * h + i] = r;
            }
        }
    }
    // evaluate_aux_transition, air.rs:155-214: main frame over the base field, aux frame / rand / result over E (D words each)
    template <class F, int D>
    static __device__ __forceinline__ void aux_transition(const typename F::T *cur, const typename F::T *next, const typename F::T (*acur)[D],
                                                          const typename F::T (*anext)[D], const typename F::T *per,
                                                          const typename F::T (*rand)[D], typename F::T (*res)[D]) {
        typedef typename F::T T;
        const T absorption_flag = per[1];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const T d0 = F::sub(next[4 * h], cur[4 * h]), d1 = F::sub(next[4 * h + 1], cur[4 * h + 1]);
#pragma unroll
            for (int d = 0; d < D; d++) {                                                     // flag * (aux[h] - (r0 * d0 + r1 * d1)), base * E
                const T copied = F::add(F::mul(rand[0][d], d0), F::mul(rand[1][d], d1));
                res[h][d] = F::mul(absorption_flag, F::sub(acur[h][d], copied));
            }
        }
        T a[D], b[D], u[D], v[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            a[d] = F::add(acur[1][d], rand[2][d]);
            b[d] = F::add(acur[0][d], rand[2][d]);
        }
        F::template ext_mul<D>(anext[2], a, u);
        F::template ext_mul<D>(acur[2], b, v);
#pragma unroll
        for (int d = 0; d < D; d++) res[2][d] = F::sub(u[d], v[d]);
    }
};

// zb[q][i] = 1 / (x_i - b_q),  x_i = offset * g_ce^i  (series table),  INV_CHUNK consecutive i per lane, one field inversion per
// workgroup (batch_inv.cuh)
// multi-value assertions (Assertion::periodic / ::sequence): group q's divisor is x^k - g^(first_step k) with k = 2^log_k[q] asserted steps
// (air/src/air/divisor.rs:64-97); log_k = 0 is the single-value case x - g^step
struct GroupLogK {
    uint32_t v[8];
};
static_assert(sizeof(GroupLogK) == 32, "one entry per boundary group");

template <class F>
__global__ __launch_bounds__(256) void divisor_inv_kernel(const typename F::T *x_lo, const typename F::T *x_hi, uint32_t x_log_lo,
                                                          uint64_t ce, const typename F::T *b, typename F::T one, uint64_t e_lo,
                                                          uint64_t e_hi, GroupLogK log_k, typename F::T *zb) {
    typedef typename F::T T;
    __shared__ T sA[256], sB[256];
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * INV_CHUNK;
    const uint32_t q = blockIdx.y;
    const T bq = b[q];
    const uint32_t cnt = i0 >= ce ? 0u : (ce - i0 < INV_CHUNK ? (uint32_t)(ce - i0) : (uint32_t)INV_CHUNK);
    T v[INV_CHUNK], pre[INV_CHUNK];
    T acc = one;
#pragma unroll
    for (int k = 0; k < INV_CHUNK; k++) {
        if ((uint32_t)k < cnt) {
            T xv = series_at<F>(x_lo, x_hi, x_log_lo, i0 + k);
            for (uint32_t sq = 0; sq < log_k.v[q]; sq++) xv = F::mul(xv, xv);        // x^k, k a power of two
            v[k] = F::sub(xv, bq);
            pre[k] = acc;
            acc = F::mul(acc, v[k]);
        }
    }
    acc = block_inverse_of_products<F>(acc, one, e_lo, e_hi, sA, sB);
#pragma unroll
    for (int k = INV_CHUNK - 1; k >= 0; k--) {
        if ((uint32_t)k < cnt) {
            zb[(uint64_t)q * ce + i0 + k] = F::mul(acc, pre[k]);
            acc = F::mul(acc, v[k]);
        }
    }
}

template <class T, int D>
struct EvalParams {
    const T *lde;
    uint64_t row_width;
    uint32_t log_n, log_lde_blowup, log_ce_blowup;
    const T *x_lo, *x_hi;         // x_i = offset * g_ce^i
    uint32_t x_log_lo;
    const T *ptab;                // periodic table [plen][NP]
    const T *zt;                  // [ce_blowup] inverse transition-divisor numerators
    T exempt, exempt2;            // g^(n-1), g^(n-2): the steps the transition divisor exempts (divisor.rs:43-51)
    const T *zb;                  // [ngroups][ce]
    const T *cc_t;                // [NT][D]
    uint32_t num_assert, ngroups;
    const uint32_t *a_col, *a_group;
    const T *a_val;               // [num_assert]
    const uint32_t *a_seq;        // [num_assert]: index of the assertion's value vector in `seq`, or ~0u for a constant value (a_val)
    const T *seq;                 // [sequences][ce]: b(x_i g^(-first_step)) of the sequence assertions over the ce domain
    const T *cc_b;                // [num_assert][D]
    // the auxiliary segment (AIR::AUX_WIDTH > 0): rows of AUX_WIDTH elements of E
    const T *aux_lde;
    uint64_t aux_row_width;       // in base elements
    const T *rand;                // [NR][D]
    uint32_t num_aux_assert;
    const uint32_t *x_col, *x_group;
    const T *x_val;               // [num_aux_assert][D]
    const T *cc_x;                // [num_aux_assert][D]
    T *out;                       // [ce][D]
};

template <class F, class AIR, int D>
__global__ __launch_bounds__(256) void constraints_kernel(EvalParams<typename F::T, D> p, typename AIR::Consts consts) {
    typedef typename F::T T;
    const uint64_t ce = 1ull << (p.log_n + p.log_ce_blowup);
    const uint64_t step = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (step >= ce) return;
    const uint64_t lde_rows = 1ull << (p.log_n + p.log_lde_blowup);
    const uint64_t lde_step = step << (p.log_lde_blowup - p.log_ce_blowup);
    const T *rc = p.lde + lde_step * p.row_width;
    const T *rn = p.lde + ((lde_step + (1ull << p.log_lde_blowup)) & (lde_rows - 1)) * p.row_width;
    T cur[AIR::WIDTH], next[AIR::WIDTH], per[AIR::NP > 0 ? AIR::NP : 1], tev[AIR::NT];
#pragma unroll
    for (int k = 0; k < AIR::WIDTH; k++) {
        cur[k] = F::load_norm(rc[k]);
        next[k] = F::load_norm(rn[k]);
    }
    if (AIR::NP > 0) {
        const uint32_t plen = (uint32_t)AIR::CYCLE << p.log_ce_blowup;
        const T *row = p.ptab + (size_t)(step & (plen - 1)) * AIR::NP;
#pragma unroll
        for (int k = 0; k < AIR::NP; k++) per[k] = row[k];
    }
    AIR::template transition<F>(cur, next, per, consts, tev);
    T acc[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        acc[d] = F::mul(p.cc_t[d], tev[0]);
#pragma unroll
        for (int k = 1; k < AIR::NT; k++) acc[d] = F::add(acc[d], F::mul(p.cc_t[k * D + d], tev[k]));
    }
    // the auxiliary segment's transition constraints join the same sum, coefficient times evaluation over E
    // (evaluate_fragment_full / evaluate_aux_transition, default.rs:214-271,301-332)
    constexpr int AW = AIR::AUX_WIDTH > 0 ? AIR::AUX_WIDTH : 1;
    T acur[AW][D];
    if constexpr (AIR::AUX_WIDTH > 0) {
        const T *ac = p.aux_lde + lde_step * p.aux_row_width;
        const T *an = p.aux_lde + ((lde_step + (1ull << p.log_lde_blowup)) & (lde_rows - 1)) * p.aux_row_width;
        T anext[AW][D], rnd[AIR::NR][D], aev[AIR::NTA][D];
#pragma unroll
        for (int k = 0; k < AIR::AUX_WIDTH; k++)
#pragma unroll
            for (int d = 0; d < D; d++) {
                acur[k][d] = F::load_norm(ac[k * D + d]);
                anext[k][d] = F::load_norm(an[k * D + d]);
            }
#pragma unroll
        for (int k = 0; k < AIR::NR; k++)
#pragma unroll
            for (int d = 0; d < D; d++) rnd[k][d] = p.rand[k * D + d];
        AIR::template aux_transition<F, D>(cur, next, acur, anext, per, rnd, aev);
#pragma unroll
        for (int k = 0; k < AIR::NTA; k++) {
            T cc[D], t[D];
#pragma unroll
            for (int d = 0; d < D; d++) cc[d] = p.cc_t[(AIR::NT + k) * D + d];
            F::template ext_mul<D>(cc, aev[k], t);
#pragma unroll
            for (int d = 0; d < D; d++) acc[d] = F::add(acc[d], t[d]);
        }
    }
    const T x = series_at<F>(p.x_lo, p.x_hi, p.x_log_lo, step);
    T ze = F::mul(p.zt[step & ((1u << p.log_ce_blowup) - 1)], F::sub(x, p.exempt));
    if (AIR::EXEMPT == 2) ze = F::mul(ze, F::sub(x, p.exempt2));
#pragma unroll
    for (int d = 0; d < D; d++) acc[d] = F::mul(acc[d], ze);
    for (uint32_t q = 0; q < p.ngroups; q++) {
        T grp[D];
#pragma unroll
        for (int d = 0; d < D; d++) grp[d] = F::zero();
        for (uint32_t k = 0; k < p.num_assert; k++) {
            if (p.a_group[k] != q) continue;
            // state[column] with a uniform column index: select without dynamic register indexing
            T sv = cur[0];
            const uint32_t col = p.a_col[k];
#pragma unroll
            for (int c = 1; c < AIR::WIDTH; c++) sv = col == (uint32_t)c ? cur[c] : sv;
            const uint32_t sq = p.a_seq[k];
            const T ev = F::sub(sv, sq == ~0u ? p.a_val[k] : p.seq[(uint64_t)sq * ce + step]);
#pragma unroll
            for (int d = 0; d < D; d++) grp[d] = F::add(grp[d], F::mul(p.cc_b[k * D + d], ev));
        }
        if constexpr (AIR::AUX_WIDTH > 0) {                       // aux single-value assertions of this divisor (boundary.rs:241-266)
            for (uint32_t k = 0; k < p.num_aux_assert; k++) {
                if (p.x_group[k] != q) continue;
                const uint32_t col = p.x_col[k];
                T ev[D], cc[D], t[D];
#pragma unroll
                for (int d = 0; d < D; d++) {
                    T sv = acur[0][d];
#pragma unroll
                    for (int c = 1; c < AIR::AUX_WIDTH; c++) sv = col == (uint32_t)c ? acur[c][d] : sv;
                    ev[d] = F::sub(sv, p.x_val[k * D + d]);
                    cc[d] = p.cc_x[k * D + d];
                }
                F::template ext_mul<D>(cc, ev, t);
#pragma unroll
                for (int d = 0; d < D; d++) grp[d] = F::add(grp[d], t[d]);
            }
        }
        const T z = p.zb[(uint64_t)q * ce + step];
#pragma unroll
        for (int d = 0; d < D; d++) acc[d] = F::add(acc[d], F::mul(grp[d], z));
    }
#pragma unroll
    for (int d = 0; d < D; d++) p.out[step * D + d] = acc[d];
}

// ---- host side --------------------------------------------------------------------------------------------------------
template <class HF>
struct HostOps {
    typedef typename HF::T T;
    static T modulus();
};
template <> HostF64::T HostOps<HostF64>::modulus() { return gl::P; }
template <> HostF128::T HostOps<HostF128>::modulus() { return f128::modulus(); }
template <> HostF62::T HostOps<HostF62>::modulus() { return f62::M; }

template <class HF>
static typename HF::T hadd(typename HF::T a, typename HF::T b) {
    const typename HF::T m = HostOps<HF>::modulus();
    const typename HF::T s = a + b;            // may wrap for f128: detect
    if (s < a || s >= m) return s - m;
    return s;
}
template <class HF>
static typename HF::T hsub(typename HF::T a, typename HF::T b) {
    return a >= b ? a - b : a + (HostOps<HF>::modulus() - b);
}

// periodic column values of the AIR (canonical integers), NP columns of CYCLE values; only the Rescue example has any
template <class HF, class AIR>
static void periodic_values(std::vector<std::vector<typename HF::T>> &cols) { cols.clear(); }
template <> void periodic_values<HostF128, AirRescue>(std::vector<std::vector<HostF128::T>> &cols) {
    cols.assign(9, std::vector<HostF128::T>(16));
    for (int i = 0; i < 16; i++) cols[0][i] = i < 14 ? 1 : 0;                 // CYCLE_MASK, examples/src/rescue/air.rs:18-35
    for (int j = 0; j < 8; j++)
        for (int i = 0; i < 16; i++) cols[1 + j][i] = RESCUE_ARK[i][j];      // get_round_constants, rescue.rs:92-107
}

template <> void periodic_values<HostF128, AirRescueRaps>(std::vector<std::vector<HostF128::T>> &cols) {
    cols.assign(10, std::vector<HostF128::T>(16));
    for (int i = 0; i < 16; i++) cols[0][i] = i < 14 ? 1 : 0;                 // CYCLE_MASK, examples/src/rescue_raps/air.rs:22-39
    for (int i = 0; i < 16; i++) cols[1][i] = i == 14 ? 1 : 0;                // the absorption column, air.rs:243-245
    for (int j = 0; j < 8; j++)
        for (int i = 0; i < 16; i++) cols[2 + j][i] = RESCUE_ARK[i][j];
}

template <class HF, class AIR>
static void fill_consts(typename AIR::Consts &c) {
    if constexpr (std::is_same<AIR, AirRescue>::value || std::is_same<AIR, AirRescueRaps>::value) {
        for (int i = 0; i < 16; i++) {
            c.mds[i] = RESCUE_MDS[i];
            c.inv_mds[i] = RESCUE_INV_MDS[i];
        }
    } else if constexpr (std::is_same<AIR, AirVdf<1>>::value || std::is_same<AIR, AirVdf<2>>::value) {
        c.forty_two = (f128::u128)HF::to_internal(HF::from_u64(42));
    } else {
        (void)c;
    }
}

// the auxiliary segment's side of a call (all NULL / 0 for a single-segment AIR)
struct AuxArgs {
    const void *d_lde = nullptr;
    uint64_t row_width = 0;                 // base elements per row
    uint32_t num_assert = 0;
    const uint32_t *h_cols = nullptr;
    const uint64_t *h_steps = nullptr;
    const void *h_vals = nullptr, *h_cc = nullptr, *h_rand = nullptr;
};

// assertions of every kind against the main segment (wf_evaluate_constraints_assertions): per assertion the stride (0 = single value)
// and the number of values; h_vals then holds all values back to back.  Null pointers: every assertion is Assertion::single.
struct MultiArgs {
    const uint64_t *h_strides = nullptr, *h_nvals = nullptr;
    // composition coefficients already on the device (wf_evaluate_constraints_dev: where wf_coin_draw put them) instead of h_cc_*
    const void *d_cc_t = nullptr, *d_cc_b = nullptr;
};

// pool blocks that go back to the context's pool on every way out (stream-ordered: safe right after the launches that use them)
struct PoolBlocks {
    wf_ctx *ctx;
    std::vector<void *> v;
    ~PoolBlocks() { for (void *p : v) (void)wf_free(ctx, p); }
};

template <class HF, class AIR, int D>
static int evaluate(wf_ctx *ctx, const void *d_lde, uint64_t row_width, uint32_t log_n, uint32_t log_lde_blowup, uint32_t log_ce_blowup,
                    const void *h_offset, const void *h_cc_t, uint32_t num_assert, const uint32_t *h_cols, const uint64_t *h_steps,
                    const void *h_vals, const void *h_cc_b, void *d_out, const AuxArgs &aux = AuxArgs(), const MultiArgs &multi = MultiArgs()) {
    typedef typename HF::T T;
    typedef typename HF::Dev F;
    if ((AIR::AUX_WIDTH > 0) != (aux.d_lde != nullptr)) return WF_ERR_INVALID_ARG;     // a multi-segment AIR comes with its aux segment, the others without
    if (AIR::AUX_WIDTH > 0 && (aux.row_width < (uint64_t)AIR::AUX_WIDTH * D || !aux.h_rand || aux.num_assert > MAX_ASSERT ||
                               (aux.num_assert && (!aux.h_cols || !aux.h_steps || !aux.h_vals || !aux.h_cc))))
        return WF_ERR_INVALID_ARG;
    if (log_ce_blowup != (uint32_t)AIR::LOG_CE) return WF_ERR_INVALID_ARG;          // AirContext fixes ce_blowup (context.rs:104-117)
    if (log_lde_blowup < log_ce_blowup) return WF_ERR_INVALID_ARG;                  // "blowup factor too small" (context.rs:119-124)
    if (row_width < (uint64_t)AIR::WIDTH || num_assert == 0 || num_assert > MAX_ASSERT) return WF_ERR_INVALID_ARG;
    if (log_n + log_lde_blowup > HF::TWO_ADICITY || log_n < 3) return WF_ERR_DOMAIN_TOO_LARGE;
    if (AIR::CYCLE && (1ull << log_n) < (uint64_t)AIR::CYCLE) return WF_ERR_INVALID_ARG;
    const uint64_t n = 1ull << log_n, ce = n << log_ce_blowup;
    const uint32_t ce_blowup = 1u << log_ce_blowup, log_ce = log_n + log_ce_blowup;
    T off;
    WF_TRY(wf_load_offset<HF>(h_offset, &off));
    const T g_ce = HF::root_of_unity(log_ce), g_trace = HF::root_of_unity(log_n);
    const T one_c = HF::from_u64(1);

    // boundary groups by (asserted first step, stride): BoundaryConstraints::new groups by divisor (air/src/air/boundary/mod.rs:168-186)
    uint64_t gsteps[MAX_GROUPS], gstride[MAX_GROUPS];
    GroupLogK glogk{};
    uint32_t ngroups = 0;
    std::vector<uint32_t> a_group(num_assert), a_col(num_assert), a_seq(num_assert, ~0u);
    std::vector<T> a_val(num_assert), cc_b((size_t)num_assert * D), cc_t((size_t)(AIR::NT + AIR::NTA) * D);
    std::vector<uint32_t> x_group(aux.num_assert + 1), x_col(aux.num_assert + 1);
    std::vector<T> x_val((size_t)aux.num_assert * D + 1), cc_x((size_t)aux.num_assert * D + 1), rnd((size_t)AIR::NR * D + 1);
    // sequence assertions: the value polynomial (inverse FFT of the values, BoundaryConstraint::new, constraint.rs:60-91) evaluated at
    // x g^(-first_step) over the whole constraint-evaluation domain = one coset LDE of 2^log_k coefficients per assertion, on the device
    T *d_seq = nullptr;
    uint32_t nseq = 0;
    PoolBlocks pool{ctx, {}};
    uint64_t voff = 0;
    for (uint32_t k = 0; k < num_assert; k++) {
        const uint64_t stride = multi.h_strides ? multi.h_strides[k] : 0, nvals = multi.h_nvals ? multi.h_nvals[k] : 1;
        if (h_cols[k] >= (uint32_t)AIR::WIDTH || h_steps[k] >= n) return WF_ERR_INVALID_ARG;
        // air/src/air/assertions/mod.rs:84-120: the stride is a power of two >= 2 and <= trace length, first_step < stride; one value
        // (periodic) or trace_length / stride of them (sequence)
        if (stride == 0 ? nvals != 1 : ((stride & (stride - 1)) || stride < 2 || stride > n || h_steps[k] >= stride || (nvals != 1 && nvals != n / stride)))
            return WF_ERR_INVALID_ARG;
        uint32_t q = 0;
        while (q < ngroups && (gsteps[q] != h_steps[k] || gstride[q] != stride)) q++;
        if (q == ngroups) {
            if (ngroups == MAX_GROUPS) return WF_ERR_UNSUPPORTED;
            gsteps[ngroups] = h_steps[k];
            gstride[ngroups] = stride;
            uint32_t lk = 0;
            while (stride && (stride << lk) < n) lk++;
            glogk.v[ngroups++] = lk;                                            // k = n / stride asserted steps (1 for a single value)
        }
        a_group[k] = q;
        a_col[k] = h_cols[k];
        memcpy((void *)&a_val[k], (const uint8_t *)h_vals + (size_t)voff * sizeof(T), sizeof(T));
        if (!HF::valid_internal(a_val[k])) return WF_ERR_INVALID_ARG;
        a_val[k] = HF::to_internal(HF::from_internal(a_val[k]));               // normalise lazy f62 words
        if (nvals > 1) a_seq[k] = nseq++;
        voff += nvals;
    }
    if (nseq) {
        void *blk;
        WF_TRY(wf_malloc(ctx, (size_t)nseq * ce * sizeof(T), &blk));
        pool.v.push_back(blk);
        d_seq = (T *)blk;
        const T g_inv = HF::invmod(g_trace);
        voff = 0;
        for (uint32_t k = 0; k < num_assert; k++) {
            const uint64_t nvals = multi.h_nvals[k];
            if (nvals > 1) {
                uint32_t lk = 0;
                while ((1ull << lk) < nvals) lk++;
                std::vector<T> vals(nvals);
                memcpy((void *)vals.data(), (const uint8_t *)h_vals + (size_t)voff * sizeof(T), nvals * sizeof(T));
                for (auto &v : vals) { if (!HF::valid_internal(v)) return WF_ERR_INVALID_ARG; v = HF::to_internal(HF::from_internal(v)); }
                void *d_poly;
                WF_TRY(wf_malloc(ctx, nvals * sizeof(T), &d_poly));
                pool.v.push_back(d_poly);
                // through a page-locked stage slot of the context (copied there before the call returns): no wait for the stream unless
                // the values exceed a slot (64 KiB), in which case the bounce-buffer copy synchronises
                WF_TRY(wf_copy_h2d_small_async(ctx, d_poly, vals.data(), nvals * sizeof(T)));
                WF_TRY(wf_fft_interpolate_poly(ctx, HF::Dev::ID, 1, d_poly, lk, 1));
                // b(x g^(-first_step)) for x = offset g_ce^i: the evaluations of the polynomial over the coset offset g^(-first_step) <g_ce>
                const T shifted = HF::to_internal(HF::mulmod(off, HF::powmod(g_inv, h_steps[k])));
                WF_TRY(wf_fft_evaluate_poly_with_offset(ctx, HF::Dev::ID, 1, d_poly, lk, &shifted, log_ce - lk, d_seq + (size_t)a_seq[k] * ce));
            }
            voff += nvals;
        }
    }
    for (uint32_t k = 0; k < aux.num_assert; k++) {                            // aux assertions join the group of their step (boundary.rs:58-73)
        if (aux.h_cols[k] >= (uint32_t)AIR::AUX_WIDTH || aux.h_steps[k] >= n) return WF_ERR_INVALID_ARG;
        uint32_t q = 0;
        while (q < ngroups && (gsteps[q] != aux.h_steps[k] || gstride[q] != 0)) q++;
        if (q == ngroups) {
            if (ngroups == MAX_GROUPS) return WF_ERR_UNSUPPORTED;
            gsteps[ngroups] = aux.h_steps[k];
            gstride[ngroups] = 0;
            glogk.v[ngroups++] = 0;
        }
        x_group[k] = q;
        x_col[k] = aux.h_cols[k];
    }
    if (AIR::AUX_WIDTH > 0) {
        memcpy((void *)x_val.data(), aux.h_vals, (size_t)aux.num_assert * D * sizeof(T));
        memcpy((void *)cc_x.data(), aux.h_cc, (size_t)aux.num_assert * D * sizeof(T));
        memcpy((void *)rnd.data(), aux.h_rand, (size_t)AIR::NR * D * sizeof(T));
        x_val.back() = cc_x.back() = rnd.back() = 0;
        for (auto *vec : {&x_val, &cc_x, &rnd})
            for (auto &v : *vec) { if (!HF::valid_internal(v)) return WF_ERR_INVALID_ARG; v = HF::to_internal(HF::from_internal(v)); }
    }
    const bool dev_cc = multi.d_cc_t != nullptr;      // the coefficients never visit the host (a device coin drew them: canonical words)
    if (dev_cc != (multi.d_cc_b != nullptr) || (!dev_cc && (!h_cc_b || !h_cc_t))) return WF_ERR_INVALID_ARG;
    if (!dev_cc) {
        memcpy((void *)cc_b.data(), h_cc_b, cc_b.size() * sizeof(T));
        memcpy((void *)cc_t.data(), h_cc_t, cc_t.size() * sizeof(T));
        for (auto &v : cc_b) { if (!HF::valid_internal(v)) return WF_ERR_INVALID_ARG; v = HF::to_internal(HF::from_internal(v)); }
        for (auto &v : cc_t) { if (!HF::valid_internal(v)) return WF_ERR_INVALID_ARG; v = HF::to_internal(HF::from_internal(v)); }
    }

    // transition divisor: inverse of x^n - 1 over its ce_blowup distinct values (get_inv_evaluation), exemption g^(n-1)
    std::vector<T> zt(ce_blowup);
    {
        const T off_n = HF::powmod(off, n);
        const T g_b = HF::powmod(g_ce, n);                                      // g_ce^n: a ce_blowup-th root of unity
        T cur = off_n;
        for (uint32_t i = 0; i < ce_blowup; i++) {
            zt[i] = HF::to_internal(HF::invmod(hsub<HF>(cur, one_c)));
            cur = HF::mulmod(cur, g_b);
        }
    }
    const T exempt = HF::to_internal(HF::powmod(g_trace, n - 1));
    std::vector<T> bvals(ngroups);
    for (uint32_t q = 0; q < ngroups; q++)      // g^(first_step k): get_trace_domain_value_at(trace_length, num_steps * first_step), divisor.rs:88-96
        bvals[q] = HF::to_internal(HF::powmod(g_trace, (gsteps[q] << glogk.v[q]) & (n - 1)));

    // periodic table (periodic_table.rs:24-75): column polynomial (inverse DFT of the cycle values) evaluated at
    // offset^(n/cycle) * g_plen^i, i < plen = cycle * ce_blowup; laid out [i][column]
    std::vector<std::vector<T>> pcols;
    periodic_values<HF, AIR>(pcols);
    const uint32_t plen = (uint32_t)AIR::CYCLE * ce_blowup;
    std::vector<T> ptab((size_t)plen * AIR::NP + 1);
    if (AIR::NP > 0) {
        const uint32_t cyc = AIR::CYCLE;
        uint32_t log_cyc = 0;
        while ((1u << log_cyc) < cyc) log_cyc++;
        const T w_inv = HF::invmod(HF::root_of_unity(log_cyc)), cyc_inv = HF::invmod(HF::from_u64(cyc));
        uint32_t log_plen = 0;
        while ((1u << log_plen) < plen) log_plen++;
        const T g_p = HF::root_of_unity(log_plen), off_c = HF::powmod(off, n / cyc);
        for (int k = 0; k < AIR::NP; k++) {
            std::vector<T> poly(cyc);
            for (uint32_t j = 0; j < cyc; j++) {                                // coefficient j = (1/cyc) sum_i v_i w^(-ij)
                T s = 0;
                const T wj = HF::powmod(w_inv, j);
                T wij = one_c;
                for (uint32_t i = 0; i < cyc; i++) {
                    s = hadd<HF>(s, HF::mulmod(pcols[k][i], wij));
                    wij = HF::mulmod(wij, wj);
                }
                poly[j] = HF::mulmod(s, cyc_inv);
            }
            T x = off_c;
            for (uint32_t i = 0; i < plen; i++) {
                T acc = 0;
                for (uint32_t j = cyc; j-- > 0;) acc = hadd<HF>(HF::mulmod(acc, x), poly[j]);
                ptab[(size_t)i * AIR::NP + k] = HF::to_internal(acc);
                x = HF::mulmod(x, g_p);
            }
        }
    }

    // device staging: one scratch block
    const size_t w_zb = (size_t)ngroups * ce, w_small = ptab.size() + zt.size() + bvals.size() + a_val.size() + cc_b.size() + cc_t.size() +
                                                        x_val.size() + cc_x.size() + rnd.size();
    const size_t bytes = (w_zb + w_small) * sizeof(T) + (3 * (size_t)num_assert + 2 * x_col.size()) * sizeof(uint32_t) + 64;
    void *tmp;
    WF_TRY(wf_scratch(ctx, 0, bytes, &tmp));
    T *d_zb = (T *)tmp, *d_ptab = d_zb + w_zb, *d_zt = d_ptab + ptab.size(), *d_b = d_zt + zt.size(), *d_aval = d_b + bvals.size(),
      *d_ccb = d_aval + a_val.size(), *d_cct = d_ccb + cc_b.size(), *d_xval = d_cct + cc_t.size(), *d_ccx = d_xval + x_val.size(),
      *d_rnd = d_ccx + cc_x.size();
    uint32_t *d_acol = (uint32_t *)(d_rnd + rnd.size()), *d_agrp = d_acol + num_assert, *d_xcol = d_agrp + num_assert, *d_xgrp = d_xcol + x_col.size(),
             *d_aseq = d_xgrp + x_col.size();
    WfUploadBatch batch(ctx, d_ptab);             // everything from d_ptab on is host data: one staged copy
    auto up = [&](void *dst, const void *src, size_t nbytes) { batch.add(dst, src, nbytes); };
    up(d_ptab, ptab.data(), ptab.size() * sizeof(T));
    up(d_zt, zt.data(), zt.size() * sizeof(T));
    up(d_b, bvals.data(), bvals.size() * sizeof(T));
    up(d_aval, a_val.data(), a_val.size() * sizeof(T));
    up(d_ccb, cc_b.data(), cc_b.size() * sizeof(T));
    up(d_cct, cc_t.data(), cc_t.size() * sizeof(T));
    up(d_acol, a_col.data(), num_assert * sizeof(uint32_t));
    up(d_agrp, a_group.data(), num_assert * sizeof(uint32_t));
    up(d_aseq, a_seq.data(), num_assert * sizeof(uint32_t));
    if (AIR::AUX_WIDTH > 0) {
        up(d_xval, x_val.data(), x_val.size() * sizeof(T));
        up(d_ccx, cc_x.data(), cc_x.size() * sizeof(T));
        up(d_rnd, rnd.data(), rnd.size() * sizeof(T));
        up(d_xcol, x_col.data(), x_col.size() * sizeof(uint32_t));
        up(d_xgrp, x_group.data(), x_group.size() * sizeof(uint32_t));
    }
    // the host vectors die with this frame: flush() waits for the stream; with the coefficients on the device the caller is queueing
    // a chain (prove() against a device coin), so the image goes through a page-locked slot of the context and nothing waits
    WF_TRY(dev_cc ? batch.flush_async() : batch.flush());

    SeriesTable xs;
    WF_TRY(wf_get_series_table<HF>(ctx, g_ce, off, log_ce, &xs));
    const T m2 = HostOps<HF>::modulus() - 2;
    const uint64_t e_lo = (uint64_t)m2, e_hi = sizeof(T) > 8 ? (uint64_t)((unsigned __int128)m2 >> 64) : 0;
    const T one_i = HF::to_internal(one_c);
    {
        const uint64_t lanes = (ce + INV_CHUNK - 1) / INV_CHUNK;
        wf_prof_begin(ctx, "divisor_inv");
        hipLaunchKernelGGL((divisor_inv_kernel<F>), dim3((uint32_t)((lanes + 255) / 256), ngroups), dim3(256), 0, ctx->stream,
                           (const T *)xs.d_lo, (const T *)xs.d_hi, xs.log_lo, ce, (const T *)d_b, one_i, e_lo, e_hi, glogk, d_zb);
        wf_prof_end(ctx);
        WF_HIP(hipGetLastError());
    }
    EvalParams<T, D> p;
    p.lde = (const T *)d_lde;
    p.row_width = row_width;
    p.log_n = log_n;
    p.log_lde_blowup = log_lde_blowup;
    p.log_ce_blowup = log_ce_blowup;
    p.x_lo = (const T *)xs.d_lo;
    p.x_hi = (const T *)xs.d_hi;
    p.x_log_lo = xs.log_lo;
    p.ptab = d_ptab;
    p.zt = d_zt;
    p.exempt = exempt;
    p.exempt2 = HF::to_internal(HF::powmod(g_trace, n - 2));
    p.zb = d_zb;
    p.cc_t = dev_cc ? (const T *)multi.d_cc_t : d_cct;
    p.num_assert = num_assert;
    p.ngroups = ngroups;
    p.a_col = d_acol;
    p.a_group = d_agrp;
    p.a_val = d_aval;
    p.a_seq = d_aseq;
    p.seq = d_seq;
    p.cc_b = dev_cc ? (const T *)multi.d_cc_b : d_ccb;
    p.aux_lde = (const T *)aux.d_lde;
    p.aux_row_width = aux.row_width;
    p.rand = d_rnd;
    p.num_aux_assert = aux.num_assert;
    p.x_col = d_xcol;
    p.x_group = d_xgrp;
    p.x_val = d_xval;
    p.cc_x = d_ccx;
    p.out = (T *)d_out;
    typename AIR::Consts consts;
    fill_consts<HF, AIR>(consts);
    wf_prof_begin(ctx, "evaluate_constraints");
    hipLaunchKernelGGL((constraints_kernel<F, AIR, D>), dim3((uint32_t)((ce + 255) / 256)), dim3(256), 0, ctx->stream, p, consts);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

template <class HF, class AIR>
static int evaluate_d(wf_ctx *ctx, uint32_t D, const void *d_lde, uint64_t row_width, uint32_t log_n, uint32_t log_lde_blowup,
                      uint32_t log_ce_blowup, const void *h_offset, const void *h_cc_t, uint32_t num_assert, const uint32_t *h_cols,
                      const uint64_t *h_steps, const void *h_vals, const void *h_cc_b, void *d_out, const AuxArgs &aux = AuxArgs(),
                      const MultiArgs &multi = MultiArgs()) {
    if (D == 1) return evaluate<HF, AIR, 1>(ctx, d_lde, row_width, log_n, log_lde_blowup, log_ce_blowup, h_offset, h_cc_t, num_assert, h_cols, h_steps, h_vals, h_cc_b, d_out, aux, multi);
    if (D == 2) return evaluate<HF, AIR, 2>(ctx, d_lde, row_width, log_n, log_lde_blowup, log_ce_blowup, h_offset, h_cc_t, num_assert, h_cols, h_steps, h_vals, h_cc_b, d_out, aux, multi);
    if constexpr (HF::Dev::MAX_EXT >= 3)
        if (D == 3) return evaluate<HF, AIR, 3>(ctx, d_lde, row_width, log_n, log_lde_blowup, log_ce_blowup, h_offset, h_cc_t, num_assert, h_cols, h_steps, h_vals, h_cc_b, d_out, aux, multi);
    return WF_ERR_UNSUPPORTED;
}

}  // namespace

static int evaluate_single_segment(wf_ctx *ctx, int air, int field, uint32_t ext_degree, const void *d_trace_lde, uint64_t row_width,
                                   uint32_t log_n, uint32_t log_lde_blowup, uint32_t log_ce_blowup, const void *h_domain_offset,
                                   const void *h_cc_transition, uint32_t num_assertions, const uint32_t *h_assert_columns,
                                   const uint64_t *h_assert_steps, const void *h_assert_values, const void *h_cc_boundary, void *d_out,
                                   const MultiArgs &multi) {
#define WF_EVAL(HF, AIR)                                                                                                          \
    return evaluate_d<HF, AIR>(ctx, ext_degree, d_trace_lde, row_width, log_n, log_lde_blowup, log_ce_blowup, h_domain_offset,    \
                               h_cc_transition, num_assertions, h_assert_columns, h_assert_steps, h_assert_values, h_cc_boundary, d_out, AuxArgs(), multi)
#define WF_EVAL_ANY_FIELD(AIR)                                  \
    switch (field) {                                            \
        case WF_FIELD_F64: WF_EVAL(HostF64, AIR);               \
        case WF_FIELD_F128: WF_EVAL(HostF128, AIR);             \
        case WF_FIELD_F62: WF_EVAL(HostF62, AIR);               \
        default: return WF_ERR_UNSUPPORTED;                     \
    }
    if (air == WF_AIR_FIB_SMALL) { WF_EVAL_ANY_FIELD(AirFibSmall) }
    if (air == WF_AIR_FIB8) { WF_EVAL_ANY_FIELD(AirFib8) }
    if (air == WF_AIR_MULFIB2) { WF_EVAL_ANY_FIELD(AirMulFib2) }
    if (air == WF_AIR_MULFIB8) { WF_EVAL_ANY_FIELD(AirMulFib8) }
    if (air == WF_AIR_VDF) { WF_EVAL_ANY_FIELD(AirVdf<1>) }
    if (air == WF_AIR_VDF_EXEMPT) { WF_EVAL_ANY_FIELD(AirVdf<2>) }
#undef WF_EVAL_ANY_FIELD
    if (air == WF_AIR_RESCUE) {
        if (field != WF_FIELD_F128) return WF_ERR_UNSUPPORTED;   // the example's constants live in f128 (examples/src/rescue/rescue.rs:8)
        WF_EVAL(HostF128, AirRescue);
    }
#undef WF_EVAL
    return WF_ERR_UNSUPPORTED;
}

extern "C" int wf_evaluate_constraints(wf_ctx *ctx, int air, int field, uint32_t ext_degree, const void *d_trace_lde, uint64_t row_width,
                                       uint32_t log_n, uint32_t log_lde_blowup, uint32_t log_ce_blowup, const void *h_domain_offset,
                                       const void *h_cc_transition, uint32_t num_assertions, const uint32_t *h_assert_columns,
                                       const uint64_t *h_assert_steps, const void *h_assert_values, const void *h_cc_boundary,
                                       void *d_out) {
    WF_ENTER(ctx);
    if (!d_trace_lde || !h_domain_offset || !h_cc_transition || !h_assert_columns || !h_assert_steps || !h_assert_values || !h_cc_boundary ||
        !d_out)
        return WF_ERR_INVALID_ARG;
    return evaluate_single_segment(ctx, air, field, ext_degree, d_trace_lde, row_width, log_n, log_lde_blowup, log_ce_blowup, h_domain_offset,
                                   h_cc_transition, num_assertions, h_assert_columns, h_assert_steps, h_assert_values, h_cc_boundary, d_out,
                                   MultiArgs());
}

extern "C" int wf_evaluate_constraints_assertions(wf_ctx *ctx, int air, int field, uint32_t ext_degree, const void *d_trace_lde,
                                                  uint64_t row_width, uint32_t log_n, uint32_t log_lde_blowup, uint32_t log_ce_blowup,
                                                  const void *h_domain_offset, const void *h_cc_transition, uint32_t num_assertions,
                                                  const uint32_t *h_assert_columns, const uint64_t *h_assert_first_steps,
                                                  const uint64_t *h_assert_strides, const uint64_t *h_assert_num_values,
                                                  const void *h_assert_values, const void *h_cc_boundary, void *d_out) {
    WF_ENTER(ctx);
    if (!d_trace_lde || !h_domain_offset || !h_cc_transition || !h_assert_columns || !h_assert_first_steps || !h_assert_strides ||
        !h_assert_num_values || !h_assert_values || !h_cc_boundary || !d_out)
        return WF_ERR_INVALID_ARG;
    MultiArgs multi;
    multi.h_strides = h_assert_strides;
    multi.h_nvals = h_assert_num_values;
    return evaluate_single_segment(ctx, air, field, ext_degree, d_trace_lde, row_width, log_n, log_lde_blowup, log_ce_blowup, h_domain_offset,
                                   h_cc_transition, num_assertions, h_assert_columns, h_assert_first_steps, h_assert_values, h_cc_boundary, d_out,
                                   multi);
}

extern "C" int wf_evaluate_constraints_dev(wf_ctx *ctx, int air, int field, uint32_t ext_degree, const void *d_trace_lde, uint64_t row_width,
                                          uint32_t log_n, uint32_t log_lde_blowup, uint32_t log_ce_blowup, const void *h_domain_offset,
                                          const void *d_cc_transition, uint32_t num_assertions, const uint32_t *h_assert_columns,
                                          const uint64_t *h_assert_first_steps, const uint64_t *h_assert_strides,
                                          const uint64_t *h_assert_num_values, const void *h_assert_values, const void *d_cc_boundary,
                                          void *d_out) {
    WF_ENTER(ctx);
    if (!d_trace_lde || !h_domain_offset || !d_cc_transition || !h_assert_columns || !h_assert_first_steps || !h_assert_values ||
        !d_cc_boundary || !d_out || ((h_assert_strides == nullptr) != (h_assert_num_values == nullptr)))
        return WF_ERR_INVALID_ARG;
    MultiArgs multi;
    multi.h_strides = h_assert_strides;
    multi.h_nvals = h_assert_num_values;
    multi.d_cc_t = d_cc_transition;
    multi.d_cc_b = d_cc_boundary;
    return evaluate_single_segment(ctx, air, field, ext_degree, d_trace_lde, row_width, log_n, log_lde_blowup, log_ce_blowup, h_domain_offset,
                                   nullptr, num_assertions, h_assert_columns, h_assert_first_steps, h_assert_values, nullptr, d_out, multi);
}

extern "C" int wf_evaluate_constraints_aux(wf_ctx *ctx, int air, int field, uint32_t ext_degree, const void *d_main_lde, uint64_t main_row_width,
                                           const void *d_aux_lde, uint64_t aux_row_width, uint32_t log_n, uint32_t log_lde_blowup,
                                           uint32_t log_ce_blowup, const void *h_domain_offset, const void *h_cc_transition,
                                           uint32_t num_assertions, const uint32_t *h_assert_columns, const uint64_t *h_assert_steps,
                                           const void *h_assert_values, const void *h_cc_boundary, uint32_t num_aux_assertions,
                                           const uint32_t *h_aux_assert_columns, const uint64_t *h_aux_assert_steps,
                                           const void *h_aux_assert_values, const void *h_cc_aux_boundary, const void *h_aux_rand_elements,
                                           void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_main_lde || !d_aux_lde || !h_domain_offset || !h_cc_transition || !h_assert_columns || !h_assert_steps ||
        !h_assert_values || !h_cc_boundary || !h_aux_rand_elements || !d_out)
        return WF_ERR_INVALID_ARG;
    AuxArgs aux;
    aux.d_lde = d_aux_lde;
    aux.row_width = aux_row_width;
    aux.num_assert = num_aux_assertions;
    aux.h_cols = h_aux_assert_columns;
    aux.h_steps = h_aux_assert_steps;
    aux.h_vals = h_aux_assert_values;
    aux.h_cc = h_cc_aux_boundary;
    aux.h_rand = h_aux_rand_elements;
    if (air == WF_AIR_RESCUE_RAPS) {
        if (field != WF_FIELD_F128) return WF_ERR_UNSUPPORTED;   // examples/src/rescue_raps/mod.rs:13: f128
        return evaluate_d<HostF128, AirRescueRaps>(ctx, ext_degree, d_main_lde, main_row_width, log_n, log_lde_blowup, log_ce_blowup,
                                                   h_domain_offset, h_cc_transition, num_assertions, h_assert_columns, h_assert_steps,
                                                   h_assert_values, h_cc_boundary, d_out, aux);
    }
    return WF_ERR_UNSUPPORTED;                                   // the other built-in AIRs have no auxiliary segment
}
