// Parameters of one NTT pass launch and small helpers shared by the pass kernels (ntt_engine.cuh: radix <= 256, two steps;
// ntt_big.cuh: radix 1024 .. 4096, three steps).  No HIP runtime types here: tests/cpp/ntt_big_host_test.cpp compiles the
// three-step pass for the host.
#pragma once
#include <stdint.h>

template <class T>
struct PassParams {
    const T *src;
    T *dst;
    uint32_t log_n;
    uint32_t npass;
    uint32_t pass;
    uint32_t log_r[6];
    uint32_t nvec;
    uint32_t src_div, src_inner, dst_inner;
    uint64_t src_vec_stride, dst_vec_stride;
    uint64_t src_inner_stride, dst_inner_stride;
    uint32_t src_es, dst_es;
    uint32_t inverse;
    // tables
    const T *w_lo, *w_hi;
    uint32_t w_log_lo;
    const T *w256, *w16;
    const T *pre_lo, *pre_hi;
    uint32_t pre_log_lo, pre_mod;
    uint64_t pre_lo_stride, pre_hi_stride;
    const T *post_lo, *post_hi;
    uint32_t post_log_lo;
    uint32_t has_post_const;
    T post_const;
    const T *tw_tab;          // non-last pass: inter-pass twiddles T[k'][rem] when the table is small (else nullptr: progression)
    uint32_t tw_pair;         // tw_tab is in the field's table layout (F::TAB_WORDS words per entry: f128 pairs), small tables only
    uint32_t scale_in_w256;   // last pass of an inverse transform: w256 already carries the 1/n (applied for k_a = 0 too)
    // row-major output mode of the last pass (NttJob::rowmajor)
    uint32_t rowmajor, rm_log_b, rm_log_i, rm_base_cols;
    uint64_t rm_row_width;
    // rows + leaves mode of the last pass (NttJob::rh_leaves, f64 + Blake3_256, rows of one 8-column group)
    uint32_t rh_log_cp;
    void *rh_leaves;
    // three-step passes (ntt_big.cuh): omega_R^e, e < R (Montgomery residues), the twiddles between the first and the second step
    const T *big_tab;
    uint32_t bt;              // block-tile plan of a single transform: the last pass reads [(k1 S1 + a2) R0 + k0] (ntt_pass, BT0)
    uint64_t vt_tiles;        // number of tiles of a non-last vector-tile pass (the persistent variant walks them)
    uint32_t vt_cols;         // vector tiles (ntt_pass<..., VT>): the number C of interleaved columns of the VT buffers, 0 = off
    uint32_t coset_order;     // first pass of a coset LDE: the cosets of a source tile run side by side on one XCD (ntt_pass)
#ifdef WF_EXPERIMENTS
    // launch stagger (experiment, WF_NTT_STAGGER="ticks,mode"): the first workgroups of a launch start `generation` x ticks x 10 ns late
    uint32_t stagger_ticks, stagger_mode;
#endif
};

// q = x / d, r = x % d for a wave-uniform divisor that is almost always 1 or a power of two (blowup, extension degree):
// a runtime 64-bit division costs ~30-100 VALU instructions per lane, this costs a shift
__device__ __forceinline__ void divmod_uniform(uint32_t x, uint32_t d, uint32_t &q, uint32_t &r) {
    if (d == 1) {
        q = x;
        r = 0;
    } else if ((d & (d - 1)) == 0) {
        const uint32_t sh = 31u - (uint32_t)__builtin_clz(d);
        q = x >> sh;
        r = x & (d - 1);
    } else {
        q = x / d;
        r = x - q * d;
    }
}

