// Lane-cooperative Rescue-Prime permutations for SMALL batches (Rp64_256 / Rp62_248 width 12, RpJive64_256 width 8).
//
// rp64.cuh / rpjive64.cuh keep one whole state per lane: best throughput, but a permutation is ~6400 dependent-ish
// modmuls, i.e. ~0.2 ms of one wave no matter how few hashes are in flight.  The upper levels of every Merkle tree, the
// late FRI layers and the Fiat-Shamir coin's single hashes are exactly that case: a launch with a handful of waves that
// each take 0.2 ms (a 2^23-leaf Rp64 tree spends ~3 ms in its top 14 levels, a 2^9-row FRI layer is all latency).
// Here a state is spread over a 16-lane group, ONE WORD PER LANE (12 or 8 lanes active): the S-boxes (152 of the ~160
// multiplication levels of a round) are lane-local, and the circulant MDS row  out_i = sum_k ROW[k] * x[(i + k) mod W]
// reads the group's words back from LDS (one ds_write + W ds_reads, rotated addresses).  Latency per permutation drops
// by ~W; lane efficiency is W/16 plus the LDS round trip, so the batched kernels stay in use above COOP_MAX hashes.
// Results are bit-identical (same field operations on the same values; integer MDS reduced once, canonical).
//
// LDS ordering: the 16 lanes of a group sit in one wavefront and a wavefront's DS instructions execute in issue order,
// so a ds_write followed by ds_reads needs no barrier; `volatile` keeps the compiler from reordering or caching them.
#pragma once
#include "rp62.cuh"
#include "rpjive64.cuh"

namespace rcoop {

constexpr int GROUP = 16;               // lanes per state
constexpr uint64_t COOP_MAX = 1u << 14;  // use the cooperative kernels up to this many hashes per launch

// S-boxes and additions shared by the two Goldilocks instances
struct GlOps {
    static __device__ __forceinline__ uint64_t fwd(uint64_t s) { return rp64::exp7(s); }
    static __device__ __forceinline__ uint64_t inv(uint64_t s) {
        uint64_t t[1] = {s};
        rp64::inv_sbox<1>(t);
        return t[0];
    }
    static __device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) { return gl::add(a, b); }
};

struct Spec12 : GlOps {                  // Rp64_256: crypto/src/hash/rescue/rp64_256/mod.rs:390 (MDS), :741 / :842 (ARK1 / ARK2)
    static constexpr int W = 12;
    static __device__ __forceinline__ uint32_t row(int k) {
        constexpr uint32_t R[12] = {7, 23, 8, 26, 13, 10, 9, 7, 6, 22, 21, 8};
        return R[k];
    }
    static __device__ __forceinline__ uint64_t ark1(int r, int i) { return rp64::ARK1_T.v[r][i]; }
    static __device__ __forceinline__ uint64_t ark2(int r, int i) { return rp64::ARK2_T.v[r][i]; }
    static __device__ __forceinline__ uint64_t mds(uint64_t s, int i, int ii, volatile uint64_t *grp);
};
struct Spec8 : GlOps {                   // RpJive64_256: rp64_256_jive/mod.rs:417-499
    static constexpr int W = 8;
    static __device__ __forceinline__ uint32_t row(int k) {
        constexpr uint32_t R[8] = {23, 8, 13, 10, 7, 6, 21, 8};
        return R[k];
    }
    static __device__ __forceinline__ uint64_t ark1(int r, int i) { return rpj::ARK1_T.v[r][i]; }
    static __device__ __forceinline__ uint64_t ark2(int r, int i) { return rpj::ARK2_T.v[r][i]; }
    static __device__ __forceinline__ uint64_t mds(uint64_t s, int i, int ii, volatile uint64_t *grp);
};

// word `from` of the group's states, as seen after every lane published `s`
__device__ __forceinline__ uint64_t exchange(uint64_t s, int i, int from, volatile uint64_t *grp) {
    grp[i] = s;
    return grp[from];
}

// circulant MDS with small entries: exact integer row sum on the 32-bit halves, reduced once (as rp64::mds / rpj::mds)
template <class S>
__device__ __forceinline__ uint64_t mds_small(uint64_t s, int i, int ii, volatile uint64_t *grp) {
    grp[i] = s;
    uint64_t al = 0, ah = 0;
#pragma unroll
    for (int k = 0; k < S::W; k++) {
        int j = ii + k;
        if (j >= S::W) j -= S::W;
        const uint64_t x = grp[j];
        al += (uint64_t)S::row(k) * (uint32_t)x;
        ah += (uint64_t)S::row(k) * (uint32_t)(x >> 32);
    }
    const uint64_t low = al + (ah << 32);
    const uint32_t carry = low < al;
    return gl::reduce160(low, (uint32_t)(ah >> 32) + carry, 0);
}

// Rp62_248 (crypto/src/hash/rescue/rp62_248/mod.rs): f62 words in [0, 2M), x^3 / x^(1/3), dense 12 x 12 MDS applied in the
// batched kernel's order (acc = m_0 x_0, then + m_j x_j), so the lazily reduced representatives agree word for word
struct Spec62 {
    static constexpr int W = 12;
    static __device__ __forceinline__ uint64_t fwd(uint64_t s) { return f62::mul(f62::mul(s, s), s); }
    static __device__ __forceinline__ uint64_t inv(uint64_t s) {
        uint64_t t[1] = {s};
        rp62::inv_sbox_w<1>(t);
        return t[0];
    }
    static __device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) { return f62::add(a, b); }
    static __device__ __forceinline__ uint64_t ark1(int r, int i) { return rp62::TBL.ark1[r][i]; }
    static __device__ __forceinline__ uint64_t ark2(int r, int i) { return rp62::TBL.ark2[r][i]; }
    static __device__ __forceinline__ uint64_t mds(uint64_t s, int i, int ii, volatile uint64_t *grp) {
        grp[i] = s;
        uint64_t acc = f62::mul(rp62::TBL.mds[ii][0], grp[0]);
#pragma unroll
        for (int j = 1; j < 12; j++) acc = f62::add(acc, f62::mul(rp62::TBL.mds[ii][j], grp[j]));
        return acc;
    }
};

// i = lane within the group, ii = min(i, W - 1) (idle lanes shadow the last word so that every address stays in range)
__device__ __forceinline__ uint64_t Spec12::mds(uint64_t s, int i, int ii, volatile uint64_t *grp) { return mds_small<Spec12>(s, i, ii, grp); }
__device__ __forceinline__ uint64_t Spec8::mds(uint64_t s, int i, int ii, volatile uint64_t *grp) { return mds_small<Spec8>(s, i, ii, grp); }

template <class S>
__device__ __forceinline__ uint64_t permute(uint64_t s, int i, int ii, volatile uint64_t *grp) {
#pragma unroll 1
    for (int r = 0; r < 7; r++) {
        s = S::add(S::mds(S::fwd(s), i, ii, grp), S::ark1(r, ii));
        s = S::add(S::mds(S::inv(s), i, ii, grp), S::ark2(r, ii));
    }
    return s;
}

// ---- Rp64_256 ---------------------------------------------------------------------------------------------------------
struct CoopRp64 {
    typedef Spec12 S;
    // merge (rp64_256/mod.rs:181-192); lanes 4..7 return the digest words
    static __device__ __forceinline__ uint64_t merge(const uint64_t *pair, int i, int ii, volatile uint64_t *grp) {
        uint64_t s = 0;
        if (i == 0) s = rp64::mont_small(8);
        if (i >= 4 && i < 12) s = pair[i - 4];
        return permute<S>(s, i, ii, grp);
    }
    // hash_elements (mod.rs:224-257)
    static __device__ __forceinline__ uint64_t hash_elements(const uint64_t *p, uint32_t n, int i, int ii, volatile uint64_t *grp) {
        uint64_t s = i == 0 ? rp64::mont_small(n) : 0;
        for (uint32_t base = 0; base < n; base += 8) {
            const uint32_t idx = base + (uint32_t)(i - 4);
            if (i >= 4 && i < 12 && idx < n) s = gl::add(s, p[idx]);
            s = permute<S>(s, i, ii, grp);
        }
        return s;
    }
    // merge_with_int (mod.rs:198-219)
    static __device__ __forceinline__ uint64_t merge_with_int(const uint32_t (&seed)[8], uint64_t value, int i, int ii, volatile uint64_t *grp) {
        constexpr uint64_t R2 = 0xfffffffe00000001ull;
        const bool big = value >= gl::P;
        uint64_t s = 0;
        if (i == 0) s = rp64::mont_small(big ? 6 : 5);
        if (i >= 4 && i < 8) s = (uint64_t)seed[2 * (i - 4)] | ((uint64_t)seed[2 * (i - 4) + 1] << 32);
        if (i == 8) s = gl::mul(big ? value - gl::P : value, R2);
        if (i == 9 && big) s = rp64::mont_small(1);
        return permute<S>(s, i, ii, grp);
    }
    static __device__ __forceinline__ bool out_lane(int i) { return i >= 4 && i < 8; }
    static __device__ __forceinline__ int out_word(int i) { return i - 4; }
};

// ---- RpJive64_256 -----------------------------------------------------------------------------------------------------
struct CoopRpJive {
    typedef Spec8 S;
    // apply_jive_summation (mod.rs:355-369): lanes 0..3 return init_i + init_{i+4} + fin_i + fin_{i+4}
    static __device__ __forceinline__ uint64_t jive(uint64_t init, uint64_t fin, int i, volatile uint64_t *grp) {
        const int from = i < 4 ? i + 4 : i;
        const uint64_t init_hi = exchange(init, i, from, grp);
        const uint64_t fin_hi = exchange(fin, i, from, grp);
        return gl::add(gl::add(init, init_hi), gl::add(fin, fin_hi));
    }
    // merge (mod.rs:186-200)
    static __device__ __forceinline__ uint64_t merge(const uint64_t *pair, int i, int ii, volatile uint64_t *grp) {
        const uint64_t init = pair[ii];
        return jive(init, permute<S>(init, i, ii, grp), i, grp);
    }
    // hash_elements (mod.rs:268-313); lanes 4..7 hold the digest
    static __device__ __forceinline__ uint64_t hash_elements(const uint64_t *p, uint32_t n, int i, int ii, volatile uint64_t *grp) {
        const uint64_t one = rp64::mont_small(1);
        uint64_t s = (i == 0 && (n & 3)) ? one : 0;
        const bool rate = i >= 4 && i < 8;
        uint32_t base = 0;
        for (; base + 4 <= n; base += 4) {
            if (rate) s = gl::add(s, p[base + (uint32_t)(i - 4)]);
            s = permute<S>(s, i, ii, grp);
        }
        const uint32_t left = n - base;
        if (left) {
            if (rate) {
                const uint32_t j = (uint32_t)(i - 4);
                s = j < left ? gl::add(s, p[base + j]) : (j == left ? one : 0);
            }
            s = permute<S>(s, i, ii, grp);
        }
        // bring the digest to lanes 0..3 so that both Jive entry points write from the same lanes
        return exchange(s, i, i < 4 ? i + 4 : i, grp);
    }
    // merge_with_int (mod.rs:223-263)
    static __device__ __forceinline__ uint64_t merge_with_int(const uint32_t (&seed)[8], uint64_t value, int i, int ii, volatile uint64_t *grp) {
        constexpr uint64_t R2 = 0xfffffffe00000001ull;
        const bool big = value >= gl::P;
        uint64_t s = 0;
        if (i < 4) s = (uint64_t)seed[2 * i] | ((uint64_t)seed[2 * i + 1] << 32);
        if (i == 4) s = gl::mul(big ? value - gl::P : value, R2);
        if (i == 5 && big) s = rp64::mont_small(1);
        if (i == 7) s = rp64::mont_small(big ? 6 : 5);
        return jive(s, permute<S>(s, i, ii, grp), i, grp);
    }
    static __device__ __forceinline__ bool out_lane(int i) { return i < 4; }
    static __device__ __forceinline__ int out_word(int i) { return i; }
};

// ---- Rp62_248 ---------------------------------------------------------------------------------------------------------
struct CoopRp62 {
    typedef Spec62 S;
    // merge (rp62_248/mod.rs:155-170): the two digests in state[0..8], 8 in state[11]; digest = state[0..4]
    static __device__ __forceinline__ uint64_t merge(const uint64_t *pair, int i, int ii, volatile uint64_t *grp) {
        uint64_t s = 0;
        if (i < 8) s = f62::norm(pair[i]);
        if (i == 11) s = rp62::to_mont(8);
        return permute<S>(s, i, ii, grp);
    }
    // hash_elements (mod.rs:203-236)
    static __device__ __forceinline__ uint64_t hash_elements(const uint64_t *p, uint32_t n, int i, int ii, volatile uint64_t *grp) {
        uint64_t s = i == 11 ? rp62::to_mont(n) : 0;
        for (uint32_t base = 0; base < n; base += 8) {
            const uint32_t idx = base + (uint32_t)i;
            if (i < 8 && idx < n) s = f62::add(s, f62::norm(p[idx]));
            s = permute<S>(s, i, ii, grp);
        }
        return s;
    }
    // merge_with_int (mod.rs:172-201)
    static __device__ __forceinline__ uint64_t merge_with_int(const uint32_t (&seed)[8], uint64_t value, int i, int ii, volatile uint64_t *grp) {
        const bool big = value >= f62::M;
        uint64_t s = 0;
        if (i < 4) s = f62::norm((uint64_t)seed[2 * i] | ((uint64_t)seed[2 * i + 1] << 32));
        if (i == 4) s = rp62::to_mont(value % f62::M);
        if (i == 5 && big) s = rp62::to_mont(value / f62::M);
        if (i == 11) s = rp62::to_mont(big ? 6 : 5);
        return permute<S>(s, i, ii, grp);
    }
    static __device__ __forceinline__ bool out_lane(int i) { return i < 4; }
    static __device__ __forceinline__ int out_word(int i) { return i; }
};

// ---- kernels: 256 threads = 16 groups per workgroup ----------------------------------------------------------------------
#define RCOOP_PROLOGUE                                                  \
    __shared__ uint64_t lds[256];                                       \
    const int i = threadIdx.x & (GROUP - 1);                            \
    const int ii = i < C::S::W ? i : C::S::W - 1;                       \
    volatile uint64_t *grp = lds + (threadIdx.x & ~(GROUP - 1));        \
    const uint64_t g = (uint64_t)blockIdx.x * (256 / GROUP) + threadIdx.x / GROUP;

// out[g] = merge(pairs[2g], pairs[2g + 1])
template <class C>
__global__ __launch_bounds__(256) void merge_kernel(const uint64_t *pairs, uint64_t count, uint64_t *out) {
    RCOOP_PROLOGUE
    if (g >= count) return;
    const uint64_t d = C::merge(pairs + g * 8, i, ii, grp);
    if (C::out_lane(i)) out[g * 4 + C::out_word(i)] = d;
}

// leaf[r * parts + k] = hash_elements(words [k * part_elems, ...) of row r)   (the batched hash_rows_kernel's contract)
template <class C>
__global__ __launch_bounds__(256) void hash_rows_kernel(const uint64_t *rows, uint64_t num_rows, uint64_t row_width, uint32_t elems_per_row,
                                                        uint32_t part_elems, uint32_t parts, uint64_t *out) {
    RCOOP_PROLOGUE
    if (g >= num_rows) return;
    const uint32_t k = blockIdx.y;
    const uint32_t e0 = k * part_elems;
    const uint32_t e1 = (e0 + part_elems < elems_per_row) ? e0 + part_elems : elems_per_row;
    const uint64_t d = C::hash_elements(rows + g * row_width + e0, e1 - e0, i, ii, grp);
    if (C::out_lane(i)) out[(g * parts + k) * 4 + C::out_word(i)] = d;
}

struct SeedWords {
    uint32_t w[8];
};
template <class C>
__global__ __launch_bounds__(256) void merge_with_int_kernel(SeedWords seed, uint64_t first, uint64_t count, uint64_t *out) {
    RCOOP_PROLOGUE
    if (g >= count) return;
    const uint64_t d = C::merge_with_int(seed.w, first + g, i, ii, grp);
    if (C::out_lane(i)) out[g * 4 + C::out_word(i)] = d;
}
#undef RCOOP_PROLOGUE

}  // namespace rcoop
