// Host-side construction + caching of the constant tables, generic over the field (HostF64 / HostF128).
#pragma once
#include <string.h>

#include "fields.cuh"
#include "wf_internal.h"

template <class T>
static inline void split128(T v, uint64_t &lo, uint64_t &hi) {
    lo = (uint64_t)v;
    if constexpr (sizeof(T) > 8) hi = (uint64_t)(v >> 64); else hi = 0;
}

template <class T>
static int wf_upload(wf_ctx *ctx, const std::vector<T> &h, void **d) {
    void *p;
    WF_TRY(wf_dev_malloc(ctx, &p, h.size() * sizeof(T)));
    ctx->owned.push_back(p);
    WF_TRY(wf_copy_h2d(ctx, p, h.data(), h.size() * sizeof(T)));   // synchronises: h goes out of scope in the caller
    *d = p;
    return WF_OK;
}

// series scale * base^i (canonical inputs), i < 2^log_len
template <class HF>
static int wf_get_series_table(wf_ctx *ctx, typename HF::T base, typename HF::T scale, uint32_t log_len, SeriesTable *out) {
    typedef typename HF::T T;
    uint64_t b0, b1, s0, s1;
    split128(base, b0, b1);
    split128(scale, s0, s1);
    SeriesKey key(HF::Dev::ID, b0, b1, s0, s1, log_len);
    auto it = ctx->series.find(key);
    if (it == ctx->series.end()) {
        SeriesTable t;
        t.log_len = log_len;
        t.log_lo = log_len < 12 ? log_len : 12;
        const uint64_t nlo = 1ull << t.log_lo, nhi = 1ull << (log_len - t.log_lo);
        std::vector<T> lo(nlo), hi(nhi);
        T cur = HF::from_u64(1);
        for (uint64_t i = 0; i < nlo; i++) {
            lo[i] = HF::to_internal(cur);
            cur = HF::mulmod(cur, base);
        }
        const T step = cur;
        cur = scale;
        for (uint64_t i = 0; i < nhi; i++) {
            hi[i] = HF::to_internal(cur);
            cur = HF::mulmod(cur, step);
        }
        WF_TRY(wf_upload(ctx, lo, &t.d_lo));
        WF_TRY(wf_upload(ctx, hi, &t.d_hi));
        it = ctx->series.emplace(key, t).first;
    }
    *out = it->second;
    return WF_OK;
}

template <class HF>
static int wf_get_omega_table(wf_ctx *ctx, uint32_t log_n, SeriesTable *out) {
    return wf_get_series_table<HF>(ctx, HF::root_of_unity(log_n), HF::from_u64(1), log_n, out);
}

// omega_256^e, e < 256 and omega_16^j, j < 8
template <class HF>
static int wf_get_small_tables(wf_ctx *ctx, void **w256, void **w16) {
    typedef typename HF::T T;
    const int id = HF::Dev::ID;
    if (!ctx->w256.count(id)) {
        std::vector<T> h(256), g(8);
        const T w = HF::root_of_unity(8);
        T cur = HF::from_u64(1);
        for (int i = 0; i < 256; i++) {
            h[i] = HF::to_internal(cur);
            if (i % 16 == 0 && i / 16 < 8) g[i / 16] = h[i];   // omega_16^j = omega_256^(16 j)
            cur = HF::mulmod(cur, w);
        }
        void *p;
        WF_TRY(wf_upload(ctx, h, &p));
        ctx->w256[id] = p;
        WF_TRY(wf_upload(ctx, g, &p));
        ctx->w16[id] = p;
    }
    *w256 = ctx->w256[id];
    *w16 = ctx->w16[id];
    return WF_OK;
}

// c * omega_256^e, e < 256 (c canonical)
template <class HF>
static int wf_get_scaled_w256(wf_ctx *ctx, typename HF::T c, void **out) {
    typedef typename HF::T T;
    uint64_t c0, c1;
    split128(c, c0, c1);
    const auto key = std::make_tuple((int)HF::Dev::ID, c0, c1);
    auto it = ctx->w256_scaled.find(key);
    if (it == ctx->w256_scaled.end()) {
        std::vector<T> h(256);
        const T w = HF::root_of_unity(8);
        T cur = c;
        for (int i = 0; i < 256; i++) {
            h[i] = HF::to_internal(cur);
            cur = HF::mulmod(cur, w);
        }
        void *p;
        WF_TRY(wf_upload(ctx, h, &p));
        it = ctx->w256_scaled.emplace(key, p).first;
    }
    *out = it->second;
    return WF_OK;
}

// NTT pass tables of a field whose table products take PAIRS (F::TAB_WORDS = 2, f128): entry e of w256 = (c omega_256^e, c omega_256^e 2^64),
// entry j of w16 = (omega_16^j, omega_16^j 2^64); c canonical (1 for the plain twiddles, 1/n for the last pass of an inverse transform)
template <class HF>
static int wf_get_pair_tables(wf_ctx *ctx, typename HF::T c, void **w256, void **w16) {
    typedef typename HF::T T;
    uint64_t c0, c1;
    split128(c, c0, c1);
    const auto key = std::make_tuple((int)HF::Dev::ID, c0, c1);
    auto it = ctx->pair_tables.find(key);
    if (it == ctx->pair_tables.end()) {
        std::vector<T> h(512), g(16);
        const T w = HF::root_of_unity(8), two64 = HF::from_u64(1ull << 32);
        const T t64 = HF::mulmod(two64, two64);
        T cur = c, plain = HF::from_u64(1);
        for (int i = 0; i < 256; i++) {
            h[2 * i] = HF::to_internal(cur);
            h[2 * i + 1] = HF::to_internal(HF::mulmod(cur, t64));
            if (i % 16 == 0 && i / 16 < 8) {                         // omega_16^j = omega_256^(16 j), never scaled
                g[2 * (i / 16)] = HF::to_internal(plain);
                g[2 * (i / 16) + 1] = HF::to_internal(HF::mulmod(plain, t64));
            }
            cur = HF::mulmod(cur, w);
            plain = HF::mulmod(plain, w);
        }
        void *p256, *p16;
        WF_TRY(wf_upload(ctx, h, &p256));
        WF_TRY(wf_upload(ctx, g, &p16));
        it = ctx->pair_tables.emplace(key, std::make_pair(p256, p16)).first;
    }
    *w256 = it->second.first;
    *w16 = it->second.second;
    return WF_OK;
}

// f64 passes (l24.cuh): rows of four plain-integer words  c * omega_256^e * T^k mod p,  k < 4, T = 2^24, e < 256 (c canonical;
// c = 1 for the plain intra-pass twiddles, 1/n for the last pass of an inverse transform)
template <class HF>
static int wf_get_w256_4form(wf_ctx *ctx, typename HF::T c, void **out) {
    typedef typename HF::T T;
    static_assert(sizeof(T) == 8, "four-word twiddle rows exist for the 64-bit Goldilocks field only");
    const auto key = std::make_tuple((int)HF::Dev::ID, (uint64_t)c, (uint64_t)0);
    auto it = ctx->w256_4form.find(key);
    if (it == ctx->w256_4form.end()) {
        std::vector<T> h(256 * 4);
        const T w = HF::root_of_unity(8);
        T cur = c;
        for (int i = 0; i < 256; i++) {
            T f = cur;
            for (int k = 0; k < 4; k++) {
                h[4 * i + k] = f;
                f = HF::mulmod(f, (T)1 << 24);
            }
            cur = HF::mulmod(cur, w);
        }
        void *p;
        WF_TRY(wf_upload(ctx, h, &p));
        it = ctx->w256_4form.emplace(key, p).first;
    }
    *out = it->second;
    return WF_OK;
}

// three-step passes (ntt_big.cuh): omega_R^e, e < R = 2^log_r, internal form, 8 bytes each
template <class HF>
static int wf_get_big_table(wf_ctx *ctx, uint32_t log_r, void **out) {
    typedef typename HF::T T;
    static_assert(sizeof(T) == 8, "three-step passes exist for the 64-bit Goldilocks field only");
    auto it = ctx->big_tables.find(log_r);
    if (it == ctx->big_tables.end()) {
        std::vector<T> h((size_t)1 << log_r);
        const T w = HF::root_of_unity(log_r);
        T cur = HF::from_u64(1);
        for (size_t i = 0; i < h.size(); i++) {
            h[i] = HF::to_internal(cur);
            cur = HF::mulmod(cur, w);
        }
        void *p;
        WF_TRY(wf_upload(ctx, h, &p));
        it = ctx->big_tables.emplace(log_r, p).first;
    }
    *out = it->second;
    return WF_OK;
}

// LDE pre-scale tables: for coset u (rows u + b*m of the LDE), series (offset * g^u)^j, j < n, g = omega_{n*b}
template <class HF>
static int wf_get_lde_tables(wf_ctx *ctx, typename HF::T offset_canon, uint32_t log_n, uint32_t log_b, wf_ctx::LdeTables *out,
                             uint64_t *lo_stride, uint64_t *hi_stride) {
    typedef typename HF::T T;
    const uint32_t log_lo = log_n < 12 ? log_n : 12;
    const uint64_t nlo = 1ull << log_lo, nhi = 1ull << (log_n - log_lo);
    *lo_stride = nlo;
    *hi_stride = nhi;
    uint64_t o0, o1;
    split128(offset_canon, o0, o1);
    auto key = std::make_tuple((int)HF::Dev::ID, o0, o1, log_n, log_b);
    auto it = ctx->lde_tables.find(key);
    if (it == ctx->lde_tables.end()) {
        const uint32_t b = 1u << log_b;
        const T g = HF::root_of_unity(log_n + log_b);
        std::vector<T> lo(nlo * b), hi(nhi * b);
        for (uint32_t u = 0; u < b; u++) {
            const T base = HF::mulmod(offset_canon, HF::powmod(g, u));
            T cur = HF::from_u64(1);
            for (uint64_t i = 0; i < nlo; i++) {
                lo[u * nlo + i] = HF::to_internal(cur);
                cur = HF::mulmod(cur, base);
            }
            const T step = cur;
            cur = HF::from_u64(1);
            for (uint64_t i = 0; i < nhi; i++) {
                hi[u * nhi + i] = HF::to_internal(cur);
                cur = HF::mulmod(cur, step);
            }
        }
        wf_ctx::LdeTables t;
        t.log_lo = log_lo;
        WF_TRY(wf_upload(ctx, lo, &t.d_lo));
        WF_TRY(wf_upload(ctx, hi, &t.d_hi));
        it = ctx->lde_tables.emplace(key, t).first;
    }
    *out = it->second;
    return WF_OK;
}

// read one base-field element (internal form) from a host pointer -> canonical value; rejects non-canonical words and 0
template <class HF>
static int wf_load_offset(const void *h_offset, typename HF::T *canon) {
    if (!h_offset) return WF_ERR_INVALID_ARG;
    typename HF::T m;
    memcpy((void *)&m, h_offset, sizeof(m));
    if (!HF::valid_internal(m)) return WF_ERR_INVALID_ARG;
    *canon = HF::from_internal(m);
    if (*canon == 0) return WF_ERR_ZERO_OFFSET;
    return WF_OK;
}
