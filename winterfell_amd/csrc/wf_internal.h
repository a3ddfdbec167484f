// Internal (C++) side of the context: stream, error latch, cached twiddle/series tables, scratch buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/winterfell_hip.h"

#define WF_HIP(call)                                  \
    do {                                              \
        hipError_t _e = (call);                       \
        if (_e != hipSuccess) {                       \
            ctx->last_hip_error = (int)_e;            \
            return WF_ERR_HIP;                        \
        }                                             \
    } while (0)

#define WF_TRY(call)                 \
    do {                             \
        int _s = (call);             \
        if (_s != WF_OK) return _s;  \
    } while (0)

// Host-side f64 helpers (plain integer arithmetic on canonical values; tables are converted to the
// internal Montgomery form once, on the host, when they are built).
namespace hostgl {
typedef unsigned __int128 u128;
constexpr uint64_t P = 0xffffffff00000001ull;
inline uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % P); }
inline uint64_t powmod(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mulmod(r, a);
        a = mulmod(a, a);
        e >>= 1;
    }
    return r;
}
inline uint64_t invmod(uint64_t a) { return powmod(a, P - 2); }
inline uint64_t to_mont(uint64_t a) { return (uint64_t)((((u128)a) << 64) % P); }   // a * 2^64 mod p
inline uint64_t from_mont(uint64_t a) { return mulmod(a, invmod(to_mont(1))); }      // a / 2^64 mod p
constexpr uint64_t ROOT_2_32 = 7277203076849721926ull;  // TWO_ADIC_ROOT_OF_UNITY, f64/mod.rs:267
inline uint64_t root_of_unity(uint32_t log_n) { return powmod(ROOT_2_32, 1ull << (32 - log_n)); }
}  // namespace hostgl

// Two-level table for a geometric series c * b^i, i < 2^log_len:  value(i) = lo[i & (2^log_lo - 1)] * hi[i >> log_lo]
// (Montgomery product).  lo has 2^log_lo entries, hi has 2^(log_len - log_lo) entries; `c` is folded into hi.
struct SeriesTable {
    uint64_t *d_lo = nullptr;
    uint64_t *d_hi = nullptr;
    uint32_t log_lo = 0;
    uint32_t log_len = 0;
};

struct wf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int last_hip_error = 0;

    // omega_n^i tables keyed by log_n
    std::map<uint32_t, SeriesTable> omega;
    // omega_256^e (256 entries) for the intra-pass twiddles
    uint64_t *d_w256 = nullptr;
    // generic series cache keyed by (base (canonical), scale (canonical), log_len)
    std::map<std::tuple<uint64_t, uint64_t, uint32_t>, SeriesTable> series;
    // LDE pre-scale tables: key (offset canonical, log_n, log_blowup) -> contiguous [u][lo] / [u][hi]
    struct LdeTables {
        uint64_t *d_lo = nullptr, *d_hi = nullptr;
        uint32_t log_lo = 0;
    };
    std::map<std::tuple<uint64_t, uint32_t, uint32_t>, LdeTables> lde_tables;

    // optional per-kernel timing (hipEvents on ctx->stream around every launch), see wf_prof_*
    bool prof_enabled = false;
    struct ProfRec {
        const char *name;
        hipEvent_t a, b;
    };
    std::vector<ProfRec> prof;

    // scratch buffers (grow-only)
    void *scratch[3] = {nullptr, nullptr, nullptr};
    size_t scratch_bytes[3] = {0, 0, 0};
    std::vector<void *> owned;  // everything hipMalloc'ed for caches
};

int wf_scratch(wf_ctx *ctx, int slot, size_t bytes, void **out);

// RAII-less helpers: bracket a kernel launch with events when profiling is on
inline void wf_prof_begin(wf_ctx *ctx, const char *name) {
    if (!ctx->prof_enabled) return;
    wf_ctx::ProfRec r;
    r.name = name;
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, ctx->stream);
    ctx->prof.push_back(r);
}
inline void wf_prof_end(wf_ctx *ctx) {
    if (!ctx->prof_enabled) return;
    (void)hipEventRecord(ctx->prof.back().b, ctx->stream);
}
int wf_get_omega_table(wf_ctx *ctx, uint32_t log_n, SeriesTable *out);
int wf_get_series_table(wf_ctx *ctx, uint64_t base_canon, uint64_t scale_canon, uint32_t log_len, SeriesTable *out);
int wf_get_w256(wf_ctx *ctx, uint64_t **out);

// ---- f64 NTT engine (ntt_f64.hip) ---------------------------------------------------------------------
struct NttJob {
    const uint64_t *src = nullptr;   // input vectors
    uint64_t *dst = nullptr;         // output vectors (may equal src)
    uint32_t log_n = 0;
    uint32_t nvec = 1;               // number of output vectors
    uint32_t src_div = 1;            // input vector index = v / src_div  (LDE: blowup cosets share one input)
    uint64_t src_vec_stride = 0;     // elements between input vectors
    uint64_t dst_vec_stride = 0;     // elements between output vectors
    uint32_t src_es = 1, dst_es = 1; // element strides (extension fields: ext_degree)
    // vector v lives at  (v / inner) * vec_stride + (v % inner) * inner_stride   (v = v_out / src_div for src)
    uint32_t src_inner = 1, dst_inner = 1;
    uint64_t src_inner_stride = 1, dst_inner_stride = 1;
    bool inverse = false;            // inverse transform (output index negated), WITHOUT scaling
    // optional input scaling by series pre(v % pre_mod, j), and output scaling by post(k) / a constant
    const uint64_t *pre_lo = nullptr, *pre_hi = nullptr;
    uint32_t pre_log_lo = 0, pre_mod = 1;
    uint64_t pre_lo_stride = 0, pre_hi_stride = 0;
    const uint64_t *post_lo = nullptr, *post_hi = nullptr;
    uint32_t post_log_lo = 0;
    uint64_t post_const = 0;         // Montgomery form; 0 = none
};
int wf_ntt_f64_run(wf_ctx *ctx, const NttJob &job);
