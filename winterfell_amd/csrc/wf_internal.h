// Internal (C++) side of the context: stream, error latch, cached twiddle/series tables, scratch buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <map>
#include <set>
#include <mutex>
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

// Two-level table for a geometric series c * b^i, i < 2^log_len:  value(i) = lo[i & (2^log_lo - 1)] * hi[i >> log_lo]
// (field product).  lo has 2^log_lo entries (b^i), hi has 2^(log_len - log_lo) entries (c * b^(i << log_lo)).
// Entries are in the field's internal representation; element size depends on the field.
struct SeriesTable {
    void *d_lo = nullptr;
    void *d_hi = nullptr;
    uint32_t log_lo = 0;
    uint32_t log_len = 0;
};

typedef std::tuple<int, uint64_t, uint64_t, uint64_t, uint64_t, uint32_t> SeriesKey;   // field, base, scale (128-bit each), log_len

#define WF_TREE_TICKETS 64
#define WF_STATUS_MERKLE_TICKET 1u    // bit of the device status word: a Merkle ticket word was not in the expected state
struct wf_ctx {
    // Every extern "C" entry point holds this for its whole duration (WF_ENTER): calls on ONE context from several host threads are
    // serialised instead of racing on the caches below (std::map, pool, scratch, ticket ring).  The reference's TraceLde is Sync and is
    // read from Rayon workers (prover/src/constraints/evaluator/default.rs:187); one context per thread stays the FAST way to call.
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int last_hip_error = 0;
    uint32_t last_device_status = 0;

    // geometric-series tables (twiddles omega_n^i, coset offsets, FRI inverse offsets), keyed by SeriesKey
    std::map<SeriesKey, SeriesTable> series;
    // per field: omega_256^e (256 entries, intra-pass twiddles) and omega_16^j (8 entries, register DFT constants)
    std::map<int, void *> w256, w16;
    // omega_256^e * c for a constant c (the 1/n of an inverse transform folded into the last pass's twiddles)
    std::map<std::tuple<int, uint64_t, uint64_t>, void *> w256_scaled;
    // the same for the f64 limb passes: rows of four words c * omega_256^e * 2^(24 k) mod p (plain integers), key (field, c, 0)
    std::map<std::tuple<int, uint64_t, uint64_t>, void *> w256_4form;
    // NTT pass tables in pair form (f128: (w, w 2^64), tables.cuh wf_get_pair_tables), key (field, c): (w256, w16)
    std::map<std::tuple<int, uint64_t, uint64_t>, std::pair<void *, void *>> pair_tables;
    // inter-pass twiddle tables T[k'][rem] = omega_n^((k' * rem) << log_mult) of the passes whose table is small enough to live in L2;
    // key (field, log_n, log_r of the pass, log_s, log_mult)
    std::map<std::tuple<int, uint32_t, uint32_t, uint32_t, uint32_t>, void *> pass_twiddles;
    // LDE pre-scale tables: key (field, offset (128 bit), log_n, log_blowup) -> contiguous [u][lo] / [u][hi]
    struct LdeTables {
        void *d_lo = nullptr, *d_hi = nullptr;
        uint32_t log_lo = 0;
    };
    std::map<std::tuple<int, uint64_t, uint64_t, uint32_t, uint32_t>, LdeTables> lde_tables;

    // optional per-kernel timing (hipEvents on ctx->stream around every launch), see wf_prof_*
    bool prof_enabled = false;
    // wf_prof_enable(ctx, 2): no per-launch brackets, ONE event before the first launch and one after the last (re-recorded at every
    // launch end): the device time of a whole multi-launch call without the ~2-4 us every bracket adds to its launch
    bool prof_span = false, span_open = false;
    hipEvent_t span_a = nullptr, span_b = nullptr;
    uint64_t span_launches = 0;
    struct ProfRec {
        const char *name;
        hipEvent_t a, b;
    };
    std::vector<ProfRec> prof;

    // wf_malloc / wf_free: a stream-ordered caching pool.  Every kernel and copy of a context is enqueued on ctx->stream, so a
    // block freed by the caller can be handed to the next wf_malloc WITHOUT synchronising: whatever still reads it was enqueued
    // before whatever will write it.  (hipFree is a device-wide synchronisation; a proof allocates ~40 buffers.)
    std::multimap<size_t, void *> pool_free;       // size -> block
    std::map<void *, size_t> pool_live;            // block -> its (rounded) size
    size_t pool_free_bytes = 0;

    // persistent launches: how many workgroups of a kernel the device keeps resident (CUs x occupancy), per kernel
    std::map<const void *, uint32_t> resident_blocks;
    // WF_NTT_PREFETCH=1 switches the persistent, next-tile-prefetching NTT launches on.  Off by default: measured slower on
    // MI355X (2^24 f64: 304 us vs 227 us; the 32 extra VGPRs cost a wave per SIMD)
    bool ntt_prefetch = false;
    // Two-pass plans with three-step passes of radix 2^10 .. 2^12 (ntt_big.cuh) for f64 transforms of 2^20 .. 2^24 points.
    // WF_NTT_BIG (read once at context creation): 0 = never, 1 = every eligible transform, unset = the default of ntt_run
    // (batched transforms, where the data stream from HBM and a pass less is worth more than the instructions it costs)
    int ntt_big = -1;
    std::map<uint32_t, void *> big_tables;     // log_r -> omega_R^e, e < R (f64 Montgomery residues)
    // kernels of this context's DEVICE that have been opted in to more than 64 KiB of dynamic LDS (hipFuncSetAttribute is per device,
    // a context is bound to one device and serialised by `mu`: a process-wide flag would skip the opt-in on a second GPU)
    std::set<const void *> big_lds_opt_in;
    // WF_NTT_F64_TABLES=0|1 (unset: where measured faster): the f64 passes take their inter-pass twiddles from one-word tables (one product
    // per element) instead of from the per-lane progression (two products per element); ntt_engine.cuh get_pass_twiddles
    int f64_tw_tables = -1;        // -1: the rule of get_pass_twiddles (batches of <= 2^19-point vectors)
    // WF_NTT_BT=0|1 (unset: 2^23-point transforms): block tiles for single three-pass f64 transforms (ntt_engine.cuh, BT0)
    int ntt_bt = -1;
    // WF_VT_PREFETCH=1 (-DWF_EXPERIMENTS builds only): the persistent prefetching variant of the non-last vector-tile passes; measured slower
    bool vt_prefetch = false;
    // WF_LDE_VT=0 (read once at context creation): no vector tiles for the coset LDE of wide f64 traces (ntt_pass<..., VT>; A/B measurements)
    bool lde_vt = true;
    // WF_ROWS_HASH_WIDE=0 (read once at context creation): rows wider than one 8-column group are hashed by the separate row-hash
    // kernel instead of the last NTT pass (A/B measurements)
    bool rows_hash_wide = true;
    // WF_NTT_COSET_ORDER=0: the first pass of a coset LDE walks its tiles vector by vector (round 4) instead of running the cosets
    // of a source tile side by side on one XCD
    bool coset_order = true;
    // WF_NTT_PLAN, parsed once at context creation (context.hip): a pass plan for transforms of 2^plan_log_n points, 0 = none
    uint32_t plan_log_n = 0, plan_npass = 0, plan_log_r[6] = {0, 0, 0, 0, 0, 0};

    // pinned bounce buffers for host <-> device copies of PAGEABLE caller memory (context.hip: staged_copy)
    void *h_bounce[2] = {nullptr, nullptr};
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    // small host -> device uploads that do NOT wait for the stream (wf_copy_h2d_small_async): a ring of page-locked slots, each
    // reused only after the copy that last read it has completed (its event)
    static constexpr int STAGE_SLOTS = 8;
    static constexpr size_t STAGE_BYTES = 64u << 10;
    void *h_stage[STAGE_SLOTS] = {};
    hipEvent_t stage_ev[STAGE_SLOTS] = {};
    bool stage_busy[STAGE_SLOTS] = {};
    uint32_t stage_next = 0;
    // device status word (one uint32_t, zero = fine): kernels that detect a protocol failure (a Merkle ticket that can never
    // complete) set a bit; wf_ctx_sync and every synchronising entry point report it as WF_ERR_DEVICE_STATUS
    uint32_t *d_status = nullptr;
    uint32_t *h_status = nullptr;     // pinned mirror

    void *d_tree_ticket = nullptr;     // merkle_finish_kernel's ticket words (WF_TREE_TICKETS of them, each zero between its launches)
    uint32_t tree_ticket_next = 0;
    uint32_t tree_ticket_epoch[WF_TREE_TICKETS] = {};   // the epoch each ticket word carries in its high 12 bits (merkle_finish_kernel)

    // scratch buffers (grow-only)
    void *scratch[3] = {nullptr, nullptr, nullptr};
    size_t scratch_bytes[3] = {0, 0, 0};
    std::vector<void *> owned;  // everything hipMalloc'ed for caches
};

int wf_scratch(wf_ctx *ctx, int slot, size_t bytes, void **out);
// All device allocations of the library go through these two (context.hip).  WF_DEBUG_GUARD=1 (read once per process) turns every
// allocation into its own virtual-address reservation with unmapped pages on both sides and the block's END on the last mapped byte,
// so that a kernel that reads or writes past a buffer faults at once instead of landing in a neighbour (tests/README_guard.md);
// WF_DEBUG_GUARD=2 keeps hipMalloc and surrounds every block with red zones that are checked when it is freed.
int wf_dev_malloc(wf_ctx *ctx, void **d_ptr, size_t bytes);
int wf_dev_free(wf_ctx *ctx, void *d_ptr);
// host <-> device copies that never hand PAGEABLE caller memory to the runtime (see context.hip); synchronise the stream
int wf_copy_h2d(wf_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int wf_copy_d2h(wf_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
// the same host -> device copy for at most wf_ctx::STAGE_BYTES, enqueued WITHOUT waiting for the stream: the bytes are copied into a
// page-locked slot of the context first, so h_src may die as soon as the call returns (larger copies fall back to wf_copy_h2d)
int wf_copy_h2d_small_async(wf_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int wf_check_status(wf_ctx *ctx);    // after a stream synchronisation: WF_ERR_DEVICE_STATUS when a kernel flagged a failure

// several small host -> device copies whose destinations lie in ONE device block: collected into a host image, flushed as one
// staged copy (one stream synchronisation instead of one per piece); gaps between the pieces are written as zeros
struct WfUploadBatch {
    wf_ctx *ctx;
    char *d_base;
    std::vector<char> image;
    WfUploadBatch(wf_ctx *c, void *base) : ctx(c), d_base((char *)base) {}
    void add(void *d_dst, const void *h_src, size_t n) {
        if (n == 0) return;
        const size_t off = (size_t)((char *)d_dst - d_base);
        if (image.size() < off + n) image.resize(off + n);
        memcpy(&image[off], h_src, n);
    }
    int flush() { return image.empty() ? (int)WF_OK : wf_copy_h2d(ctx, d_base, image.data(), image.size()); }
    // the same without waiting for the stream (images of at most wf_ctx::STAGE_BYTES; larger ones synchronise as flush() does)
    int flush_async() { return image.empty() ? (int)WF_OK : wf_copy_h2d_small_async(ctx, d_base, image.data(), image.size()); }
};

#define WF_ENTER(ctx)                                   \
    if (!(ctx)) return WF_ERR_INVALID_ARG;              \
    std::lock_guard<std::recursive_mutex> wf_lock_((ctx)->mu)
int wf_resident_blocks(wf_ctx *ctx, const void *kernel, uint32_t *out);   // 256-thread workgroups resident on the whole device

// RAII-less helpers: bracket a kernel launch with events when profiling is on
inline void wf_prof_begin(wf_ctx *ctx, const char *name) {
    if (ctx->prof_span) {
        if (!ctx->span_open) {
            if (!ctx->span_a) {
                (void)hipEventCreate(&ctx->span_a);
                (void)hipEventCreate(&ctx->span_b);
            }
            (void)hipEventRecord(ctx->span_a, ctx->stream);
            ctx->span_open = true;
        }
        ctx->span_launches++;
        return;
    }
    if (!ctx->prof_enabled) return;
    wf_ctx::ProfRec r;
    r.name = name;
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, ctx->stream);
    ctx->prof.push_back(r);
}
inline void wf_prof_end(wf_ctx *ctx) {
    if (ctx->prof_span) {
        (void)hipEventRecord(ctx->span_b, ctx->stream);
        return;
    }
    if (!ctx->prof_enabled) return;
    (void)hipEventRecord(ctx->prof.back().b, ctx->stream);
}
// ---- NTT engine (ntt_engine.cuh, instantiated per field in ntt_f64.hip / ntt_f128.hip) ---------------------
// All strides are in ELEMENTS of the field's word type.
struct NttJob {
    int field = WF_FIELD_F64;
    const void *src = nullptr;       // input vectors
    void *dst = nullptr;             // output vectors (may equal src)
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
    const void *pre_lo = nullptr, *pre_hi = nullptr;
    uint32_t pre_log_lo = 0, pre_mod = 1;
    uint64_t pre_lo_stride = 0, pre_hi_stride = 0;
    const void *post_lo = nullptr, *post_hi = nullptr;
    uint32_t post_log_lo = 0;
    bool has_post_const = false;
    uint64_t post_const[2] = {0, 0}; // internal representation, little-endian words
    // Row-major output (the LDE of RowMatrix::evaluate_polys_over): output vector v = bc * 2^rm_log_b + u is the coset u of
    // base column bc; element m of it goes to dst[(u + (m << rm_log_b)) * rm_row_width + bc].  The last pass then tiles
    // over groups of 2^rm_log_i columns (fastest) so that a row's columns are stored together, and the lanes of the last
    // group also zero the padding columns base_cols .. rm_row_width.  nvec must be rm_base_cols << rm_log_b.
    bool rowmajor = false;
    uint32_t rm_log_b = 0, rm_log_i = 0, rm_base_cols = 0;
    uint64_t rm_row_width = 0;
    // Rows + leaves (narrow traces, one 8-column group, f64, Blake3_256; wf_ntt_rows_mode_ok says when): the last pass writes the
    // padded row-major rows (same addressing as rowmajor, rm_row_width = 8) AND leaf r = Blake3_256::hash_elements(row r) to
    // rh_leaves, straight from the transform — no coset-major buffer, no transpose launch.  rm_log_b / rm_base_cols as above.
    void *rh_leaves = nullptr;
};
int wf_ntt_rows_mode_ok(int field, uint32_t log_n, uint32_t log_blowup, uint32_t base_cols);   // ntt_f64.hip
int wf_evaluate_polys_over_fused(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_polys, uint32_t num_cols, uint64_t col_stride,
                                 uint32_t log_n, uint32_t log_blowup, const void *h_offset, void *d_lde, int hash, void *d_leaves,
                                 int *fused);   // fft_api.hip
int wf_lde_transpose_hash(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_tmp, void *d_lde, uint32_t base_cols,
                          uint32_t log_n, uint32_t log_b, uint32_t log_tm, void *d_leaves, int *done);   // hash_kernels.hip
int wf_fri_fold_commit(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, uint32_t log_nf, const void *d_transposed, uint64_t rc, const void *io_lo,
                       const void *io_hi, uint32_t io_log_lo, const void *w16, uint64_t inv_n, const void *d_alpha, uint64_t g_step, void *d_folded,
                       void *d_transposed_next, void *d_leaves_next, int *done);
int wf_merkle_build_coin(wf_ctx *ctx, int hash, const void *d_leaves, uint64_t num_leaves, void *d_nodes, int field, uint32_t ext_degree, void *d_coin,
                         void *d_root_out, void *d_alpha_out, int *done);   // merkle.hip
int wf_fri_tail(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, uint32_t log_nf, const void *d_evals, uint32_t log_len, uint32_t num_layers,
                void *const *d_transposed, void *const *d_leaves, void *const *d_nodes, void *const *d_folded, void *d_roots, void *d_alphas, void *d_coin,
                const void *io_lo, const void *io_hi, uint32_t io_log_lo, const void *w16, uint64_t inv_n, void *d_remainder, uint32_t rem_size,
                uint64_t rem_w_inv, uint64_t rem_off_inv, uint64_t rem_n_inv, int *done);   // fri_rows.hip
int wf_fri_tail_ok(int hash, int field, uint32_t ext_degree, uint32_t log_nf, uint32_t log_len, uint32_t nt);   // fri_rows.hip
int wf_fri_tail_rem_ok(uint32_t ext_degree, uint32_t log_rem_n, uint32_t rem_size);
int wf_fri_transpose_hash(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_evals, uint32_t log_rc, uint32_t log_nf,
                          void *d_transposed, void *d_leaves, int *done);   // hash_kernels.hip
int wf_ntt_run(wf_ctx *ctx, const NttJob &job);          // dispatches on job.field
int wf_ntt_run_f64(wf_ctx *ctx, const NttJob &job);
int wf_ntt_run_f128(wf_ctx *ctx, const NttJob &job);
int wf_ntt_run_f62(wf_ctx *ctx, const NttJob &job);
