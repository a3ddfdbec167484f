// Multi-device entry points behind the C ABI: one rank per GPU, the reference's own multi-device commitment (PartitionOptions,
// air/src/options.rs:391-451; row_matrix.rs:204-223) — every rank owns one partition's columns — with the two exchange steps
// of SURVEY 8(e) done on the device interconnect:
//   * all-to-all of the 32-byte partition digests (each rank ends up with every partition's digest of ITS row range),
//   * all-gather of the G sub-roots (the top log2 G levels are recomputed on every rank).
// Transports:
//   RCCL      one process per GPU (or one thread per GPU): ncclCommInitRank on the context's device, collectives on the
//             context's stream.  librccl is dlopen()ed on first use — a process that already holds an RCCL (PyTorch ships its
//             own copy) is not handed a second one at load time, and a single-GPU user never loads it at all.
//   loopback  all ranks are threads of THIS process (any mix of devices, including several logical ranks on one GPU): the
//             collectives are peer copies between the ranks' buffers around a thread barrier.  This is how the sharded path is
//             exercised rank for rank on a one-GPU box (tests/test_gpu_comm.py), and a fallback for single-process
//             multi-GPU hosts without RCCL.
#include <dlfcn.h>
#include <pthread.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "wf_internal.h"

namespace {

// ---- the slice of the RCCL API used here (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclCommDestroy :260,
// ncclAllGather :678, ncclAllToAll :790; ncclUint8 = 1) ------------------------------------------------------------------------
struct RcclId {
    char internal[WF_COMM_ID_BYTES];
};
typedef void *RcclComm;
struct Rccl {
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(RcclComm *, int, RcclId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*AllToAll)(const void *, void *, size_t, int, RcclComm, hipStream_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static std::atomic<bool> tried{false};
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    if (tried.load()) return r;
    pthread_mutex_lock(&mu);
    if (!tried.load()) {
        void *h = nullptr;
        // an RCCL that is already mapped (PyTorch's) wins; otherwise the ROCm one
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (h) break;
        }
        if (!h)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
        if (h) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
            r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
            r.AllToAll = (decltype(r.AllToAll))dlsym(h, "ncclAllToAll");
            r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllToAll;
        }
        tried.store(true);
    }
    pthread_mutex_unlock(&mu);
    return r;
}

// ---- loopback transport: shared by the `world` ranks of one wf_comm_init_loopback call --------------------------------------
// The barrier can be ABORTED: a rank that fails between two collectives (an allocation, a kernel launch, a stream error) would
// otherwise leave its peers waiting for it forever.  abort() wakes every waiter; from then on every wait() returns false and
// the collectives return WF_ERR_COMM_ABORTED on every rank.
struct AbortableBarrier {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, count = 0;
    uint64_t generation = 0;
    bool aborted = false;
    bool wait() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t gen = generation;
        if (++count == world) {
            count = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || aborted; });
        }
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
};
struct Loopback {
    int world = 0;
    AbortableBarrier bar;
    std::vector<const void *> send;      // what every rank published for the collective in flight
    std::atomic<int> refs{0};
};

// frees what a multi-step entry point allocated from the context's pool, on every way out; on an error exit of a loopback rank
// it also aborts the communicator so that the peers' collectives return instead of waiting
struct CommScope {
    wf_ctx *ctx;
    Loopback *loop;
    std::vector<void *> blocks;
    bool ok = false;
    CommScope(wf_ctx *c, Loopback *l) : ctx(c), loop(l) {}
    int alloc(size_t bytes, void **out) {
        const int st = wf_malloc(ctx, bytes, out);
        if (st == WF_OK) blocks.push_back(*out);
        return st;
    }
    void release(void *p) {
        for (size_t i = 0; i < blocks.size(); i++)
            if (blocks[i] == p) {
                blocks.erase(blocks.begin() + (long)i);
                (void)wf_free(ctx, p);
                return;
            }
    }
    ~CommScope() {
        if (!ok && loop) {
            // error exit of a loopback rank: release the peers FIRST, and keep this rank's temporaries allocated — a peer that is past
            // the first barrier of a collective may still be copying from them on its own stream.  They stay live blocks of the
            // context's pool and go back to the driver with wf_ctx_destroy.  After WF_ERR_COMM_ABORTED the communicator is dead:
            // every rank destroys its wf_comm and a new one is created (include/winterfell_hip.h).
            loop->bar.abort();
            return;
        }
        for (void *p : blocks) (void)wf_free(ctx, p);
    }
};

}  // namespace

struct wf_comm {
    wf_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    RcclComm nccl = nullptr;             // RCCL transport
    Loopback *loop = nullptr;            // loopback transport
};

extern "C" int wf_comm_get_unique_id(uint8_t *id) {
    if (!id) return WF_ERR_INVALID_ARG;
    Rccl &r = rccl();
    if (!r.ok) return WF_ERR_UNSUPPORTED;
    RcclId u;
    if (r.GetUniqueId(&u) != 0) return WF_ERR_HIP;
    memcpy(id, u.internal, WF_COMM_ID_BYTES);
    return WF_OK;
}

extern "C" int wf_comm_init_rank(wf_ctx *ctx, const uint8_t *id, int rank, int world, wf_comm **out) {
    WF_ENTER(ctx);
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return WF_ERR_INVALID_ARG;
    Rccl &r = rccl();
    if (!r.ok) return WF_ERR_UNSUPPORTED;
    WF_HIP(hipSetDevice(ctx->device));
    RcclId u;
    memcpy(u.internal, id, WF_COMM_ID_BYTES);
    RcclComm c = nullptr;
    if (r.CommInitRank(&c, world, u, rank) != 0) return WF_ERR_HIP;
    wf_comm *cm = new wf_comm();
    cm->ctx = ctx;
    cm->rank = rank;
    cm->world = world;
    cm->nccl = c;
    *out = cm;
    return WF_OK;
}

extern "C" int wf_comm_init_loopback(wf_ctx *const *ctxs, int world, wf_comm **out) {
    if (!ctxs || !out || world < 1) return WF_ERR_INVALID_ARG;
    for (int i = 0; i < world; i++)
        if (!ctxs[i]) return WF_ERR_INVALID_ARG;
    Loopback *lb = new Loopback();
    lb->world = world;
    lb->send.assign(world, nullptr);
    lb->bar.world = world;
    lb->refs.store(world);
    for (int i = 0; i < world; i++) {
        wf_comm *cm = new wf_comm();
        cm->ctx = ctxs[i];
        cm->rank = i;
        cm->world = world;
        cm->loop = lb;
        out[i] = cm;
    }
    return WF_OK;
}

extern "C" int wf_comm_destroy(wf_comm *cm) {
    if (!cm) return WF_ERR_INVALID_ARG;
    if (cm->nccl) (void)rccl().CommDestroy(cm->nccl);
    if (cm->loop && cm->loop->refs.fetch_sub(1) == 1) delete cm->loop;
    delete cm;
    return WF_OK;
}

extern "C" int wf_comm_rank(const wf_comm *cm) { return cm ? cm->rank : -1; }
extern "C" int wf_comm_size(const wf_comm *cm) { return cm ? cm->world : -1; }

// every rank contributes `bytes` bytes; d_recv gets world * bytes, rank order
extern "C" int wf_comm_all_gather(wf_comm *cm, const void *d_send, void *d_recv, uint64_t bytes) {
    if (!cm || !d_send || !d_recv || bytes == 0) return WF_ERR_INVALID_ARG;
    wf_ctx *ctx = cm->ctx;
    std::lock_guard<std::recursive_mutex> wf_lock_(ctx->mu);    // a rank thread holds only its own context's lock across the barriers
    if (cm->nccl) {
        if (rccl().AllGather(d_send, d_recv, bytes, /*ncclUint8*/ 1, cm->nccl, ctx->stream) != 0) return WF_ERR_HIP;
        return WF_OK;
    }
    if (!cm->loop) {                                    // a communicator of one rank
        WF_HIP(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return WF_OK;
    }
    Loopback *lb = cm->loop;
    // a rank that fails here aborts the barrier instead of leaving: its peers are (or will be) waiting on it
    auto fail = [&](hipError_t e) {
        ctx->last_hip_error = (int)e;
        lb->bar.abort();
        return WF_ERR_HIP;
    };
    hipError_t e = hipStreamSynchronize(ctx->stream);   // what this rank publishes is complete
    if (e != hipSuccess) return fail(e);
    lb->send[cm->rank] = d_send;
    if (!lb->bar.wait()) return WF_ERR_COMM_ABORTED;
    for (int k = 0; k < lb->world; k++) {
        e = hipMemcpyAsync((uint8_t *)d_recv + (size_t)k * bytes, lb->send[k], bytes, hipMemcpyDefault, ctx->stream);
        if (e != hipSuccess) return fail(e);
    }
    e = hipStreamSynchronize(ctx->stream);              // nobody reuses a send buffer before every reader is done
    if (e != hipSuccess) return fail(e);
    if (!lb->bar.wait()) return WF_ERR_COMM_ABORTED;
    return WF_OK;
}

// d_send holds `world` blocks of `bytes` bytes, block k goes to rank k; d_recv block k comes from rank k (distinct buffers)
extern "C" int wf_comm_all_to_all(wf_comm *cm, const void *d_send, void *d_recv, uint64_t bytes) {
    if (!cm || !d_send || !d_recv || bytes == 0 || d_send == d_recv) return WF_ERR_INVALID_ARG;
    wf_ctx *ctx = cm->ctx;
    std::lock_guard<std::recursive_mutex> wf_lock_(ctx->mu);    // a rank thread holds only its own context's lock across the barriers
    if (cm->nccl) {
        if (rccl().AllToAll(d_send, d_recv, bytes, 1, cm->nccl, ctx->stream) != 0) return WF_ERR_HIP;
        return WF_OK;
    }
    if (!cm->loop) {
        WF_HIP(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return WF_OK;
    }
    Loopback *lb = cm->loop;
    auto fail = [&](hipError_t e) {
        ctx->last_hip_error = (int)e;
        lb->bar.abort();
        return WF_ERR_HIP;
    };
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(e);
    lb->send[cm->rank] = d_send;
    if (!lb->bar.wait()) return WF_ERR_COMM_ABORTED;
    for (int k = 0; k < lb->world; k++) {
        e = hipMemcpyAsync((uint8_t *)d_recv + (size_t)k * bytes, (const uint8_t *)lb->send[k] + (size_t)cm->rank * bytes, bytes, hipMemcpyDefault,
                           ctx->stream);
        if (e != hipSuccess) return fail(e);
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(e);
    if (!lb->bar.wait()) return WF_ERR_COMM_ABORTED;
    return WF_OK;
}

namespace {
// [k][row] -> [row][k] on 32-byte digests (what wf_hash_merge_many_batch reads)
__global__ __launch_bounds__(256) void regroup_digests_kernel(const uint4 *in, uint4 *out, uint64_t rows, uint32_t k) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte half of a digest
    if (i >= rows * k * 2) return;
    const uint64_t half = i & 1, d = i >> 1, row = d / k, part = d % k;
    out[i] = in[(part * rows + row) * 2 + half];
}
}  // namespace

// Column-sharded trace commitment, this rank's part (call it on every rank).  The rank holds partition `rank` of the columns:
// d_trace_shard = shard_cols columns of 2^log_n evaluations.  Output, all on this rank's device:
//   d_trace_shard  the columns' interpolation polynomials (in place, unless skip_interpolate)
//   d_lde_shard    the shard's LDE, row-major [N = 2^(log_n + log_blowup)][row width of shard_cols]
//   d_leaves       the leaves of this rank's row range [rank N/G, (rank + 1) N/G): merge_many over the G partition digests
//   d_nodes        the subtree over those leaves, heap order (N/G digests)
//   d_top          the top tree over the G sub-roots, heap order (G digests; d_top[1] is the root), identical on every rank
// The result is node for node the single-device commitment under PartitionOptions(G, hash_rate) with shard_cols columns per
// partition (G = 1: the plain row hashes — partition_size == num_cols, row_matrix.rs:193).
extern "C" int wf_comm_sharded_commit(wf_comm *cm, int hash, int field, uint32_t ext_degree, void *d_trace_shard, uint32_t shard_cols,
                                      uint64_t col_stride, uint32_t log_n, uint32_t log_blowup, const void *h_offset, int skip_interpolate,
                                      void *d_lde_shard, void *d_leaves, void *d_nodes, void *d_top, void *h_root) {
    if (!cm || !d_trace_shard || !d_lde_shard || !d_leaves || !d_nodes || !d_top || shard_cols == 0 || ext_degree == 0) return WF_ERR_INVALID_ARG;
    wf_ctx *ctx = cm->ctx;
    std::lock_guard<std::recursive_mutex> wf_lock_(ctx->mu);    // a rank thread holds only its own context's lock across the barriers
    const uint32_t G = (uint32_t)cm->world;
    const uint64_t N = 1ull << (log_n + log_blowup);
    if (G & (G - 1)) return WF_ERR_NOT_POWER_OF_TWO;     // the sub-trees must tile a binary tree
    if (N % G) return WF_ERR_INVALID_ARG;
    const uint64_t per = N / G;
    CommScope scope(ctx, cm->loop);                      // temporaries freed, peers released, on every error exit below
    WF_HIP(hipSetDevice(ctx->device));
    if (!skip_interpolate) WF_TRY(wf_interpolate_columns(ctx, field, ext_degree, d_trace_shard, shard_cols, col_stride, log_n));
    WF_TRY(wf_evaluate_polys_over(ctx, field, ext_degree, d_trace_shard, shard_cols, col_stride, log_n, log_blowup, h_offset, d_lde_shard));
    const uint64_t rw = wf_row_width(shard_cols, ext_degree);
    if (G == 1) {
        WF_TRY(wf_hash_rows(ctx, hash, field, ext_degree, d_lde_shard, N, rw, shard_cols * ext_degree, 1, 1, d_leaves));
    } else {
        void *digests, *recv, *grouped;
        WF_TRY(scope.alloc(N * 32, &digests));
        WF_TRY(scope.alloc(N * 32, &recv));
        WF_TRY(scope.alloc(N * 32, &grouped));
        // this partition's digest of every row, blocks of `per` rows = the row ranges of the ranks
        WF_TRY(wf_hash_rows(ctx, hash, field, ext_degree, d_lde_shard, N, rw, shard_cols * ext_degree, 1, 1, digests));
        WF_TRY(wf_comm_all_to_all(cm, digests, recv, per * 32));                       // recv[k][row]: partition k, my rows
        hipLaunchKernelGGL(regroup_digests_kernel, dim3((uint32_t)((per * G * 2 + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4 *)recv,
                           (uint4 *)grouped, per, G);
        WF_HIP(hipGetLastError());
        WF_TRY(wf_hash_merge_many_batch(ctx, hash, grouped, per, G, d_leaves));        // leaf = merge_many(partition digests)
        scope.release(digests);
        scope.release(recv);
        scope.release(grouped);
    }
    const void *sub_root = d_leaves;
    if (per > 1) {
        WF_TRY(wf_merkle_build(ctx, hash, d_leaves, per, d_nodes));
        sub_root = (const uint8_t *)d_nodes + 32;
    } else {
        WF_HIP(hipMemcpyAsync(d_nodes, d_leaves, 32, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (G == 1) {
        WF_HIP(hipMemcpyAsync((uint8_t *)d_top, sub_root, 32, hipMemcpyDeviceToDevice, ctx->stream));
        if (h_root) WF_TRY(wf_memcpy_d2h(ctx, h_root, sub_root, 32));
        scope.ok = true;
        return WF_OK;
    }
    void *roots;
    WF_TRY(scope.alloc((size_t)G * 32, &roots));
    WF_TRY(wf_comm_all_gather(cm, sub_root, roots, 32));
    WF_TRY(wf_merkle_build(ctx, hash, roots, G, d_top));
    scope.release(roots);
    if (h_root) WF_TRY(wf_memcpy_d2h(ctx, h_root, (const uint8_t *)d_top + 32, 32));
    scope.ok = true;
    return WF_OK;
}

namespace {
// piece chunks [jj][g] -> send blocks [g][jj] (16-byte units): what rank g needs from this rank, in the order it will read it
__global__ __launch_bounds__(256) void fri_pack_chunks_kernel(const uint4 *piece, uint4 *send, uint64_t chunk_u4, uint32_t G, uint32_t per_dest) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = chunk_u4 * G * per_dest;
    if (i >= total) return;
    const uint64_t c = i / chunk_u4, w = i - c * chunk_u4;          // c = destination-major chunk index g * per_dest + jj
    const uint32_t g = (uint32_t)(c / per_dest), jj = (uint32_t)(c % per_dest);
    send[i] = piece[((uint64_t)jj * G + g) * chunk_u4 + w];
}
// from the whole layer: chunk (j, rank) of every strided run j -> [j][rows_local]
__global__ __launch_bounds__(256) void fri_cut_chunks_kernel(const uint4 *full, uint4 *out, uint64_t chunk_u4, uint32_t G, uint32_t rank, uint32_t N) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chunk_u4 * N) return;
    const uint64_t j = i / chunk_u4, w = i - j * chunk_u4;
    out[i] = full[(j * G + rank) * chunk_u4 + w];
}
}  // namespace

extern "C" int wf_comm_sharded_fri_layers(wf_comm *cm, int hash, int field, uint32_t ext_degree, const void *d_piece, uint32_t log_len,
                                          uint32_t folding, uint32_t num_layers, const void *h_domain_offset, void *d_coin,
                                          void *const *d_rows, void *const *d_leaves, void *const *d_nodes, void *const *d_top,
                                          void *const *d_folded, void *d_roots, void *d_alphas) {
    if (!cm || !d_piece || !h_domain_offset || !d_coin || !d_roots || !d_alphas || ext_degree == 0) return WF_ERR_INVALID_ARG;
    if (num_layers == 0) return WF_OK;
    if (!d_rows || !d_leaves || !d_nodes || !d_top || !d_folded) return WF_ERR_INVALID_ARG;
    if (folding != 2 && folding != 4 && folding != 8 && folding != 16) return WF_ERR_UNSUPPORTED;
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128 && field != WF_FIELD_F62) return WF_ERR_UNSUPPORTED;
    wf_ctx *ctx = cm->ctx;
    std::lock_guard<std::recursive_mutex> wf_lock_(ctx->mu);    // a rank thread holds only its own context's lock across the barriers
    const uint32_t G = (uint32_t)cm->world, r = (uint32_t)cm->rank, N = folding;
    if (G & (G - 1)) return WF_ERR_NOT_POWER_OF_TWO;
    uint32_t log_nf = 0, log_g = 0;
    while ((1u << log_nf) < N) log_nf++;
    while ((1u << log_g) < G) log_g++;
    const size_t eb = (size_t)ext_degree * (field == WF_FIELD_F128 ? 16 : 8);       // bytes per element of E
    // every layer's shape, alignment and pointers are checked BEFORE the first collective: a bad layer k must not be found after
    // layers 0 .. k-1 ran collectives and reseeded the coin (every rank sees the same arguments, so every rank returns here)
    for (uint32_t k = 0, ll = log_len; k < num_layers; k++, ll -= log_nf) {
        if (!d_rows[k] || !d_leaves[k] || !d_nodes[k] || !d_top[k] || !d_folded[k]) return WF_ERR_INVALID_ARG;
        if (ll < log_nf + log_g + 1) return WF_ERR_INVALID_ARG;                        // at least two rows per rank (a subtree)
        if ((((size_t)1 << (ll - log_nf - log_g)) * eb) % 16) return WF_ERR_INVALID_ARG;
    }
    CommScope scope(ctx, cm->loop);                      // temporaries freed, peers released, on every error exit below
    WF_HIP(hipSetDevice(ctx->device));
    const void *piece = d_piece;
    for (uint32_t k = 0; k < num_layers; k++) {
        const uint64_t len = 1ull << log_len, rows_local = len >> (log_nf + log_g);    // = elements per chunk
        const size_t chunk_bytes = (size_t)rows_local * eb;
        const uint64_t chunk_u4 = chunk_bytes / 16;
        // 1. the rank's chunk-major buffer [N][rows_local]: element (j, i) = e[r rows_local + i + j rc]
        void *cmaj = nullptr, *tmp = nullptr;
        if (G == 1) {
            cmaj = const_cast<void *>(piece);
        } else if (N % G == 0) {
            const uint32_t per_dest = N / G;                                           // chunks this rank sends to every rank
            WF_TRY(scope.alloc(chunk_bytes * N, &tmp));
            WF_TRY(scope.alloc(chunk_bytes * N, &cmaj));
            const uint64_t total = chunk_u4 * N;
            hipLaunchKernelGGL(fri_pack_chunks_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4 *)piece, (uint4 *)tmp,
                               chunk_u4, G, per_dest);
            WF_HIP(hipGetLastError());
            WF_TRY(wf_comm_all_to_all(cm, tmp, cmaj, chunk_bytes * per_dest));         // source rank h sends chunks j = h per_dest .. in order
        } else {
            WF_TRY(scope.alloc(chunk_bytes * N * G, &tmp));                            // the whole layer
            WF_TRY(scope.alloc(chunk_bytes * N, &cmaj));
            WF_TRY(wf_comm_all_gather(cm, piece, tmp, chunk_bytes * N));
            const uint64_t total = chunk_u4 * N;
            hipLaunchKernelGGL(fri_cut_chunks_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4 *)tmp, (uint4 *)cmaj,
                               chunk_u4, G, r, N);
            WF_HIP(hipGetLastError());
        }
        // 2. rows, leaves, subtree: the chunk-major buffer is a "layer" of N * rows_local points for the commit
        WF_TRY(wf_fri_layer_commit(ctx, hash, field, ext_degree, cmaj, log_len - log_g, N, d_rows[k], d_leaves[k], d_nodes[k], nullptr));
        if (tmp) scope.release(tmp);
        if (G > 1) scope.release(cmaj);
        // 3. sub-roots -> top tree (identical on every rank)
        const uint8_t *root;
        if (G == 1) {
            WF_HIP(hipMemcpyAsync(d_top[k], (const uint8_t *)d_nodes[k] + 32, 32, hipMemcpyDeviceToDevice, ctx->stream));
            root = (const uint8_t *)d_top[k];
        } else {
            void *subs;
            WF_TRY(scope.alloc((size_t)G * 32, &subs));
            WF_TRY(wf_comm_all_gather(cm, (const uint8_t *)d_nodes[k] + 32, subs, 32));
            WF_TRY(wf_merkle_build(ctx, hash, subs, G, d_top[k]));
            scope.release(subs);
            root = (const uint8_t *)d_top[k] + 32;
        }
        // 4. channel.commit_fri_layer(root); alpha = channel.draw_fri_alpha() — on this rank's copy of the coin
        void *alpha = (uint8_t *)d_alphas + (size_t)k * eb;
        WF_TRY(wf_coin_reseed_draw(ctx, hash, field, ext_degree, d_coin, root, (uint8_t *)d_roots + (size_t)k * 32, alpha));
        // 5. fold the local rows; x_i uses the global row index
        WF_TRY(wf_fri_apply_drp_rows_dev(ctx, field, ext_degree, d_rows[k], log_len, N, (uint64_t)r * rows_local, rows_local, h_domain_offset, alpha,
                                         d_folded[k]));
        piece = d_folded[k];
        log_len -= log_nf;
    }
    scope.ok = true;
    return WF_OK;
}
