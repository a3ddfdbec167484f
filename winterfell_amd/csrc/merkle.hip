// MerkleTree::new = build_merkle_nodes (crypto/src/merkle/mod.rs:344-368, concurrent.rs:26-75): every node of the tree in the
// reference's heap layout, several levels per launch.
#include "hashers.cuh"
#include "coin_state.cuh"
#include "merkle_stage.cuh"

namespace {

// One stage of the tree: `count` input digests (a power of two), each workgroup reduces a chunk of
// CH = min(count, 1024) of them through log2(CH) levels.  Level d of the stage has count >> (d+1) nodes that
// live at heap indices [count >> (d+1), count >> d) of `nodes`.
// THREADS = 1024 for launches of at most 256 workgroups (the chip is not full anyway): level 0 is then one compression deep
// instead of two, and the four-lane levels start at 256 merges instead of 128 — these launches are chains of dependent
// compressions, their cost is their depth.
template <class H, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void merkle_stage_kernel(const void *in, void *nodes, uint64_t count, uint32_t log_ch) {
    __shared__ uint4 bufA[512 * 2];
    __shared__ uint4 bufB[256 * 2];
    merkle_stage_wg<H, THREADS>(in, nodes, count, log_ch, blockIdx.x, (int)threadIdx.x, bufA, bufB);
}

// The launch that finishes a tree, followed in the same kernel by what the FRI layer loop does with the root: channel.commit_fri_layer
// (coin.reseed) and channel.draw_fri_alpha (fri/src/prover/mod.rs:212-216).  One launch and its event bracket less per layer than the
// separate coin kernel (round 3: six of them per 2^24 commit phase, ~7 us each in event time for two dependent compressions).
template <class H, int THREADS, int FIELD, int D>
__global__ __launch_bounds__(THREADS) void merkle_stage_coin_kernel(const void *in, void *nodes, uint64_t count, uint32_t log_ch, CoinState *coin,
                                                                    uint32_t *root_out, uint64_t *alpha_out) {
    __shared__ uint4 bufA[512 * 2];
    __shared__ uint4 bufB[256 * 2];
    merkle_stage_wg<H, THREADS>(in, nodes, count, log_ch, 0, (int)threadIdx.x, bufA, bufB);
    __syncthreads();                         // the root (nodes[1]) is in global memory, written by this workgroup
    const uint32_t *root = reinterpret_cast<const uint32_t *>(nodes) + 8;
    if constexpr (H::QUAD_MERGE) {                                          // Blake3_256
        uint32_t *scratch = reinterpret_cast<uint32_t *>(bufA);          // the tree is done with its LDS
        coin_reseed_draw_quad_wg<FIELD, D>(coin, root, root_out, alpha_out, threadIdx.x, scratch, scratch + 16, reinterpret_cast<int *>(scratch + 24));
    } else {
        if (threadIdx.x == 0) coin_reseed_draw_lane<H, FIELD, D>(coin, root, root_out, alpha_out);
    }
}

// A whole tree of 2^11 .. 2^18 inputs in ONE launch: workgroup w reduces its 1024 inputs through ten levels (merkle_stage_wg), publishes
// them (device-scope fence: the workgroups sit on eight XCDs with an L2 each) and takes a ticket; the workgroup that draws the last
// ticket — every other subtree is then in memory — walks the remaining levels from the gridDim.x subtree tops and, for an FRI layer,
// runs the coin step.  Before, the last levels were a second launch of one workgroup: one launch, its gap and its cold start less per
// tree (round 3: every tree of an FRI commit phase and of a trace / constraint commitment ends this way).  `ticket`: one zero word
// of device memory per context, left zero again by the last workgroup.
// LOG_IN = 12: 4096 inputs per workgroup (trees of 2^19 and 2^20 inputs in one launch as well): the two levels below the stage go
// global to global, two and one merges per lane, and the workgroup reads its own 1024 outputs back (same CU, behind a barrier).
template <class H, int FIELD, int D, bool COIN, int LOG_IN = 10>
__global__ __launch_bounds__(1024) void merkle_finish_kernel(const void *in, void *nodes, uint64_t count, uint32_t *ticket, uint32_t epoch,
                                                             uint32_t *status, CoinState *coin, uint32_t *root_out, uint64_t *alpha_out) {
    __shared__ uint4 bufA[512 * 2];
    __shared__ uint4 bufB[256 * 2];
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x;
    const uint32_t wgs = gridDim.x;
    const void *src = in;
    uint64_t cnt = count, wg = blockIdx.x;
    uint32_t lc = 10;
    if constexpr (LOG_IN > 10) {
#pragma unroll 1
        for (int pre = 0; pre < LOG_IN - 10; pre++) {
            const uint64_t half = cnt >> 1;                                   // this level's nodes: heap indices [half, cnt)
            const uint32_t per_wg = 1u << (LOG_IN - 1 - pre);
            for (uint32_t i = tid; i < per_wg; i += 1024) {
                uint32_t m[16], d[8];
                load_pair(src, wg * per_wg + i, m);
                H::merge(m, d);
                store_digest(nodes, half + wg * per_wg + i, d);
            }
            __syncthreads();
            src = reinterpret_cast<const uint8_t *>(nodes) + half * 32;
            cnt = half;
        }
    }
    for (;;) {                                 // two trips at most; one inlined copy of the stage
        const bool top = (cnt >> lc) == 1;
        merkle_stage_wg<H, 1024>(src, nodes, cnt, lc | (top ? 0x80000000u : 0u), wg, tid, bufA, bufB);
        if (top) break;
        // one fence per workgroup, by the lane that takes the ticket, after the barrier that orders the other lanes' stores before it:
        // a device-scope release writes the XCD's L2 back, and 16 wavefronts x 256 workgroups doing it cost 30 us a tree
        __syncthreads();
        if (tid == 0) {
            // release: the XCD's L2 is written back; the explicit wait keeps the ticket from overtaking the write-back (the compiler drops
            // the wait after buffer_wbl2 when it can prove the wave's memory counter empty, MI355X_MICROARCH.md "Compiler hazard")
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the ticket word carries the epoch of its use in the high 12 bits: a word that is not in the state this launch expects — never
            // zeroed, shared with another tree in flight, written by something else — is REPORTED (the context's status word, checked by the
            // next synchronising call) instead of silently leaving the top of the tree unwritten
            const uint32_t old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t seen = old & 0xfffffu;
            uint32_t last = 0;
            if ((old >> 20) != epoch || seen >= wgs) {
                __hip_atomic_fetch_or(status, WF_STATUS_MERKLE_TICKET, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else if (seen == wgs - 1) {
                last = 1;
                __hip_atomic_store(ticket, ((epoch + 1) & 0xfffu) << 20, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // this CU's L1 forgets what it held of the other subtrees
            }
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        // the last workgroup reads subtree tops written on other XCDs: EVERY wave acquires at device scope before its first such read
        // (one wave's buffer_inv happens to invalidate the CU's L1 today; the memory model does not promise it) — 16 waves, once per tree
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        cnt = wgs;
        src = reinterpret_cast<const uint8_t *>(nodes) + cnt * 32;
        lc = 31 - __builtin_clz(wgs);
        wg = 0;
    }
    if constexpr (COIN) {
        __syncthreads();                         // the root (nodes[1]) is in global memory, written by this workgroup
        const uint32_t *root = reinterpret_cast<const uint32_t *>(nodes) + 8;
        uint32_t *scratch = reinterpret_cast<uint32_t *>(bufA);
        coin_reseed_draw_quad_wg<FIELD, D>(coin, root, root_out, alpha_out, threadIdx.x, scratch, scratch + 16, reinterpret_cast<int *>(scratch + 24));
    }
}

// The same for 4096 inputs per workgroup, 12 levels per launch.  In merkle_stage_kernel every level below 64 merges still costs
// one wavefront step (levels 4..9: six steps for 63 merges out of 21 per 1024 inputs); here a workgroup takes four 1024-input
// chunks through levels 0..3 one after the other, parks their 4 x 64 digests in LDS and runs the thin levels ONCE for all four:
// 69 wavefront steps for 4095 merges (93 % of lanes busy instead of 76 %), and a 2^23-leaf tree is two launches.
template <class H>
__global__ __launch_bounds__(256) void merkle_stage4k_kernel(const void *in, void *nodes, uint64_t count) {
    __shared__ uint4 bufA[512 * 2];
    __shared__ uint4 bufB[256 * 2];
    __shared__ uint4 top[256 * 2];
    const uint64_t wg = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    auto from_lds = [&](const uint4 *src, uint32_t i, uint32_t (&m)[16]) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = src[4 * i + q];
            m[4 * q] = v.x;
            m[4 * q + 1] = v.y;
            m[4 * q + 2] = v.z;
            m[4 * q + 3] = v.w;
        }
    };
    auto to_lds = [&](uint4 *dst, uint32_t i, const uint32_t (&d)[8]) {
        dst[2 * i] = make_uint4(d[0], d[1], d[2], d[3]);
        dst[2 * i + 1] = make_uint4(d[4], d[5], d[6], d[7]);
    };
    // level d (0-based) of this launch: count >> (d + 1) nodes at heap index (count >> (d + 1)) + position
    for (uint32_t q = 0; q < 4; q++) {
        const uint64_t base = wg * 4096 + q * 1024;             // first input of the chunk
        for (uint32_t i = tid; i < 512; i += 256) {             // level 0: from global
            uint32_t m[16], d[8];
            load_pair(in, (base >> 1) + i, m);
            H::merge(m, d);
            store_digest(nodes, (count >> 1) + (base >> 1) + i, d);
            to_lds(bufA, i, d);
        }
        __syncthreads();
        {                                                       // level 1: 256 merges
            uint32_t m[16], d[8];
            from_lds(bufA, tid, m);
            H::merge(m, d);
            store_digest(nodes, (count >> 2) + (base >> 2) + tid, d);
            to_lds(bufB, tid, d);
        }
        __syncthreads();
        if (tid < 128) {                                        // level 2
            uint32_t m[16], d[8];
            from_lds(bufB, tid, m);
            H::merge(m, d);
            store_digest(nodes, (count >> 3) + (base >> 3) + tid, d);
            to_lds(bufA, tid, d);
        }
        __syncthreads();
        if (tid < 64) {                                         // level 3 -> the chunk's 64 digests
            uint32_t m[16], d[8];
            from_lds(bufA, tid, m);
            H::merge(m, d);
            store_digest(nodes, (count >> 4) + (base >> 4) + tid, d);
            to_lds(top, q * 64 + tid, d);
        }
        __syncthreads();
    }
    uint4 *src = top, *dst = bufA;
    for (uint32_t lvl = 4; lvl < 12; lvl++) {                   // 256 -> 1
        const uint32_t cnt = 4096u >> (lvl + 1);
        if (tid < cnt) {
            uint32_t m[16], d[8];
            from_lds(src, tid, m);
            H::merge(m, d);
            store_digest(nodes, (count >> (lvl + 1)) + ((wg * 4096) >> (lvl + 1)) + tid, d);
            to_lds(dst, tid, d);
        }
        __syncthreads();
        uint4 *t = src;
        src = dst;
        dst = (t == top) ? bufB : t;
    }
}

// Barrier-free Merkle stage for the byte hashers: every WAVEFRONT owns a contiguous run of 128 * 2^T input digests and builds
// the T + 1 levels above them with all 64 lanes busy at every level, no LDS buffer and no workgroup barrier.
//   level 0: lane L merges input pair P0 + 64 b + L of batch b (coalesced 64-byte loads), b = 0 .. 2^T - 1;
//   level l: two level-(l-1) sets X, Y of 64 sibling-adjacent digests (Y follows X in the tree) are re-dealt so that lanes
//            0..31 hold the 32 sibling pairs of X and lanes 32..63 those of Y — ds_bpermute through the LDS crossbar (no LDS
//            memory), one gather per word and side after X / Y have been interleaved by lane parity (a quad-permute DPP move
//            and a select) — and merged: 64 merges,
//            64 consecutive nodes of level l, stored coalesced.
// The sets are produced depth first (a two-iteration loop per level, so the code holds T + 1 compressions, not 2^T), which keeps
// T digests live.  The stage kernels above serialise the thin upper levels on one wavefront behind barriers (a 4096-input
// workgroup's critical path is 28 compressions for 16 per wavefront of work: the BLAKE3 tree ran at half the 55e9
// compressions/s the arithmetic sustains, tools/microbench_blake3.hip); here a launch of 2^23 leaves is 31/32 of the tree at
// full lane utilisation.
// the two level-0 sets under one level-1 set: both loads are issued before the first compression, so that a wavefront has 128
// bytes per lane in flight while it hashes (level 0 is where the input stream enters)
template <class H>
struct WaveTreeLeaves {
    static __device__ __forceinline__ void build2(const void *in, void *nodes, uint64_t count, uint64_t pair0, uint32_t set, uint32_t lane,
                                                  uint32_t (&x)[8], uint32_t (&y)[8]) {
        uint32_t m0[16], m1[16];
        const uint64_t p0 = pair0 + (uint64_t)(2 * set) * 64 + lane, p1 = p0 + 64;
        load_pair(in, p0, m0);
        load_pair(in, p1, m1);
        H::merge(m0, x);
        store_digest(nodes, (count >> 1) + p0, x);
        H::merge(m1, y);
        store_digest(nodes, (count >> 1) + p1, y);
    }
};

template <class H, int L>
struct WaveTree {
    // returns, in d, lane `lane`'s node of the 64-node set number `set` (counted within the wave's run) of level L
    static __device__ __forceinline__ void build(const void *in, void *nodes, uint64_t count, uint64_t pair0, uint32_t set, uint32_t lane,
                                                 uint32_t (&d)[8]) {
        uint32_t x[8], y[8];
        if constexpr (L == 1) {
            WaveTreeLeaves<H>::build2(in, nodes, count, pair0, set, lane, x, y);
        } else {
#pragma unroll 1
            for (uint32_t h = 0; h < 2; h++) {
                uint32_t c[8];
                WaveTree<H, L - 1>::build(in, nodes, count, pair0, 2 * set + h, lane, c);
                if (h == 0) {
#pragma unroll
                    for (int w = 0; w < 8; w++) x[w] = c[w];
                } else {
#pragma unroll
                    for (int w = 0; w < 8; w++) y[w] = c[w];
                }
            }
        }
        // lanes < 32: (X[2 lane], X[2 lane + 1]); lanes >= 32: (Y[2 (lane - 32)], Y[2 (lane - 32) + 1]).
        // u = X on even lanes, Y[s - 1] on odd lanes s;  v = X on odd lanes, Y[s + 1] on even lanes s
        uint32_t m[16];
        const uint32_t j = lane & 31u, hi = lane >> 5;
        const int src_l = (int)((2 * j + hi) << 2), src_r = (int)((2 * j + 1 - hi) << 2);
        const bool odd = lane & 1u;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            // quad_perm [0,0,2,2]: odd lanes read their left neighbour; [1,1,3,3]: even lanes read their right neighbour
            const uint32_t yprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y[w], 0xA0 /* quad_perm:[0,0,2,2] */, 0xf, 0xf, false);
            const uint32_t ynext = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y[w], 0xF5 /* quad_perm:[1,1,3,3] */, 0xf, 0xf, false);
            const uint32_t u = odd ? yprev : x[w];
            const uint32_t v = odd ? x[w] : ynext;
            m[w] = (uint32_t)__builtin_amdgcn_ds_bpermute(src_l, (int)u);
            m[8 + w] = (uint32_t)__builtin_amdgcn_ds_bpermute(src_r, (int)v);
        }
        H::merge(m, d);
        store_digest(nodes, (count >> (L + 1)) + (pair0 >> L) + (uint64_t)set * 64 + lane, d);
    }
};
template <class H>
struct WaveTree<H, 0> {
    static __device__ __forceinline__ void build(const void *in, void *nodes, uint64_t count, uint64_t pair0, uint32_t set, uint32_t lane,
                                                 uint32_t (&d)[8]) {
        uint32_t m[16];
        const uint64_t pr = pair0 + (uint64_t)set * 64 + lane;
        load_pair(in, pr, m);
        H::merge(m, d);
        store_digest(nodes, (count >> 1) + pr, d);
    }
};

template <class H, int T>
__global__ __launch_bounds__(256) void merkle_wave_kernel(const void *in, void *nodes, uint64_t count) {
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t d[8];
    WaveTree<H, T>::build(in, nodes, count, wave << (6 + T), 0, lane, d);
}

struct CoinTailArgs {
    int field;
    uint32_t D;
    void *coin, *root_out, *alpha_out;
    bool done;
};

// the final single-workgroup stage launch with the coin step appended, for the (hasher, field, degree) combinations that have a kernel
template <class H>
bool launch_final_with_coin(wf_ctx *ctx, const void *in, void *nodes, uint64_t count, uint32_t arg, CoinTailArgs *ct) {
    if constexpr (!H::QUAD_MERGE) {
        return false;
    } else {
        CoinState *c = (CoinState *)ct->coin;
        uint32_t *ro = (uint32_t *)ct->root_out;
        uint64_t *ao = (uint64_t *)ct->alpha_out;
#define WF_MC(F, DD) if (ct->field == F && ct->D == DD) { hipLaunchKernelGGL((merkle_stage_coin_kernel<H, 1024, F, DD>), dim3(1), dim3(1024), 0, ctx->stream, in, nodes, count, arg, c, ro, ao); return true; }
        WF_MC(WF_FIELD_F64, 1) WF_MC(WF_FIELD_F64, 2) WF_MC(WF_FIELD_F64, 3) WF_MC(WF_FIELD_F128, 1) WF_MC(WF_FIELD_F128, 2)
#undef WF_MC
        return false;
    }
}

// the one-launch tree of merkle_finish_kernel (BLAKE3-256: it needs the four-lane merge for its thin levels and coin step)
template <class H>
int launch_finish(wf_ctx *ctx, const void *in, void *nodes, uint64_t count, CoinTailArgs *ct, bool *launched) {
    *launched = false;
    if constexpr (!H::QUAD_MERGE) {
        return WF_OK;
    } else {
        // the ticket ring was allocated and zeroed by wf_ctx_create (no allocation, no synchronisation on the launch path)
        const bool big = count > (1u << 18);                       // 2^19, 2^20 inputs: 4096 per workgroup
        const dim3 grid((uint32_t)(count >> (big ? 12 : 10)));
        // a ring of ticket words: a caller that moves the context to another stream (wf_ctx_set_stream) may have two trees in flight
        const uint32_t slot = ctx->tree_ticket_next++ % WF_TREE_TICKETS;
        uint32_t *tk = (uint32_t *)ctx->d_tree_ticket + slot;
        const uint32_t ep = ctx->tree_ticket_epoch[slot];
        ctx->tree_ticket_epoch[slot] = (ep + 1) & 0xfffu;
        uint32_t *st = ctx->d_status;
        if (ct) {
            CoinState *c = (CoinState *)ct->coin;
            uint32_t *ro = (uint32_t *)ct->root_out;
            uint64_t *ao = (uint64_t *)ct->alpha_out;
#define WF_MF(F, DD)                                                                                                                                        \
    if (ct->field == F && ct->D == DD) {                                                                                                                    \
        if (big) hipLaunchKernelGGL((merkle_finish_kernel<H, F, DD, true, 12>), grid, dim3(1024), 0, ctx->stream, in, nodes, count, tk, ep, st, c, ro, ao);  \
        else hipLaunchKernelGGL((merkle_finish_kernel<H, F, DD, true, 10>), grid, dim3(1024), 0, ctx->stream, in, nodes, count, tk, ep, st, c, ro, ao);      \
        ct->done = true;                                                                                                                                    \
        *launched = true;                                                                                                                                   \
        return WF_OK;                                                                                                                                       \
    }
            WF_MF(WF_FIELD_F64, 1) WF_MF(WF_FIELD_F64, 2) WF_MF(WF_FIELD_F64, 3) WF_MF(WF_FIELD_F128, 1) WF_MF(WF_FIELD_F128, 2)
#undef WF_MF
        }
        if (big)
            hipLaunchKernelGGL((merkle_finish_kernel<H, WF_FIELD_F64, 1, false, 12>), grid, dim3(1024), 0, ctx->stream, in, nodes, count, tk, ep, st,
                               (CoinState *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr);
        else
            hipLaunchKernelGGL((merkle_finish_kernel<H, WF_FIELD_F64, 1, false, 10>), grid, dim3(1024), 0, ctx->stream, in, nodes, count, tk, ep, st,
                               (CoinState *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr);
        *launched = true;
        return WF_OK;
    }
}

template <class H>
int launch_merkle(wf_ctx *ctx, const void *leaves, uint64_t num_leaves, void *nodes, CoinTailArgs *ct = nullptr) {
    // nodes[0] = Digest::default(): written by the stage launch that finishes the tree; the single-level hashers keep the fill
    if (H::STAGE_LEVELS == 1) WF_HIP(hipMemsetAsync(nodes, 0, 32, ctx->stream));
    const uint8_t *in = (const uint8_t *)leaves;
    uint64_t count = num_leaves;
    while (count > 1) {
        if (H::STAGE_LEVELS == 1) {
            // one level: nodes[count/2 + i] = merge(in[2i], in[2i+1]), one merge per lane
            const uint64_t half = count >> 1;
            const uint64_t blocks = (half + 255) / 256;
            if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
            wf_prof_begin(ctx, H::merkle_name());
            bool coop = false;
            if constexpr (H::COOP) {
                if (half <= rcoop::COOP_MAX) {           // the upper levels: one 0.2 ms wave per 64 merges otherwise
                    coop = true;
                    hipLaunchKernelGGL((rcoop::merge_kernel<typename H::Coop>), dim3((uint32_t)((half + 15) / 16)), dim3(256), 0, ctx->stream,
                                       (const uint64_t *)in, half, (uint64_t *)((uint8_t *)nodes + half * 32));
                }
            }
            if (!coop)
                hipLaunchKernelGGL(merge_batch_kernel<H>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, (const void *)in, half,
                                   (void *)((uint8_t *)nodes + half * 32));
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            count = half;
            in = (const uint8_t *)nodes + count * 32;
            continue;
        }
#ifndef WF_NO_MERKLE_FINISH
        if (H::QUAD_MERGE && (count == (1u << 19) || count == (1u << 20))) {      // 4096 inputs per workgroup, the rest as below: one launch
            bool launched = false;
            wf_prof_begin(ctx, H::merkle_name());
            WF_TRY(launch_finish<H>(ctx, (const void *)in, nodes, count, ct, &launched));
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            return WF_OK;
        }
#endif
        if (H::WAVE_TREE && count >= (1u << 20)) {
            // T + 1 levels per launch, 128 * 2^T inputs per wavefront, all lanes busy at every level; T as large as still leaves
            // four wavefronts per SIMD (4096 on the chip): 2^23 inputs and up take five levels per launch
            uint32_t lg = 0;
            while ((2ull << lg) <= count) lg++;
            const uint32_t T = lg >= 23 ? 4 : lg - 19;
            const uint64_t waves = count >> (7 + T);
            if ((waves + 3) / 4 > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
            wf_prof_begin(ctx, H::merkle_name());
            if constexpr (H::WAVE_TREE) {
                const dim3 grid((uint32_t)((waves + 3) / 4));
                switch (T) {
                    case 1: hipLaunchKernelGGL((merkle_wave_kernel<H, 1>), grid, dim3(256), 0, ctx->stream, (const void *)in, nodes, count); break;
                    case 2: hipLaunchKernelGGL((merkle_wave_kernel<H, 2>), grid, dim3(256), 0, ctx->stream, (const void *)in, nodes, count); break;
                    case 3: hipLaunchKernelGGL((merkle_wave_kernel<H, 3>), grid, dim3(256), 0, ctx->stream, (const void *)in, nodes, count); break;
                    default: hipLaunchKernelGGL((merkle_wave_kernel<H, 4>), grid, dim3(256), 0, ctx->stream, (const void *)in, nodes, count); break;
                }
            }
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            count >>= T + 1;
            in = (const uint8_t *)nodes + count * 32;
            continue;
        }
        if (H::STAGE_LEVELS >= 10 && count >= (1u << 20)) {      // 12 levels per launch, 4096 inputs per workgroup: only when that still fills the chip (>= 256 workgroups)
            const uint64_t wgs4 = count >> 12;
            if (wgs4 > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
            wf_prof_begin(ctx, H::merkle_name());
            hipLaunchKernelGGL(merkle_stage4k_kernel<H>, dim3((uint32_t)wgs4), dim3(256), 0, ctx->stream, (const void *)in, nodes, count);
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            count = wgs4;
            in = (const uint8_t *)nodes + count * 32;
            continue;
        }
        uint32_t log_ch = 0;
        while ((1ull << log_ch) < count && log_ch < H::STAGE_LEVELS) log_ch++;
        const uint64_t wgs = count >> log_ch;
        if (wgs > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
#ifndef WF_NO_MERKLE_FINISH
        if (H::QUAD_MERGE && log_ch == 10 && wgs >= 2 && wgs <= 256) {   // stage + the levels above it + the coin step: one launch
            bool launched = false;
            wf_prof_begin(ctx, H::merkle_name());
            WF_TRY(launch_finish<H>(ctx, (const void *)in, nodes, count, ct, &launched));
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            return WF_OK;
        }
#endif
        wf_prof_begin(ctx, H::merkle_name());
        const uint32_t arg = log_ch | (wgs == 1 ? 0x80000000u : 0u);     // the last launch of the tree writes nodes[0] too
        bool wide = false;
        if (ct && wgs == 1 && launch_final_with_coin<H>(ctx, (const void *)in, nodes, count, arg, ct)) {
            ct->done = true;
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            count = wgs;
            in = (const uint8_t *)nodes + count * 32;
            continue;
        }
        if constexpr (H::QUAD_MERGE) {
            if (wgs <= 256) {
                wide = true;
                hipLaunchKernelGGL((merkle_stage_kernel<H, 1024>), dim3((uint32_t)wgs), dim3(1024), 0, ctx->stream, (const void *)in, nodes, count, arg);
            }
        }
        if (!wide) hipLaunchKernelGGL((merkle_stage_kernel<H, 256>), dim3((uint32_t)wgs), dim3(256), 0, ctx->stream, (const void *)in, nodes, count, arg);
        wf_prof_end(ctx);
        WF_HIP(hipGetLastError());
        count = wgs;
        in = (const uint8_t *)nodes + count * 32;  // this stage's top level = next stage's inputs
    }
    return WF_OK;
}

}  // namespace

extern "C" int wf_merkle_build(wf_ctx *ctx, int hash, const void *d_leaves, uint64_t num_leaves, void *d_nodes) {
    WF_ENTER(ctx);
    if (!ctx || !d_leaves || !d_nodes) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (num_leaves < 2) return WF_ERR_TOO_FEW_LEAVES;
    if (num_leaves & (num_leaves - 1)) return WF_ERR_NOT_POWER_OF_TWO;
    return with_hasher(hash, [&](auto h) { return launch_merkle<decltype(h)>(ctx, d_leaves, num_leaves, d_nodes); });
}

// MerkleTree::new followed, in the tree's last launch, by coin.reseed(root) and alpha = coin.draw::<E>() (the two channel calls of
// FriProver::build_layer).  *done = 0: no fused kernel for this hasher / field / degree — the tree is built, the caller runs
// wf_coin_reseed_draw itself.
int wf_merkle_build_coin(wf_ctx *ctx, int hash, const void *d_leaves, uint64_t num_leaves, void *d_nodes, int field, uint32_t ext_degree, void *d_coin,
                         void *d_root_out, void *d_alpha_out, int *done) {
    *done = 0;
    if (!ctx || !d_leaves || !d_nodes || !d_coin || !d_alpha_out) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (num_leaves < 2) return WF_ERR_TOO_FEW_LEAVES;
    if (num_leaves & (num_leaves - 1)) return WF_ERR_NOT_POWER_OF_TWO;
    CoinTailArgs ct{field, ext_degree, d_coin, d_root_out, d_alpha_out, false};
    WF_TRY(with_hasher(hash, [&](auto h) { return launch_merkle<decltype(h)>(ctx, d_leaves, num_leaves, d_nodes, &ct); }));
    *done = ct.done ? 1 : 0;
    return WF_OK;
}
