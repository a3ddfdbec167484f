// crypto::DefaultRandomCoin with its state in device memory (crypto/src/random/default.rs): wf_coin_* of the C ABI.
#include "hashers.cuh"
#include "coin_state.cuh"

namespace {

// ---- crypto::DefaultRandomCoin with its state in device memory (crypto/src/random/default.rs) -------------------------------
// One lane: every step of the coin is one small hash that depends on the previous one.  What this buys is that a chain of
// commit -> reseed -> draw -> use (the FRI layers) is queued on the stream without a host round trip per link.
// reseed (default.rs:150-153): seed = merge(seed, data), counter = 0; the digest is also copied to root_out when given
template <class H>
__global__ void coin_reseed_kernel(CoinState *c, const uint32_t *digest, uint32_t *root_out) {
    uint32_t m[16], d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        m[i] = c->seed[i];
        m[8 + i] = digest[i];
    }
    H::merge(m, d);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c->seed[i] = d[i];
        if (root_out) root_out[i] = digest[i];
    }
    c->counter = 0;
}

// draw::<E>() `count` times (default.rs:185-199): next() = merge_with_int(seed, ++counter) until the bytes decode
template <class H, int FIELD, int D>
__global__ void coin_draw_kernel(CoinState *c, uint32_t count, uint64_t *out) {
    constexpr int WORDS = (FIELD == WF_FIELD_F128 ? 2 : 1) * D;
    uint32_t seed[8];
#pragma unroll
    for (int i = 0; i < 8; i++) seed[i] = c->seed[i];
    uint64_t counter = c->counter;
    for (uint32_t k = 0; k < count; k++) {
        bool ok = false;
        for (int tries = 0; tries < 1000 && !ok; tries++) {
            uint32_t d[8], b[8];
            counter++;
            H::merge_with_int(seed, counter, d);
            H::as_bytes(d, b);
            ok = coin_element<FIELD, D>(b, out + (uint64_t)k * WORDS);
        }
        if (!ok) {
            c->failed |= 1u;
            break;
        }
    }
    c->counter = counter;
}

// commit_fri_layer + draw_fri_alpha in one launch: reseed with `digest`, then one draw
template <class H, int FIELD, int D>
__global__ void coin_reseed_draw_kernel(CoinState *c, const uint32_t *digest, uint32_t *root_out, uint64_t *out) {
    coin_reseed_draw_lane<H, FIELD, D>(c, digest, root_out, out);
}

// The same for Blake3_256 on FOUR lanes (b3::quad_hash_block): the two compressions of a reseed + draw are the whole kernel, and
// each is ~300 dependent instructions on a quad instead of ~800 on one lane.  Messages go through LDS (seed || digest, then
// seed' || counter); lane 0 decodes the drawn bytes.
template <int FIELD, int D>
__global__ __launch_bounds__(64) void coin_reseed_draw_quad_kernel(CoinState *c, const uint32_t *digest, uint32_t *root_out, uint64_t *out) {
    __shared__ uint32_t msg[16], drawn[8];
    __shared__ int ok_flag;
    coin_reseed_draw_quad_wg<FIELD, D>(c, digest, root_out, out, threadIdx.x, msg, drawn, &ok_flag);
}


#ifndef WF_COOP_COIN
#define WF_COOP_COIN 1       // 0: the Rescue coin on one lane as in round 4 (A/B builds)
#endif
// ---- the Rescue family: a coin step on ONE lane is a whole permutation (~6400 dependent modular multiplications, 0.2 ms), which is
// why round 4 kept the host transcript for these hashers.  Here a step runs on a 16-lane group with one state word per lane
// (rescue_coop.cuh: the S-boxes lane-local, the MDS rows through LDS), ~17 us, and independent steps — the draws of one
// wf_coin_draw, the query positions — on as many groups as there are steps.  crypto/src/random/default.rs:82-248.
template <class C>
__device__ __forceinline__ void coop_digest(uint64_t word, int i, volatile uint64_t *grp, uint32_t (&d)[8]) {
    // the digest words sit on the group's output lanes: publish them (the group's exchange slots are free once the permutation has
    // returned; one wavefront's DS operations execute in order) and let every lane of the group read all four
    if (C::out_lane(i)) grp[C::out_word(i)] = word;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint64_t v = grp[w];
        d[2 * w] = (uint32_t)v;
        d[2 * w + 1] = (uint32_t)(v >> 32);
    }
}

// draw::<E>() `count` times on the groups of ONE workgroup, speculating over COUNTERS, not over draws: the groups hash the counters
// base + 1 .. base + ng at once, then one lane walks the window in counter order and hands every value that decodes to the next open
// draw — the reference's loop word for word (a value that does not decode costs the current draw one of its 1000 tries,
// default.rs:185-199), whatever the rejection rate: ~2^-32 per base element for f64, ~2^-88 for f128, but ~3/4 of all 8-byte values
// for f62 (M ~ 2^62; the reference's try_from rejects v >= M), where speculating over draws fell back to one lane nearly always.
// EVERY thread of the workgroup calls it (it synchronises); `seed` is the coin's seed, counter0 its counter before the first draw.
struct CoopDrawLds {
    uint64_t cand[64][4];      // the decoded element of each group's counter
    int ok[64];                // ... and whether it decoded
    uint32_t done, miss, stop; // walker state: draws served, consecutive misses of the open draw, finished
};
template <class H, int FIELD, int D>
__device__ __forceinline__ void coop_draw_windows(CoinState *c, const uint32_t (&seed)[8], uint64_t counter0, uint32_t count, uint64_t *out, int i, int ii,
                                                  volatile uint64_t *grp, CoopDrawLds *w) {
    typedef typename H::Coop C;
    constexpr int WORDS = (FIELD == WF_FIELD_F128 ? 2 : 1) * D;
    static_assert(WORDS <= 4, "32 digest bytes");
    const uint32_t g = threadIdx.x / rcoop::GROUP, ng = blockDim.x / rcoop::GROUP;
    if (threadIdx.x == 0) w->done = w->miss = w->stop = 0;
    uint64_t base = counter0;
    for (;;) {
        __syncthreads();
        if (*(volatile uint32_t *)&w->stop) break;
        uint32_t d[8], b[8];
        coop_digest<C>(C::merge_with_int(seed, base + 1 + g, i, ii, grp), i, grp, d);
        if (i == 0) {
            uint64_t e[WORDS];
            H::as_bytes(d, b);
            w->ok[g] = coin_element<FIELD, D>(b, e) ? 1 : 0;
#pragma unroll
            for (int t = 0; t < WORDS; t++) w->cand[g][t] = e[t];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t done = w->done, miss = w->miss, stop = 0;
            for (uint32_t k = 0; k < ng && !stop; k++) {
                if (w->ok[k]) {
#pragma unroll
                    for (int t = 0; t < WORDS; t++) out[(uint64_t)done * WORDS + t] = w->cand[k][t];
                    miss = 0;
                    if (++done == count) stop = 1;
                } else if (++miss >= 1000) {
                    c->failed |= 1u;                     // FailedToDrawFieldElement (bit 0; bit 1 = nonce not found)
                    stop = 1;
                }
                if (stop) c->counter = base + 1 + k;     // the last counter the reference's loop would have consumed
            }
            w->done = done;
            w->miss = miss;
            w->stop = stop;
        }
        base += ng;
    }
}

#define WF_COOP_PROLOGUE                                                              \
    typedef typename H::Coop C;                                                       \
    const int i = threadIdx.x & (rcoop::GROUP - 1);                                   \
    const int ii = i < C::S::W ? i : C::S::W - 1;                                     \
    volatile uint64_t *grp = coop_lds + (threadIdx.x & ~(rcoop::GROUP - 1));

// reseed: one group.  seed = merge(seed, digest), counter = 0
template <class H>
__global__ __launch_bounds__(16) void coin_reseed_coop_kernel(CoinState *c, const uint32_t *digest, uint32_t *root_out) {
    __shared__ uint64_t coop_lds[16];
    __shared__ uint64_t pair[8];
    WF_COOP_PROLOGUE
    if (i < 4) pair[i] = (uint64_t)c->seed[2 * i] | ((uint64_t)c->seed[2 * i + 1] << 32);
    else if (i < 8) pair[i] = (uint64_t)digest[2 * (i - 4)] | ((uint64_t)digest[2 * (i - 4) + 1] << 32);
    if (root_out && i < 8) root_out[i] = digest[i];
    __syncthreads();
    uint32_t d[8];
    coop_digest<C>(C::merge(pair, i, ii, grp), i, grp, d);
    if (i < 8) c->seed[i] = d[i];
    if (i == 0) c->counter = 0;
}

// draw::<E>() `count` times: coop_draw_windows on up to 64 groups
template <class H, int FIELD, int D>
__global__ __launch_bounds__(1024) void coin_draw_coop_kernel(CoinState *c, uint32_t count, uint64_t *out) {
    __shared__ uint64_t coop_lds[1024];
    __shared__ CoopDrawLds win;
    WF_COOP_PROLOGUE
    uint32_t seed[8];
#pragma unroll
    for (int w = 0; w < 8; w++) seed[w] = c->seed[w];
    const uint64_t counter0 = c->counter;
    __syncthreads();                                     // everybody has read the counter before the walker writes it
    coop_draw_windows<H, FIELD, D>(c, seed, counter0, count, out, i, ii, grp, &win);
}

// commit_fri_layer + draw_fri_alpha: reseed on group 0, then one draw over a window of blockDim / 16 counters (one group for the
// 64-bit and 128-bit fields; for f62, whose try decodes with probability 4^-D, 8 / 32 / 64 of them)
template <class H, int FIELD, int D>
__global__ __launch_bounds__(1024) void coin_reseed_draw_coop_kernel(CoinState *c, const uint32_t *digest, uint32_t *root_out, uint64_t *out) {
    __shared__ uint64_t coop_lds[1024];
    __shared__ uint64_t pair[8];
    __shared__ uint32_t ns[8];
    __shared__ CoopDrawLds win;
    WF_COOP_PROLOGUE
    if (threadIdx.x < 4) pair[i] = (uint64_t)c->seed[2 * i] | ((uint64_t)c->seed[2 * i + 1] << 32);
    else if (threadIdx.x < 8) pair[i] = (uint64_t)digest[2 * (i - 4)] | ((uint64_t)digest[2 * (i - 4) + 1] << 32);
    if (root_out && threadIdx.x < 8) root_out[i] = digest[i];
    __syncthreads();
    if (threadIdx.x < rcoop::GROUP) {
        uint32_t d[8];
        coop_digest<C>(C::merge(pair, i, ii, grp), i, grp, d);
        if (i < 8) {
            c->seed[i] = d[i];
            ns[i] = d[i];
        }
    }
    __syncthreads();
    uint32_t seed[8];
#pragma unroll
    for (int w = 0; w < 8; w++) seed[w] = ns[w];
    coop_draw_windows<H, FIELD, D>(c, seed, 0, 1, out, i, ii, grp, &win);
}

// draw_integers: the new seed on group 0, then one value per group
template <class H>
__global__ __launch_bounds__(1024) void coin_draw_integers_coop_kernel(CoinState *c, const unsigned long long *nonce, uint32_t num_values, uint64_t mask,
                                                                       uint64_t *out) {
    __shared__ uint64_t coop_lds[1024];
    __shared__ uint32_t ns[8];
    WF_COOP_PROLOGUE
    const uint32_t g = threadIdx.x / rcoop::GROUP, ng = blockDim.x / rcoop::GROUP;
    if (g == 0) {
        uint32_t seed[8], d[8];
#pragma unroll
        for (int w = 0; w < 8; w++) seed[w] = c->seed[w];
        coop_digest<C>(C::merge_with_int(seed, (uint64_t)*nonce, i, ii, grp), i, grp, d);
        if (i < 8) ns[i] = d[i];
    }
    __syncthreads();
    uint32_t seed[8];
#pragma unroll
    for (int w = 0; w < 8; w++) seed[w] = ns[w];
    for (uint32_t v = g; v < num_values; v += ng) {
        uint32_t d[8], b[8];
        coop_digest<C>(C::merge_with_int(seed, (uint64_t)v + 1, i, ii, grp), i, grp, d);
        if (i == 0) {
            H::as_bytes(d, b);
            out[v] = ((uint64_t)b[0] | ((uint64_t)b[1] << 32)) & mask;
        }
    }
    __syncthreads();
    if (threadIdx.x < 8) c->seed[threadIdx.x] = ns[threadIdx.x];
    if (threadIdx.x == 0) c->counter = num_values;
}
#undef WF_COOP_PROLOGUE

template <class H>
int launch_coin_reseed_draw(wf_ctx *ctx, int field, uint32_t D, CoinState *c, const uint32_t *dg, uint32_t *cp, uint64_t *o) {
    if constexpr (H::QUAD_MERGE) {
#define WF_RQ(FIELD, DEG) hipLaunchKernelGGL((coin_reseed_draw_quad_kernel<FIELD, DEG>), dim3(1), dim3(64), 0, ctx->stream, c, dg, cp, o)
        if (field == WF_FIELD_F128) {
            if (D == 1) WF_RQ(WF_FIELD_F128, 1);
            else WF_RQ(WF_FIELD_F128, 2);
            return WF_OK;
        } else if (field == WF_FIELD_F64) {
            if (D == 1) WF_RQ(WF_FIELD_F64, 1);
            else if (D == 2) WF_RQ(WF_FIELD_F64, 2);
            else WF_RQ(WF_FIELD_F64, 3);
            return WF_OK;
        }
#undef WF_RQ
    }
    if constexpr (H::COOP && WF_COOP_COIN != 0) {
#define WF_RC(FIELD, DEG) \
    hipLaunchKernelGGL((coin_reseed_draw_coop_kernel<H, FIELD, DEG>), dim3(1), dim3(FIELD == WF_FIELD_F62 ? (DEG == 1 ? 128 : DEG == 2 ? 512 : 1024) : 16), 0, ctx->stream, c, dg, cp, o)
        if (field == WF_FIELD_F128) {
            if (D == 1) WF_RC(WF_FIELD_F128, 1);
            else WF_RC(WF_FIELD_F128, 2);
        } else if (field == WF_FIELD_F64) {
            if (D == 1) WF_RC(WF_FIELD_F64, 1);
            else if (D == 2) WF_RC(WF_FIELD_F64, 2);
            else WF_RC(WF_FIELD_F64, 3);
        } else {
            if (D == 1) WF_RC(WF_FIELD_F62, 1);
            else if (D == 2) WF_RC(WF_FIELD_F62, 2);
            else WF_RC(WF_FIELD_F62, 3);
        }
#undef WF_RC
        return WF_OK;
    }
#define WF_RD(FIELD, DEG) hipLaunchKernelGGL((coin_reseed_draw_kernel<H, FIELD, DEG>), dim3(1), dim3(1), 0, ctx->stream, c, dg, cp, o)
    if (field == WF_FIELD_F128) {
        if (D == 1) WF_RD(WF_FIELD_F128, 1);
        else WF_RD(WF_FIELD_F128, 2);
    } else if (field == WF_FIELD_F64) {
        if (D == 1) WF_RD(WF_FIELD_F64, 1);
        else if (D == 2) WF_RD(WF_FIELD_F64, 2);
        else WF_RD(WF_FIELD_F64, 3);
    } else {
        if (D == 1) WF_RD(WF_FIELD_F62, 1);
        else if (D == 2) WF_RD(WF_FIELD_F62, 2);
        else WF_RD(WF_FIELD_F62, 3);
    }
#undef WF_RD
    return WF_OK;
}

template <class H>
int launch_coin_draw(wf_ctx *ctx, int field, uint32_t D, CoinState *c, uint32_t count, uint64_t *o) {
    if constexpr (H::COOP && WF_COOP_COIN != 0) {
        // as many 16-lane groups as counters the request is expected to consume (f62: 4^D per draw), up to the 64 of one workgroup
        const uint64_t want = field == WF_FIELD_F62 ? ((uint64_t)count << (2 * D)) + 4 : count;
        const uint32_t groups = want < 64 ? (uint32_t)want : 64;
#define WF_DC(FIELD, DEG) hipLaunchKernelGGL((coin_draw_coop_kernel<H, FIELD, DEG>), dim3(1), dim3(16 * groups), 0, ctx->stream, c, count, o)
        if (field == WF_FIELD_F128) {
            if (D == 1) WF_DC(WF_FIELD_F128, 1);
            else WF_DC(WF_FIELD_F128, 2);
        } else if (field == WF_FIELD_F64) {
            if (D == 1) WF_DC(WF_FIELD_F64, 1);
            else if (D == 2) WF_DC(WF_FIELD_F64, 2);
            else WF_DC(WF_FIELD_F64, 3);
        } else {
            if (D == 1) WF_DC(WF_FIELD_F62, 1);
            else if (D == 2) WF_DC(WF_FIELD_F62, 2);
            else WF_DC(WF_FIELD_F62, 3);
        }
#undef WF_DC
        return WF_OK;
    }
#define WF_DRAW(FIELD, DEG) hipLaunchKernelGGL((coin_draw_kernel<H, FIELD, DEG>), dim3(1), dim3(1), 0, ctx->stream, c, count, o)
    if (field == WF_FIELD_F128) {
        if (D == 1) WF_DRAW(WF_FIELD_F128, 1);
        else WF_DRAW(WF_FIELD_F128, 2);
    } else if (field == WF_FIELD_F64) {
        if (D == 1) WF_DRAW(WF_FIELD_F64, 1);
        else if (D == 2) WF_DRAW(WF_FIELD_F64, 2);
        else WF_DRAW(WF_FIELD_F64, 3);
    } else {
        if (D == 1) WF_DRAW(WF_FIELD_F62, 1);
        else if (D == 2) WF_DRAW(WF_FIELD_F62, 2);
        else WF_DRAW(WF_FIELD_F62, 3);
    }
#undef WF_DRAW
    return WF_OK;
}


// ProverChannel::grind_query_seed (prover/src/channel.rs:169-185) against the device coin: lanes test nonces first + gid with the seed
// READ FROM THE COIN; the minimum of the passing nonces goes to *best.  Launched as a fixed sequence of batches in increasing
// nonce order: a batch returns at once when an earlier one has found a nonce (stream order makes *best final by then), so the
// result is the serial reference's answer — the smallest nonce — and nothing comes back to the host between the batches.
template <class H>
__global__ __launch_bounds__(256) void coin_grind_kernel(const CoinState *c, uint64_t first, uint64_t count, uint32_t factor,
                                                         unsigned long long *best) {
    if (*(volatile unsigned long long *)best != ~0ull && *(volatile unsigned long long *)best < first) return;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    uint32_t seed[8], d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) seed[i] = c->seed[i];
    H::merge_with_int(seed, first + gid, d);
    const uint64_t h = H::head(d);
    const uint32_t tz = h ? (uint32_t)__builtin_ctzll(h) : 64u;
    if (tz >= factor) atomicMin(best, (unsigned long long)(first + gid));
}

// no nonce in the searched range: bit 1 of the coin's failed flag (bit 0 = a draw ran out of tries), so that the host can tell the
// reference's "nonce not found" panic (prover/src/channel.rs:169-185) from FailedToDrawFieldElement; wf_coin_read -> WF_ERR_NOT_FOUND
__global__ void coin_grind_check_kernel(CoinState *c, const unsigned long long *best) {
    if (*best == ~0ull) c->failed |= 2u;
}

// RandomCoin::draw_integers (crypto/src/random/default.rs:209-248): seed = merge_with_int(seed, nonce), counter = 0, then
// value i = the first 8 bytes (little-endian) of next() = merge_with_int(seed, i + 1), masked to the domain.  The values are
// independent given the new seed: one lane each.  The counter ends at num_values, as after the reference's loop.
template <class H>
__global__ __launch_bounds__(256) void coin_draw_integers_kernel(CoinState *c, const unsigned long long *nonce, uint32_t num_values, uint64_t mask,
                                                                 uint64_t *out) {
    __shared__ uint32_t ns[8];
    if (threadIdx.x == 0) {
        uint32_t seed[8], d[8];
#pragma unroll
        for (int i = 0; i < 8; i++) seed[i] = c->seed[i];
        H::merge_with_int(seed, (uint64_t)*nonce, d);
#pragma unroll
        for (int i = 0; i < 8; i++) ns[i] = d[i];
    }
    __syncthreads();
    uint32_t seed[8], d[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) seed[i] = ns[i];
    if (threadIdx.x < num_values) {
        H::merge_with_int(seed, (uint64_t)threadIdx.x + 1, d);
        H::as_bytes(d, b);
        out[threadIdx.x] = ((uint64_t)b[0] | ((uint64_t)b[1] << 32)) & mask;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) c->seed[i] = ns[i];
        c->counter = num_values;
    }
}

}  // namespace

extern "C" int wf_coin_grind(wf_ctx *ctx, int hash, const void *d_coin, uint32_t grinding_factor, uint32_t log_max_tries, void *d_nonce) {
    WF_ENTER(ctx);
    if (!ctx || !d_coin || !d_nonce || grinding_factor > 32 || log_max_tries > 40) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    // nonces 1 .. 2^log_max_tries in batches of up to 2^22 lanes (Rescue: 2^20), every batch behind the early-out of the kernel
    uint32_t lg = grinding_factor + 1;
    if (lg < 16) lg = 16;
    if (lg > 22) lg = 22;
    if ((hash == WF_HASH_RP64_256 || hash == WF_HASH_RPJIVE64_256 || hash == WF_HASH_RP62_248) && lg > 20) lg = 20;
    if (lg > log_max_tries) lg = log_max_tries;
    const uint64_t batch = 1ull << lg, total = 1ull << log_max_tries;
    if (total / batch > 4096) return WF_ERR_INVALID_ARG;                 // a queue of launches, not a search loop: bound it
    unsigned long long *best = (unsigned long long *)d_nonce;
    WF_HIP(hipMemsetAsync(best, 0xff, 8, ctx->stream));
    return with_hasher(hash, [&](auto h) -> int {
        typedef decltype(h) H;
        wf_prof_begin(ctx, H::grind_name());
        for (uint64_t first = 1; first <= total; first += batch) {
            const uint64_t count = total - first + 1 < batch ? total - first + 1 : batch;
            hipLaunchKernelGGL(coin_grind_kernel<H>, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, (const CoinState *)d_coin, first,
                               count, grinding_factor, best);
        }
        hipLaunchKernelGGL(coin_grind_check_kernel, dim3(1), dim3(1), 0, ctx->stream, (CoinState *)d_coin, (const unsigned long long *)best);
        wf_prof_end(ctx);
        WF_HIP(hipGetLastError());
        return (int)WF_OK;
    });
}

extern "C" int wf_coin_draw_integers(wf_ctx *ctx, int hash, void *d_coin, const void *d_nonce, uint32_t num_values, uint32_t log_domain_size,
                                     void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_coin || !d_nonce || !d_out || num_values == 0 || num_values > 256 || log_domain_size == 0 || log_domain_size > 63)
        return WF_ERR_INVALID_ARG;
    if ((uint64_t)num_values >= (1ull << log_domain_size)) return WF_ERR_INVALID_ARG;      // "number of values must be smaller than domain size"
    WF_TRY(check_hash(hash));
    wf_prof_begin(ctx, "coin");
    WF_TRY(with_hasher(hash, [&](auto h) {
        typedef decltype(h) H;
        if constexpr (H::COOP && WF_COOP_COIN != 0) {
            const uint32_t groups = num_values < 64 ? num_values : 64;
            hipLaunchKernelGGL(coin_draw_integers_coop_kernel<H>, dim3(1), dim3(16 * groups), 0, ctx->stream, (CoinState *)d_coin,
                               (const unsigned long long *)d_nonce, num_values, (1ull << log_domain_size) - 1, (uint64_t *)d_out);
        } else {
            hipLaunchKernelGGL(coin_draw_integers_kernel<H>, dim3(1), dim3(256), 0, ctx->stream, (CoinState *)d_coin,
                               (const unsigned long long *)d_nonce, num_values, (1ull << log_domain_size) - 1, (uint64_t *)d_out);
        }
        return (int)WF_OK;
    }));
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_coin_init(wf_ctx *ctx, void *d_coin, const void *h_seed) {
    WF_ENTER(ctx);
    if (!ctx || !d_coin || !h_seed) return WF_ERR_INVALID_ARG;
    CoinState st;
    memset(&st, 0, sizeof(st));
    memcpy(st.seed, h_seed, 32);
    uint8_t image[WF_COIN_BYTES] = {0};
    memcpy(image, &st, sizeof(st));
    return wf_copy_h2d(ctx, d_coin, image, WF_COIN_BYTES);    // synchronises: `image` is on this frame
}

extern "C" int wf_coin_reseed(wf_ctx *ctx, int hash, void *d_coin, const void *d_digest, void *d_digest_copy) {
    WF_ENTER(ctx);
    if (!ctx || !d_coin || !d_digest) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    wf_prof_begin(ctx, "coin");
    WF_TRY(with_hasher(hash, [&](auto h) {
        typedef decltype(h) H;
        if constexpr (H::COOP && WF_COOP_COIN != 0)
            hipLaunchKernelGGL(coin_reseed_coop_kernel<H>, dim3(1), dim3(16), 0, ctx->stream, (CoinState *)d_coin, (const uint32_t *)d_digest,
                               (uint32_t *)d_digest_copy);
        else
            hipLaunchKernelGGL(coin_reseed_kernel<H>, dim3(1), dim3(1), 0, ctx->stream, (CoinState *)d_coin, (const uint32_t *)d_digest,
                               (uint32_t *)d_digest_copy);
        return (int)WF_OK;
    }));
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_coin_draw(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, void *d_coin, uint32_t count, void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_coin || !d_out) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    const uint32_t max_ext = field == WF_FIELD_F128 ? 2 : 3;       // 32 digest bytes hold two f128 or three 64-bit elements
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128 && field != WF_FIELD_F62) return WF_ERR_UNSUPPORTED;
    if (ext_degree < 1 || ext_degree > max_ext) return WF_ERR_UNSUPPORTED;
    if (count == 0) return WF_OK;
    wf_prof_begin(ctx, "coin");
    WF_TRY(with_hasher(hash, [&](auto h) { return launch_coin_draw<decltype(h)>(ctx, field, ext_degree, (CoinState *)d_coin, count, (uint64_t *)d_out); }));
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_coin_reseed_draw(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, void *d_coin, const void *d_digest, void *d_digest_copy,
                                  void *d_out) {
    WF_ENTER(ctx);
    if (!ctx || !d_coin || !d_digest || !d_out) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128 && field != WF_FIELD_F62) return WF_ERR_UNSUPPORTED;
    if (ext_degree < 1 || ext_degree > (field == WF_FIELD_F128 ? 2u : 3u)) return WF_ERR_UNSUPPORTED;
    wf_prof_begin(ctx, "coin");
    WF_TRY(with_hasher(hash, [&](auto h) {
        return launch_coin_reseed_draw<decltype(h)>(ctx, field, ext_degree, (CoinState *)d_coin, (const uint32_t *)d_digest, (uint32_t *)d_digest_copy,
                                                    (uint64_t *)d_out);
    }));
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_coin_read(wf_ctx *ctx, const void *d_coin, void *h_seed, uint64_t *h_counter) {
    WF_ENTER(ctx);
    if (!ctx || !d_coin || !h_seed || !h_counter) return WF_ERR_INVALID_ARG;
    CoinState st;
    WF_TRY(wf_copy_d2h(ctx, &st, d_coin, sizeof(st)));
    WF_TRY(wf_check_status(ctx));
    memcpy(h_seed, st.seed, 32);
    *h_counter = st.counter;
    return st.failed ? WF_ERR_NOT_FOUND : WF_OK;
}
