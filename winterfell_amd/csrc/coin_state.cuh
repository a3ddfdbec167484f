// The device-resident DefaultRandomCoin state and the element decoding shared by the coin kernels (coin.hip) and by kernels that run
// coin steps in line (the FRI tail kernel of fri_rows.hip).
#pragma once
#include "hashers.cuh"

namespace {

struct CoinState {
    uint32_t seed[8];
    uint64_t counter;
    uint32_t failed;     // bit field (kernels OR their bit in): bit 0 = a draw ran out of its 1000 tries (default.rs:185-199:
                         // FailedToDrawFieldElement), bit 1 = grinding found no nonce in the searched range (prover/src/channel.rs:169-185)
    uint32_t pad;
};
static_assert(sizeof(CoinState) <= WF_COIN_BYTES, "WF_COIN_BYTES");

// E::from_random_bytes over the first ELEMENT_BYTES of as_bytes: every base element must already be canonical
template <int FIELD, int D>
__device__ __forceinline__ bool coin_element(const uint32_t (&b)[8], uint64_t *out) {
    if constexpr (FIELD == WF_FIELD_F128) {
        uint64_t w[2 * D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const f128::u128 v = f128::join(b[4 * d], b[4 * d + 1], b[4 * d + 2], b[4 * d + 3]);
            if (v >= f128::modulus()) return false;
            w[2 * d] = (uint64_t)v;
            w[2 * d + 1] = (uint64_t)(v >> 64);
        }
#pragma unroll
        for (int i = 0; i < 2 * D; i++) out[i] = w[i];
    } else {
        uint64_t w[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint64_t v = (uint64_t)b[2 * d] | ((uint64_t)b[2 * d + 1] << 32);
            if constexpr (FIELD == WF_FIELD_F64) {
                if (v >= gl::P) return false;
                w[d] = gl::mul(v, 0xfffffffe00000001ull);   // BaseElement::new: times R^2 = 2^128 mod p
            } else {
                if (v >= f62::M) return false;
                w[d] = rp62::to_mont(v);
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) out[d] = w[d];
    }
    return true;
}

// commit_fri_layer + draw_fri_alpha on ONE lane (default.rs:150-153,185-199): seed = merge(seed, digest), counter = 0, then next() =
// merge_with_int(seed, ++counter) until the bytes decode; the digest is copied to root_out when given
template <class H, int FIELD, int D>
__device__ __forceinline__ void coin_reseed_draw_lane(CoinState *c, const uint32_t *digest, uint32_t *root_out, uint64_t *out) {
    uint32_t m[16], seed[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        m[i] = c->seed[i];
        m[8 + i] = digest[i];
    }
    H::merge(m, seed);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c->seed[i] = seed[i];
        if (root_out) root_out[i] = digest[i];
    }
    uint64_t counter = 0;
    bool ok = false;
    for (int tries = 0; tries < 1000 && !ok; tries++) {
        uint32_t d[8], b[8];
        counter++;
        H::merge_with_int(seed, counter, d);
        H::as_bytes(d, b);
        ok = coin_element<FIELD, D>(b, out);
    }
    if (!ok) c->failed |= 1u;
    c->counter = counter;
}

// The same for Blake3_256 on FOUR lanes (b3::quad_hash_block), as a workgroup-wide device function: EVERY thread of the workgroup
// calls it (it synchronises with __syncthreads), lanes 0..3 do the work.  msg: 16 words, drawn: 8 words, ok_flag: one int, all in LDS.
// Used by the coin kernel itself (one wavefront), by the tree launch that finishes an FRI layer (merkle.hip) and by the FRI tail.
template <int FIELD, int D>
__device__ __forceinline__ void coin_reseed_draw_quad_wg(CoinState *c, const uint32_t *digest, uint32_t *root_out, uint64_t *out, uint32_t q, uint32_t *msg,
                                                         uint32_t *drawn, int *ok_flag_p) {
    volatile int *ok_flag_v = ok_flag_p;
#define ok_flag (*ok_flag_v)
    if (q < 4) {
        msg[q] = c->seed[q];
        msg[4 + q] = c->seed[4 + q];
        msg[8 + q] = digest[q];
        msg[12 + q] = digest[4 + q];
        if (root_out) {
            root_out[q] = digest[q];
            root_out[4 + q] = digest[4 + q];
        }
    }
    __syncthreads();
    uint32_t lo = 0, hi = 0;
    if (q < 4) {
        const b3::Quad k = b3::quad_init(q, 64, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);     // merge: 64 bytes
        b3::quad_hash_block(k, msg, lo, hi);
        c->seed[q] = lo;
        c->seed[4 + q] = hi;
    }
    const b3::Quad k40 = b3::quad_init(q & 3, 40, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT);   // merge_with_int: seed || u64
    uint64_t counter = 0;
    bool ok = false;
    for (int tries = 0; tries < 1000 && !ok; tries++) {
        counter++;
        __syncthreads();
        if (q < 4) {
            msg[q] = lo;
            msg[4 + q] = hi;
            msg[8 + q] = q == 0 ? (uint32_t)counter : q == 1 ? (uint32_t)(counter >> 32) : 0u;
            msg[12 + q] = 0;
        }
        __syncthreads();
        if (q < 4) {
            uint32_t dl, dh;
            b3::quad_hash_block(k40, msg, dl, dh);
            drawn[q] = dl;
            drawn[4 + q] = dh;
        }
        __syncthreads();
        if (q == 0) {
            uint32_t b[8];
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = drawn[i];
            ok_flag = coin_element<FIELD, D>(b, out) ? 1 : 0;
        }
        __syncthreads();
        ok = ok_flag != 0;
    }
    if (q == 0) {
        if (!ok) c->failed |= 1u;
        c->counter = counter;
    }
#undef ok_flag
}

}  // namespace
