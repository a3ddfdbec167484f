// Keccak-f[1600] and the SHA3-256 sponge (FIPS 202), one hash per lane.  The reference's Sha3_256 hasher
// (crypto/src/hash/sha/mod.rs:21-66) delegates to the `sha3` crate; this is the published algorithm.
// State: 25 64-bit lanes in registers, lane index x + 5y; rate 136 bytes = 17 lanes; digest = lanes 0..3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace k3 {

__device__ __forceinline__ uint64_t rotl(uint64_t v, int r) { return r == 0 ? v : (v << r) | (v >> (64 - r)); }

__device__ const uint64_t RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

__host__ __device__ constexpr int rho(int i) {
    constexpr int R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    return R[i];
}

__device__ __forceinline__ void f1600(uint64_t (&a)[25]) {
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], d[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];            // theta
#pragma unroll
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
#pragma unroll
        for (int x = 0; x < 5; x++)                                                                          // rho + pi
#pragma unroll
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y] ^ d[x], rho(x + 5 * y));
#pragma unroll
        for (int y = 0; y < 5; y++)                                                                          // chi
#pragma unroll
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];                                                                                   // iota
    }
}

// SHA3-256 of n 64-bit little-endian words produced by w(i); digest = first four lanes
template <class W>
__device__ __forceinline__ void sha3_256_words(const W &w, uint32_t n, uint64_t (&digest)[4]) {
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    uint32_t base = 0;
    for (; base + 17 <= n; base += 17) {
#pragma unroll
        for (int k = 0; k < 17; k++) st[k] ^= w(base + k);
        f1600(st);
    }
    const uint32_t left = n - base;                // < 17 words in the final (padded) block
#pragma unroll
    for (int k = 0; k < 17; k++) {
        if ((uint32_t)k < left) st[k] ^= w(base + k);
        else if ((uint32_t)k == left) st[k] ^= 0x06ull;          // domain bits 01 + first pad bit
    }
    st[16] ^= 0x8000000000000000ull;                             // last pad bit of the 136-byte block
    f1600(st);
#pragma unroll
    for (int i = 0; i < 4; i++) digest[i] = st[i];
}

// the same sponge with a block source: fetch(blk, m) fills the 17 words of rate block `blk` (words past the message zero);
// n = message length in words.  Lets a wavefront load the blocks of 64 rows cooperatively (hash_kernels.hip, wide rows).
template <class FB>
__device__ __forceinline__ void sha3_256_blocks(const FB &fetch, uint32_t n, uint64_t (&digest)[4]) {
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    const uint32_t full = n / 17, left = n - full * 17;
    for (uint32_t blk = 0; blk <= full; blk++) {
        uint64_t m[17];
        fetch(blk, m);
#pragma unroll
        for (int k = 0; k < 17; k++) st[k] ^= m[k];
        if (blk == full) {
#pragma unroll
            for (int k = 0; k < 17; k++)
                if ((uint32_t)k == left) st[k] ^= 0x06ull;
            st[16] ^= 0x8000000000000000ull;
        }
        f1600(st);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) digest[i] = st[i];
}

// SHA3-256 of an nbytes-long byte string delivered as 64-bit little-endian words by fetch(blk, m) (17 words per rate
// block, bytes past the message zero): the domain/pad byte 0x06 lands at byte nbytes of the message
template <class FB>
__device__ __forceinline__ void sha3_256_bytes(const FB &fetch, uint64_t nbytes, uint64_t (&digest)[4]) {
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    const uint32_t pad_word = (uint32_t)(nbytes / 8), pad_shift = (uint32_t)(nbytes % 8) * 8;
    const uint32_t last = pad_word / 17, left = pad_word - last * 17;
    for (uint32_t blk = 0; blk <= last; blk++) {
        uint64_t m[17];
        fetch(blk, m);
#pragma unroll
        for (int k = 0; k < 17; k++) st[k] ^= m[k];
        if (blk == last) {
#pragma unroll
            for (int k = 0; k < 17; k++)
                if ((uint32_t)k == left) st[k] ^= 0x06ull << pad_shift;
            st[16] ^= 0x8000000000000000ull;
        }
        f1600(st);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) digest[i] = st[i];
}

}  // namespace k3
