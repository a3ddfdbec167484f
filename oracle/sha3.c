/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * SHA3-256 (FIPS 202), restated from the published algorithm: the reference's Sha3_256 hasher
 * (crypto/src/hash/sha/mod.rs:21-66) delegates to the third-party crate `sha3` (crypto/Cargo.toml: sha3 = "0.10",
 * not vendored under /root/reference).  Pinned in tests against Python's hashlib.sha3_256 (an independent
 * implementation of the same standard) and against the reference's call-site structure (merge = hash of the 64
 * concatenated digest bytes, merge_with_int = hash of seed || value.to_le_bytes(), hash_elements = hash of the
 * elements' canonical little-endian bytes).
 */
#include <stdint.h>
#include <string.h>

static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
/* rotation offsets r[x][y] and the pi permutation, FIPS 202 section 3.2.2-3.2.3, lane index = x + 5 y */
static const unsigned KECCAK_RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

static inline uint64_t rotl64(uint64_t v, unsigned r) { return r ? (v << r) | (v >> (64 - r)) : v; }

void or_keccak_f1600(uint64_t a[25]) {
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];            /* theta */
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)                                                                          /* rho + pi */
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y], KECCAK_RHO[x + 5 * y]);
        for (int y = 0; y < 5; y++)                                                                          /* chi */
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= KECCAK_RC[round];                                                                            /* iota */
    }
}

/* SHA3-256: rate 136 bytes, domain-separation / padding byte 0x06 ... 0x80 */
void or_sha3_256(const uint8_t *in, uint64_t len, uint8_t out[32]) {
    uint64_t st[25];
    uint8_t block[136];
    memset(st, 0, sizeof st);
    while (len >= 136) {
        for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, in + 8 * i, 8); st[i] ^= w; }
        or_keccak_f1600(st);
        in += 136;
        len -= 136;
    }
    memset(block, 0, 136);
    memcpy(block, in, len);
    block[len] ^= 0x06;
    block[135] ^= 0x80;
    for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, block + 8 * i, 8); st[i] ^= w; }
    or_keccak_f1600(st);
    memcpy(out, st, 32);
}
