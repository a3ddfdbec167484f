/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * Rp62_248 (crypto/src/hash/rescue/rp62_248/mod.rs): Rescue-Prime over the 62-bit field, state width 12 (rate = state[0..8],
 * capacity = state[8..12] with the element count in state[11]), alpha = 3, 7 rounds, digest = state[0..4].
 *   apply_permutation / apply_round   mod.rs:244-264   (cube, MDS, ARK1, inverse S-box, MDS, ARK2)
 *   apply_mds                         mod.rs:269-278   (full 12x12 matrix product)
 *   hash_elements                     mod.rs:208-239
 *   merge / merge_many / merge_with_int   mod.rs:156-201
 *   ElementDigest::as_bytes           digest.rs:37-51  (4 x 62 bits packed into 31 bytes)
 * Values are Montgomery words; this file keeps them normalised (see f62.h).  Pinned against the permutation known-answer
 * test of tests.rs:34-70.
 */
#include <stdint.h>
#include <string.h>
#include "f62.h"
#include "rp62_248_constants.h"

#define W62 12
#define RP62_INV_ALPHA 3074416663688030891ULL /* mod.rs:39 */

static inline uint64_t n_add(uint64_t a, uint64_t b) { return f62_normalize(f62_add(a, b)); }
static inline uint64_t n_mul(uint64_t a, uint64_t b) { return f62_normalize(f62_mul(a, b)); }

void or_rp62_apply_permutation(uint64_t st[W62]) {
    for (int r = 0; r < 7; r++) {
        uint64_t t[W62];
        for (int i = 0; i < W62; i++) st[i] = n_mul(n_mul(st[i], st[i]), st[i]);
        for (int i = 0; i < W62; i++) {
            uint64_t acc = 0;
            for (int j = 0; j < W62; j++) acc = n_add(acc, n_mul(f62_normalize(f62_new(RP62_MDS[i][j])), st[j]));
            t[i] = acc;
        }
        for (int i = 0; i < W62; i++) st[i] = n_add(t[i], f62_normalize(f62_new(RP62_ARK1[r][i])));
        for (int i = 0; i < W62; i++) st[i] = f62_normalize(f62_exp(st[i], RP62_INV_ALPHA));
        for (int i = 0; i < W62; i++) {
            uint64_t acc = 0;
            for (int j = 0; j < W62; j++) acc = n_add(acc, n_mul(f62_normalize(f62_new(RP62_MDS[i][j])), st[j]));
            t[i] = acc;
        }
        for (int i = 0; i < W62; i++) st[i] = n_add(t[i], f62_normalize(f62_new(RP62_ARK2[r][i])));
    }
}

void or_rp62_hash_elements(const uint64_t *e, uint64_t n, uint64_t digest[4]) {
    uint64_t st[W62];
    memset(st, 0, sizeof st);
    st[11] = f62_normalize(f62_new(n));
    unsigned i = 0;
    for (uint64_t k = 0; k < n; k++) {
        st[i] = n_add(st[i], f62_normalize(e[k]));
        if (++i == 8) { or_rp62_apply_permutation(st); i = 0; }
    }
    if (i > 0) or_rp62_apply_permutation(st);
    memcpy(digest, st, 32);
}

void or_rp62_merge(const uint64_t two[8], uint64_t digest[4]) {
    uint64_t st[W62];
    memset(st, 0, sizeof st);
    for (int i = 0; i < 8; i++) st[i] = f62_normalize(two[i]);
    st[11] = f62_normalize(f62_new(8));
    or_rp62_apply_permutation(st);
    memcpy(digest, st, 32);
}

void or_rp62_merge_with_int(const uint64_t seed[4], uint64_t value, uint64_t digest[4]) {
    uint64_t st[W62];
    memset(st, 0, sizeof st);
    for (int i = 0; i < 4; i++) st[i] = f62_normalize(seed[i]);
    st[4] = f62_normalize(f62_new(value));
    if (value < F62_M) st[11] = f62_normalize(f62_new(5));
    else { st[5] = f62_normalize(f62_new(value / F62_M)); st[11] = f62_normalize(f62_new(6)); }
    or_rp62_apply_permutation(st);
    memcpy(digest, st, 32);
}

/* ElementDigest::as_bytes — digest.rs:37-51 */
void or_rp62_digest_as_bytes(const uint64_t digest[4], uint8_t out[32]) {
    uint64_t v1 = f62_as_int(digest[0]), v2 = f62_as_int(digest[1]), v3 = f62_as_int(digest[2]), v4 = f62_as_int(digest[3]);
    uint64_t w[4] = {v1 | (v2 << 62), (v2 >> 2) | (v3 << 60), (v3 >> 4) | (v4 << 58), v4 >> 6};
    memcpy(out, w, 32);
}
