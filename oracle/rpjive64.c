/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * RpJive64_256 (crypto/src/hash/rescue/rp64_256_jive/mod.rs): Rescue-Prime over f64 with state width 8, 7 rounds, and
 * the Jive compression mode for 2-to-1 hashing.
 *   apply_permutation / apply_round   mod.rs:331-353   (sbox x^7, MDS, ARK1, inverse sbox, MDS, ARK2)
 *   apply_jive_summation              mod.rs:355-369
 *   hash_elements                     mod.rs:268-313   (capacity[0] = 1 iff len % 4 != 0; pad with 1, 0, ...)
 *   merge / merge_many / merge_with_int   mod.rs:186-263
 *   hash (bytes)                      mod.rs:68-180    (7-byte chunks, last chunk padded with a 1 byte)
 * The MDS product is the plain matrix product (the reference's frequency-domain mds_multiply equals it,
 * tests.rs:180-207).  Pinned against the permutation known-answer test of tests.rs:69-97.
 */
#include <stdint.h>
#include <string.h>
#include "f64.h"
#include "rpjive64_constants.h"

#define JW 8
#define INV_ALPHA 10540996611094048183ULL /* mod.rs:55 */

void or_rpjive_apply_permutation(uint64_t st[JW]) {
    for (int r = 0; r < 7; r++) {
        uint64_t t[JW];
        for (int i = 0; i < JW; i++) st[i] = f64_exp(st[i], 7);
        for (int i = 0; i < JW; i++) {
            uint64_t acc = f64_new(0);
            for (int j = 0; j < JW; j++) acc = f64_add(acc, f64_mul(f64_new(RPJ64_MDS[i][j]), st[j]));
            t[i] = acc;
        }
        for (int i = 0; i < JW; i++) st[i] = f64_add(t[i], f64_new(RPJ64_ARK1[r][i]));
        for (int i = 0; i < JW; i++) st[i] = f64_exp(st[i], INV_ALPHA);
        for (int i = 0; i < JW; i++) {
            uint64_t acc = f64_new(0);
            for (int j = 0; j < JW; j++) acc = f64_add(acc, f64_mul(f64_new(RPJ64_MDS[i][j]), st[j]));
            t[i] = acc;
        }
        for (int i = 0; i < JW; i++) st[i] = f64_add(t[i], f64_new(RPJ64_ARK2[r][i]));
    }
}

static void jive_sum(const uint64_t init[JW], const uint64_t fin[JW], uint64_t digest[4]) {
    for (int i = 0; i < 4; i++) digest[i] = f64_add(f64_add(init[i], init[4 + i]), f64_add(fin[i], fin[4 + i]));
}

void or_rpjive_hash_elements(const uint64_t *e, uint64_t n, uint64_t digest[4]) {
    uint64_t st[JW];
    for (int i = 0; i < JW; i++) st[i] = f64_new(0);
    if (n % 4 != 0) st[0] = f64_new(1);
    unsigned i = 0;
    for (uint64_t k = 0; k < n; k++) {
        st[4 + i] = f64_add(st[4 + i], e[k]);
        if (++i == 4) { or_rpjive_apply_permutation(st); i = 0; }
    }
    if (i > 0) {
        st[4 + i] = f64_new(1);
        for (i++; i < 4; i++) st[4 + i] = f64_new(0);
        or_rpjive_apply_permutation(st);
    }
    memcpy(digest, st + 4, 32);
}

void or_rpjive_merge(const uint64_t two[8], uint64_t digest[4]) {
    uint64_t st[JW];
    memcpy(st, two, 64);
    or_rpjive_apply_permutation(st);
    jive_sum(two, st, digest);
}

void or_rpjive_merge_with_int(const uint64_t seed[4], uint64_t value, uint64_t digest[4]) {
    uint64_t st[JW], init[JW];
    for (int i = 0; i < JW; i++) st[i] = f64_new(0);
    memcpy(st, seed, 32);
    st[4] = f64_new(value);
    if (value < F64_M) st[7] = f64_new(5);
    else { st[5] = f64_new(value / F64_M); st[7] = f64_new(6); }
    memcpy(init, st, 64);
    or_rpjive_apply_permutation(st);
    jive_sum(init, st, digest);
}

void or_rpjive_hash_bytes(const uint8_t *bytes, uint64_t len, uint64_t digest[4]) {
    uint64_t num = (len % 7 == 0) ? len / 7 : len / 7 + 1;
    uint64_t st[JW];
    for (int i = 0; i < JW; i++) st[i] = f64_new(0);
    if (num % 4 != 0) st[0] = f64_new(1);
    unsigned i = 0;
    for (uint64_t idx = 0; idx < num; idx++) {
        uint8_t buf[8] = {0};
        uint64_t clen = (idx < num - 1) ? 7 : len - 7 * idx;
        memcpy(buf, bytes + 7 * idx, clen);
        if (idx == num - 1) buf[clen] = 1;
        uint64_t v;
        memcpy(&v, buf, 8);
        st[4 + i] = f64_add(st[4 + i], f64_new(v));
        if (++i == 4) { or_rpjive_apply_permutation(st); i = 0; }
    }
    if (i > 0) {
        st[4 + i] = f64_new(1);
        for (i++; i < 4; i++) st[4 + i] = f64_new(0);
        or_rpjive_apply_permutation(st);
    }
    memcpy(digest, st + 4, 32);
}
