/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see f64.h header note).
 *
 * Rescue-Prime Rp64_256 (width 12, rate 8, capacity 4, 7 rounds, alpha = 7) restated from
 *   crypto/src/hash/rescue/rp64_256/mod.rs:123-257 (sponge: hash, merge, merge_many, merge_with_int,
 *       hash_elements), :299-384 (permutation, S-box, inverse S-box addition chain)
 *   crypto/src/hash/rescue/mod.rs:20-28 (exp_acc)
 *   crypto/src/hash/mds/mds_f64_12x12.rs:41-158 (frequency-domain circulant MDS)
 *   math/src/fft/real_u64.rs:8-45 (tiny real FFTs)
 * Constant tables (MDS first row, ARK1, ARK2) are data generated into rp64_256_constants.h.
 * State words are the reference's internal Montgomery residues.  Pinned by the reference's
 * apply_permutation([0..11]) known-answer test (rp64_256/tests.rs:70-105).
 */
#include <stdint.h>
#include <string.h>
#include "f64.h"
#include "rp64_256_constants.h"

#define STATE_WIDTH 12
#define NUM_ROUNDS 7

static uint64_t ARK1_M[NUM_ROUNDS][STATE_WIDTH], ARK2_M[NUM_ROUNDS][STATE_WIDTH];
static int consts_ready = 0;
static void init_consts(void) {
    if (consts_ready) return;
    for (int r = 0; r < NUM_ROUNDS; r++)
        for (int i = 0; i < STATE_WIDTH; i++) {
            ARK1_M[r][i] = f64_new(RP64_ARK1[r][i]);
            ARK2_M[r][i] = f64_new(RP64_ARK2[r][i]);
        }
    consts_ready = 1;
}

/* ---- real_u64.rs ---- */
static inline void fft4_real(const uint64_t x[4], int64_t *y0, int64_t *y1r, int64_t *y1i, int64_t *y2) {
    int64_t z0 = (int64_t)x[0] + (int64_t)x[2], z2 = (int64_t)x[0] - (int64_t)x[2];
    int64_t z1 = (int64_t)x[1] + (int64_t)x[3], z3 = (int64_t)x[1] - (int64_t)x[3];
    *y0 = z0 + z1;
    *y1r = z2;
    *y1i = -z3;
    *y2 = z0 - z1;
}
static inline void ifft4_real_unreduced(int64_t y0, int64_t y1r, int64_t y1i, int64_t y2, uint64_t x[4]) {
    int64_t z0 = y0 + y2, z1 = y0 - y2, z2 = y1r, z3 = -y1i;
    x[0] = (uint64_t)(z0 + z2);
    x[2] = (uint64_t)(z0 - z2);
    x[1] = (uint64_t)(z1 + z3);
    x[3] = (uint64_t)(z1 - z3);
}

/* ---- mds_f64_12x12.rs:72-158 ---- */
static const int64_t B1[3] = {16, 8, 16};
static const int64_t B2[3][2] = {{-1, 2}, {-1, 1}, {4, 8}};
static const int64_t B3[3] = {-8, 1, 1};

static void mds_multiply_freq(uint64_t s[12]) {
    int64_t u0, u1r, u1i, u2, u4, u5r, u5i, u6, u8, u9r, u9i, u10;
    uint64_t a[4] = {s[0], s[3], s[6], s[9]}, b[4] = {s[1], s[4], s[7], s[10]}, c[4] = {s[2], s[5], s[8], s[11]};
    fft4_real(a, &u0, &u1r, &u1i, &u2);
    fft4_real(b, &u4, &u5r, &u5i, &u6);
    fft4_real(c, &u8, &u9r, &u9i, &u10);

    /* block1 */
    int64_t x0 = u0, x1 = u4, x2 = u8;
    int64_t v0 = x0 * B1[0] + x1 * B1[2] + x2 * B1[1];
    int64_t v4 = x0 * B1[1] + x1 * B1[0] + x2 * B1[2];
    int64_t v8 = x0 * B1[2] + x1 * B1[1] + x2 * B1[0];

    /* block2 */
    int64_t x0r = u1r, x0i = u1i, x1r = u5r, x1i = u5i, x2r = u9r, x2i = u9i;
    int64_t y0r = B2[0][0], y0i = B2[0][1], y1r = B2[1][0], y1i = B2[1][1], y2r = B2[2][0], y2i = B2[2][1];
    int64_t x0s = x0r + x0i, x1s = x1r + x1i, x2s = x2r + x2i;
    int64_t y0s = y0r + y0i, y1s = y1r + y1i, y2s = y2r + y2i;
    int64_t m0a, m0b, m1a, m1b, m2a, m2b;
    m0a = x0r * y0r; m0b = x0i * y0i; m1a = x1r * y2r; m1b = x1i * y2i; m2a = x2r * y1r; m2b = x2i * y1i;
    int64_t z0r = (m0a - m0b) + (x1s * y2s - m1a - m1b) + (x2s * y1s - m2a - m2b);
    int64_t z0i = (x0s * y0s - m0a - m0b) + (-m1a + m1b) + (-m2a + m2b);
    m0a = x0r * y1r; m0b = x0i * y1i; m1a = x1r * y0r; m1b = x1i * y0i; m2a = x2r * y2r; m2b = x2i * y2i;
    int64_t z1r = (m0a - m0b) + (m1a - m1b) + (x2s * y2s - m2a - m2b);
    int64_t z1i = (x0s * y1s - m0a - m0b) + (x1s * y0s - m1a - m1b) + (-m2a + m2b);
    m0a = x0r * y2r; m0b = x0i * y2i; m1a = x1r * y1r; m1b = x1i * y1i; m2a = x2r * y0r; m2b = x2i * y0i;
    int64_t z2r = (m0a - m0b) + (m1a - m1b) + (m2a - m2b);
    int64_t z2i = (x0s * y2s - m0a - m0b) + (x1s * y1s - m1a - m1b) + (x2s * y0s - m2a - m2b);

    /* block3 */
    x0 = u2; x1 = u6; x2 = u10;
    int64_t v2 = x0 * B3[0] - x1 * B3[2] - x2 * B3[1];
    int64_t v6 = x0 * B3[1] + x1 * B3[0] - x2 * B3[2];
    int64_t v10 = x0 * B3[2] + x1 * B3[1] + x2 * B3[0];

    ifft4_real_unreduced(v0, z0r, z0i, v2, a);
    ifft4_real_unreduced(v4, z1r, z1i, v6, b);
    ifft4_real_unreduced(v8, z2r, z2i, v10, c);
    s[0] = a[0]; s[3] = a[1]; s[6] = a[2]; s[9] = a[3];
    s[1] = b[0]; s[4] = b[1]; s[7] = b[2]; s[10] = b[3];
    s[2] = c[0]; s[5] = c[1]; s[8] = c[2]; s[11] = c[3];
}

/* mds_multiply — mds_f64_12x12.rs:41-68 */
static void mds_multiply(uint64_t state[12]) {
    uint64_t lo[12], hi[12];
    for (int r = 0; r < 12; r++) {
        hi[r] = state[r] >> 32;
        lo[r] = (uint64_t)(uint32_t)state[r];
    }
    mds_multiply_freq(hi);
    mds_multiply_freq(lo);
    for (int r = 0; r < 12; r++) {
        u128 s = (u128)lo[r] + ((u128)hi[r] << 32);
        uint64_t s_hi = (uint64_t)(s >> 64), s_lo = (uint64_t)s;
        uint64_t z = (s_hi << 32) - s_hi;
        uint64_t res = s_lo + z;
        uint32_t over = res < s_lo;
        state[r] = res + (uint64_t)(uint32_t)(0u - over);
    }
}

/* naive dense MDS (definition) — used by tests to check the frequency-domain version */
void or_rp64_mds_naive(uint64_t state[12]) {
    uint64_t out[12];
    for (int i = 0; i < 12; i++) {
        uint64_t acc = f64_new(0);
        for (int j = 0; j < 12; j++) acc = f64_add(acc, f64_mul(f64_new(RP64_MDS[i][j]), state[j]));
        out[i] = acc;
    }
    memcpy(state, out, sizeof out);
}
void or_rp64_mds_freq(uint64_t state[12]) { mds_multiply(state); }

static inline uint64_t exp7(uint64_t x) { /* f64/mod.rs:93-98 */
    uint64_t x2 = f64_square(x), x4 = f64_square(x2), x3 = f64_mul(x2, x);
    return f64_mul(x3, x4);
}

static void exp_acc(const uint64_t base[12], const uint64_t tail[12], int m, uint64_t out[12]) {
    for (int i = 0; i < 12; i++) {
        uint64_t r = base[i];
        for (int k = 0; k < m; k++) r = f64_square(r);
        out[i] = f64_mul(r, tail[i]);
    }
}

/* apply_inv_sbox — rp64_256/mod.rs:351-384 */
static void apply_inv_sbox(uint64_t s[12]) {
    uint64_t t1[12], t2[12], t3[12], t4[12], t5[12], t6[12], t7[12];
    for (int i = 0; i < 12; i++) t1[i] = f64_square(s[i]);
    for (int i = 0; i < 12; i++) t2[i] = f64_square(t1[i]);
    exp_acc(t2, t2, 3, t3);
    exp_acc(t3, t3, 6, t4);
    exp_acc(t4, t4, 12, t5);
    exp_acc(t5, t3, 6, t6);
    exp_acc(t6, t6, 31, t7);
    for (int i = 0; i < 12; i++) {
        uint64_t a = f64_square(f64_square(f64_mul(f64_square(t7[i]), t6[i])));
        uint64_t b = f64_mul(f64_mul(t1[i], t2[i]), s[i]);
        s[i] = f64_mul(a, b);
    }
}

/* apply_permutation — rp64_256/mod.rs:299-319 */
void or_rp64_apply_permutation(uint64_t s[12]) {
    init_consts();
    for (int r = 0; r < NUM_ROUNDS; r++) {
        for (int i = 0; i < 12; i++) s[i] = exp7(s[i]);
        mds_multiply(s);
        for (int i = 0; i < 12; i++) s[i] = f64_add(s[i], ARK1_M[r][i]);
        apply_inv_sbox(s);
        mds_multiply(s);
        for (int i = 0; i < 12; i++) s[i] = f64_add(s[i], ARK2_M[r][i]);
    }
}

/* hash_elements — rp64_256/mod.rs:224-257.  `elements`: n base-field words (extension elements are
 * flattened by slice_as_base_elements). */
void or_rp64_hash_elements(const uint64_t *elements, uint64_t n, uint64_t digest[4]) {
    uint64_t st[12];
    for (int i = 0; i < 12; i++) st[i] = f64_new(0);
    st[0] = f64_new(n);
    unsigned i = 0;
    for (uint64_t k = 0; k < n; k++) {
        st[4 + i] = f64_add(st[4 + i], elements[k]);
        i++;
        if (i % 8 == 0) {
            or_rp64_apply_permutation(st);
            i = 0;
        }
    }
    if (i > 0) or_rp64_apply_permutation(st);
    memcpy(digest, st + 4, 32);
}

/* merge — rp64_256/mod.rs:181-192 */
void or_rp64_merge(const uint64_t two[8], uint64_t digest[4]) {
    uint64_t st[12];
    for (int i = 0; i < 4; i++) st[i] = f64_new(0);
    memcpy(st + 4, two, 64);
    st[0] = f64_new(8);
    or_rp64_apply_permutation(st);
    memcpy(digest, st + 4, 32);
}

/* merge_with_int — rp64_256/mod.rs:198-219 */
void or_rp64_merge_with_int(const uint64_t seed[4], uint64_t value, uint64_t digest[4]) {
    uint64_t st[12];
    for (int i = 0; i < 12; i++) st[i] = f64_new(0);
    memcpy(st + 4, seed, 32);
    st[8] = f64_new(value);
    if (value < F64_M) {
        st[0] = f64_new(5);
    } else {
        st[9] = f64_new(value / F64_M);
        st[0] = f64_new(6);
    }
    or_rp64_apply_permutation(st);
    memcpy(digest, st + 4, 32);
}

/* Hasher::hash(bytes) — rp64_256/mod.rs:128-179 (7-byte chunks, final chunk padded with a 1 byte) */
void or_rp64_hash_bytes(const uint8_t *bytes, uint64_t len, uint64_t digest[4]) {
    uint64_t num_elements = (len % 7 == 0) ? len / 7 : len / 7 + 1;
    uint64_t st[12];
    for (int i = 0; i < 12; i++) st[i] = f64_new(0);
    st[0] = f64_new(num_elements);
    unsigned i = 0;
    uint8_t buf[8] = {0};
    uint64_t k = 0;
    for (uint64_t off = 0; off < len; off += 7, k++) {
        uint64_t clen = (len - off) < 7 ? (len - off) : 7;
        if (k < num_elements - 1) {
            memcpy(buf, bytes + off, 7);
        } else {
            memset(buf, 0, 8);
            memcpy(buf, bytes + off, clen);
            buf[clen] = 1;
        }
        uint64_t v;
        memcpy(&v, buf, 8);
        st[4 + i] = f64_add(st[4 + i], f64_new(v));
        i++;
        if (i % 8 == 0) {
            or_rp64_apply_permutation(st);
            i = 0;
        }
    }
    if (i > 0) or_rp64_apply_permutation(st);
    memcpy(digest, st + 4, 32);
}

/* ElementDigest::as_bytes — rp64_256/digest.rs:36-45 (canonical little-endian) */
void or_rp64_digest_as_bytes(const uint64_t digest[4], uint8_t out[32]) {
    for (int i = 0; i < 4; i++) {
        uint64_t v = f64_as_int(digest[i]);
        memcpy(out + 8 * i, &v, 8);
    }
}
