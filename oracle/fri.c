/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see f64.h header note).
 *
 * CPU restatement of the FRI commit phase for the f64 field and its extensions:
 *   utils/core/src/lib.rs:166-183        transpose_slice
 *   fri/src/prover/mod.rs:179-239         build_layers / build_layer / set_remainder
 *   fri/src/prover/mod.rs:321-336         build_layer_commitment
 *   fri/src/folding/mod.rs:86-118,181-188 apply_drp / get_inv_offsets
 *   math/src/polynom/mod.rs:55-61         eval (Horner)
 *   fri/src/options.rs:85-93              num_fri_layers
 *   crypto/src/random/default.rs:82-185   DefaultRandomCoin (new / reseed / next / draw)
 *   fri/src/prover/channel.rs:117-127     DefaultProverChannel::{commit_fri_layer, draw_fri_alpha}
 * An element is D consecutive base-field words (D = 1, 2, 3).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "f64.h"

void or_hash_elements(int hasher, const uint64_t *elems, uint64_t n, uint8_t digest[32]);
void or_hash_merge(int hasher, const uint8_t two[64], uint8_t digest[32]);
void or_hash_merge_with_int(int hasher, const uint8_t seed[32], uint64_t value, uint8_t digest[32]);
int or_merkle_build(int hasher, const uint8_t *leaves, uint64_t n_leaves, uint8_t *nodes);
int or_merkle_build_par(int hasher, const uint8_t *leaves, uint64_t n_leaves, uint8_t *nodes);
void or_f64_get_inv_twiddles(uint64_t *out, uint64_t n);
void or_f64_fft_in_place(uint64_t *values, uint64_t n, unsigned D, const uint64_t *twiddles);
void or_f64_permute(uint64_t *v, uint64_t n, unsigned D);
void or_f64_interpolate_poly_with_offset(uint64_t *ev, uint64_t n, unsigned D, const uint64_t *inv_twiddles,
                                         uint64_t domain_offset);
void or_rp64_digest_as_bytes(const uint64_t digest[4], uint8_t out[32]);
void or_rp62_digest_as_bytes(const uint64_t digest[4], uint8_t out[32]);

/* transpose_slice::<E, N> — utils/core/src/lib.rs:166-183: result[i][j] = source[i + j * row_count] */
void or_transpose_slice(const uint64_t *src, uint64_t len, unsigned D, uint64_t N, uint64_t *dst) {
    uint64_t rc = len / N;
    for (uint64_t i = 0; i < rc; i++)
        for (uint64_t j = 0; j < N; j++)
            memcpy(dst + (i * N + j) * D, src + (i + j * rc) * D, D * 8);
}

/* build_layer_commitment — fri/src/prover/mod.rs:321-336: leaf[i] = hash_elements(values[i]); tree */
int or_fri_layer_commit(int hasher, const uint64_t *transposed, uint64_t rows, unsigned D, uint64_t N, uint8_t *leaves,
                        uint8_t *nodes) {
    for (uint64_t i = 0; i < rows; i++) or_hash_elements(hasher, transposed + i * N * D, N * D, leaves + 32 * i);
    return or_merkle_build(hasher, leaves, rows, nodes);
}

/* apply_drp — fri/src/folding/mod.rs:86-118, for `rows` consecutive rows starting at row_start of a layer whose full
 * domain has domain_size points (a shard of the layer; row_start = 0, rows = domain_size / N is the reference call).
 * values: rows x N elements; result: rows elements. */
void or_apply_drp_rows(const uint64_t *values, uint64_t rows, unsigned D, uint64_t N, uint64_t domain_size, uint64_t row_start,
                       uint64_t domain_offset, const uint64_t *alpha, uint64_t *result) {
    /* get_inv_offsets — folding/mod.rs:181-188 */
    uint64_t n = domain_size;
    uint64_t g_inv = f64_inv(f64_root_of_unity((unsigned)__builtin_ctzll(n)));
    uint64_t *inv_tw = (uint64_t *)malloc((N / 2 ? N / 2 : 1) * 8);
    or_f64_get_inv_twiddles(inv_tw, N);
    uint64_t len_offset = f64_inv(f64_new((uint32_t)N));
    uint64_t io = f64_mul(f64_inv(domain_offset), f64_exp(g_inv, row_start)); /* inv_offsets[i] = offset^-1 * g_inv^i */
    uint64_t poly[16 * 3];
    for (uint64_t i = 0; i < rows; i++) {
        memcpy(poly, values + i * N * D, N * D * 8);
        or_f64_fft_in_place(poly, N, D, inv_tw); /* serial_fft — math/src/fft/mod.rs:405-429 */
        or_f64_permute(poly, N, D);
        uint64_t offset = len_offset;
        for (uint64_t k = 0; k < N; k++) {
            for (unsigned d = 0; d < D; d++) poly[k * D + d] = f64_mul(poly[k * D + d], offset);
            offset = f64_mul(offset, io);
        }
        /* polynom::eval — Horner at alpha in the extension field */
        uint64_t acc[3] = {f64_new(0), f64_new(0), f64_new(0)}, t[3];
        for (uint64_t k = N; k-- > 0;) {
            f64_extD_mul(D, acc, alpha, t);
            for (unsigned d = 0; d < D; d++) acc[d] = f64_add(t[d], poly[k * D + d]);
        }
        memcpy(result + i * D, acc, D * 8);
        io = f64_mul(io, g_inv);
    }
    free(inv_tw);
}

void or_apply_drp(const uint64_t *values, uint64_t rows, unsigned D, uint64_t N, uint64_t domain_offset,
                  const uint64_t *alpha, uint64_t *result) {
    or_apply_drp_rows(values, rows, D, N, rows * N, 0, domain_offset, alpha, result);
}

/* num_fri_layers — fri/src/options.rs:85-93 */
uint64_t or_fri_num_layers(uint64_t domain_size, uint64_t folding, uint64_t blowup, uint64_t remainder_max_degree) {
    uint64_t result = 0, max_rem = (remainder_max_degree + 1) * blowup;
    while (domain_size > max_rem) {
        domain_size /= folding;
        result++;
    }
    return result;
}

/* set_remainder — fri/src/prover/mod.rs:230-239: interpolate over the coset, keep len/blowup coefficients, reversed */
void or_fri_remainder(int hasher, uint64_t *evals, uint64_t len, unsigned D, uint64_t domain_offset, uint64_t blowup,
                      uint64_t *remainder_poly, uint8_t commitment[32]) {
    uint64_t *inv_tw = (uint64_t *)malloc((len / 2 ? len / 2 : 1) * 8);
    or_f64_get_inv_twiddles(inv_tw, len);
    or_f64_interpolate_poly_with_offset(evals, len, D, inv_tw, domain_offset);
    uint64_t sz = len / blowup;
    for (uint64_t i = 0; i < sz; i++) memcpy(remainder_poly + i * D, evals + (sz - 1 - i) * D, D * 8);
    or_hash_elements(hasher, remainder_poly, sz * D, commitment);
    free(inv_tw);
}

/* ---- DefaultRandomCoin — crypto/src/random/default.rs ---------------------------------------------------- */
typedef struct {
    uint8_t seed[32];
    uint64_t counter;
    int hasher;
} or_coin;

void or_coin_new(or_coin *c, int hasher, const uint64_t *seed_elems, uint64_t n) { /* :114-117 */
    c->hasher = hasher;
    or_hash_elements(hasher, seed_elems, n, c->seed);
    c->counter = 0;
}
void or_coin_reseed(or_coin *c, const uint8_t data[32]) { /* :150-153 */
    uint8_t two[64];
    memcpy(two, c->seed, 32);
    memcpy(two + 32, data, 32);
    or_hash_merge(c->hasher, two, c->seed);
    c->counter = 0;
}
/* draw::<E> — :185-199: first ELEMENT_BYTES of next().as_bytes() must decode to canonical values < M */
int or_coin_draw(or_coin *c, unsigned D, uint64_t *out) {
    for (int iter = 0; iter < 1000; iter++) {
        uint8_t d[32], bytes[32];
        c->counter += 1;
        or_hash_merge_with_int(c->hasher, c->seed, c->counter, d);
        if (c->hasher == 1 || c->hasher == 3) or_rp64_digest_as_bytes((const uint64_t *)d, bytes); /* Digest::as_bytes */
        else memcpy(bytes, d, 32);
        int ok = 1;
        for (unsigned k = 0; k < D; k++) {
            uint64_t v;
            memcpy(&v, bytes + 8 * k, 8);
            if (v >= F64_M) ok = 0;
            else out[k] = f64_new(v);
        }
        if (ok) return 0;
    }
    return 1;
}
uint64_t or_coin_sizeof(void) { return sizeof(or_coin); }

/* first 8 bytes of Digest::as_bytes(), little-endian */
static uint64_t digest_head(int hasher, const uint8_t d[32]) {
    uint8_t bytes[32];
    uint64_t v;
    if (hasher == 1 || hasher == 3) or_rp64_digest_as_bytes((const uint64_t *)d, bytes);   /* both ElementDigest */
    else if (hasher == 4) or_rp62_digest_as_bytes((const uint64_t *)d, bytes);
    else memcpy(bytes, d, 32);
    memcpy(&v, bytes, 8);
    return v;
}
/* check_leading_zeros — random/default.rs:141-146: trailing_zeros of the little-endian head of
 * merge_with_int(seed, value) */
uint32_t or_coin_check_leading_zeros(const or_coin *c, uint64_t value) {
    uint8_t d[32];
    or_hash_merge_with_int(c->hasher, c->seed, value, d);
    uint64_t h = digest_head(c->hasher, d);
    return h ? (uint32_t)__builtin_ctzll(h) : 64u;
}
/* grind_query_seed, serial path — prover/src/channel.rs:169-185: first nonce >= 1 that passes; 0 = none below limit */
uint64_t or_coin_grind(const or_coin *c, uint32_t grinding_factor, uint64_t limit) {
    for (uint64_t nonce = 1; nonce < limit; nonce++)
        if (or_coin_check_leading_zeros(c, nonce) >= grinding_factor) return nonce;
    return 0;
}
/* draw_integers — random/default.rs:209-248: reseed with the nonce, then draw masked 8-byte heads until num_values
 * were collected (at most 1000 draws); returns the number written (== num_values on success) */
uint64_t or_coin_draw_integers(or_coin *c, uint64_t num_values, uint64_t domain_size, uint64_t nonce, uint64_t *out) {
    uint8_t d[32];
    or_hash_merge_with_int(c->hasher, c->seed, nonce, d);
    memcpy(c->seed, d, 32);
    c->counter = 0;
    const uint64_t mask = domain_size - 1;
    uint64_t n = 0;
    for (int iter = 0; iter < 1000 && n < num_values; iter++) {
        c->counter += 1;
        or_hash_merge_with_int(c->hasher, c->seed, c->counter, d);
        out[n++] = digest_head(c->hasher, d) & mask;
    }
    return n;
}
void or_coin_seed(const or_coin *c, uint8_t out[32]) { memcpy(out, c->seed, 32); }


/* FriProver::build_layers with the reference's `concurrent` feature (fri/src/prover/mod.rs:179-239: transpose_slice and
 * build_layer_commitment batch their rows over the thread pool, utils/core/src/lib.rs:185-203; apply_drp maps rows in parallel,
 * fri/src/folding/mod.rs:101-117; the Merkle tree builds one subtree per thread) against a DefaultProverChannel: the CPU side of
 * the FRI number in bench.py.  evals (len * D words) is consumed.  roots: one per layer + the remainder commitment; returns the
 * number of layers.  Same values as the serial functions above (tests/test_oracle_f64.py). */
uint64_t or_fri_build_layers_par(int hasher, uint64_t *evals, uint64_t len, unsigned D, uint64_t N, uint64_t blowup, uint64_t remainder_max_degree,
                                 uint64_t domain_offset, uint8_t *roots, uint64_t *alphas) {
    or_coin coin;
    or_coin_new(&coin, hasher, NULL, 0);
    const uint64_t nl = or_fri_num_layers(len, N, blowup, remainder_max_degree);
    uint64_t *cur = evals, *tr = (uint64_t *)malloc(len * D * 8), *folded = (uint64_t *)malloc((len / N) * D * 8);
    uint8_t *leaves = (uint8_t *)malloc((len / N) * 32), *nodes = (uint8_t *)malloc((len / N) * 32);
    uint64_t *inv_tw = (uint64_t *)malloc((N / 2 ? N / 2 : 1) * 8);
    or_f64_get_inv_twiddles(inv_tw, N);
    const uint64_t len_offset = f64_inv(f64_new((uint32_t)N)), off_inv = f64_inv(domain_offset);
    for (uint64_t k = 0; k < nl; k++) {
        const uint64_t rc = len / N;
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < rc; i++) {
            for (uint64_t j = 0; j < N; j++) memcpy(tr + (i * N + j) * D, cur + (i + j * rc) * D, D * 8);
            or_hash_elements(hasher, tr + i * N * D, N * D, leaves + 32 * i);
        }
        or_merkle_build_par(hasher, leaves, rc, nodes);
        memcpy(roots + 32 * k, nodes + 32, 32);
        or_coin_reseed(&coin, nodes + 32);
        uint64_t alpha[3];
        if (or_coin_draw(&coin, D, alpha)) return ~0ull;
        memcpy(alphas + k * D, alpha, D * 8);
        const uint64_t g_inv = f64_inv(f64_root_of_unity((unsigned)__builtin_ctzll(len)));
#pragma omp parallel
        {
            uint64_t poly[16 * 3], io = 0;
            int have = 0;
#pragma omp for schedule(static)
            for (uint64_t i = 0; i < rc; i++) {
                if (!have) { io = f64_mul(off_inv, f64_exp(g_inv, i)); have = 1; }      /* a thread's rows are consecutive */
                memcpy(poly, tr + i * N * D, N * D * 8);
                or_f64_fft_in_place(poly, N, D, inv_tw);
                or_f64_permute(poly, N, D);
                uint64_t offset = len_offset;
                for (uint64_t q = 0; q < N; q++) {
                    for (unsigned d = 0; d < D; d++) poly[q * D + d] = f64_mul(poly[q * D + d], offset);
                    offset = f64_mul(offset, io);
                }
                uint64_t acc[3] = {f64_new(0), f64_new(0), f64_new(0)}, t[3];
                for (uint64_t q = N; q-- > 0;) {
                    f64_extD_mul(D, acc, alpha, t);
                    for (unsigned d = 0; d < D; d++) acc[d] = f64_add(t[d], poly[q * D + d]);
                }
                memcpy(folded + i * D, acc, D * 8);
                io = f64_mul(io, g_inv);
            }
        }
        memcpy(cur, folded, rc * D * 8);
        len = rc;
    }
    uint64_t *rem = (uint64_t *)malloc((len / blowup ? len / blowup : 1) * D * 8);
    or_fri_remainder(hasher, cur, len, D, domain_offset, blowup, rem, roots + 32 * nl);
    free(rem);
    free(inv_tw);
    free(nodes);
    free(leaves);
    free(folded);
    free(tr);
    return nl;
}
