/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see f64.h header note).
 *
 * CPU restatement of the reference's math::fft for the f64 field and its extensions.
 * An "element" is D consecutive base-field words (D = 1, 2, 3: base / quadratic / cubic extension,
 * #[repr(C)] AoS — math/src/field/extensions/quadratic.rs:30-33, cubic.rs:30-33); multiplication by
 * a twiddle is `mul_base`, i.e. component-wise (f64/mod.rs:425-429, 484-488).
 *
 * Follows:
 *   math/src/fft/mod.rs:455-505  get_twiddles / get_inv_twiddles
 *   math/src/fft/mod.rs:570-578  permute_index
 *   math/src/fft/fft_inputs.rs:101-144,215-252  butterflies + fft_in_place
 *   math/src/fft/serial.rs:18-101  evaluate/interpolate (with offset)
 *   math/src/fft/concurrent.rs:18-236  4-step "split radix" variant (used for the timed CPU baseline)
 *   math/src/utils/mod.rs:36-79   get_power_series(_with_offset)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "f64.h"

#define MAX_LOOP 256 /* fft_inputs.rs:10 */

/* ---------------------------------------------------------------------------------------------- */
/* permute_index — mod.rs:570-578 */
uint64_t or_permute_index(uint64_t size, uint64_t index) {
    unsigned bits = (unsigned)__builtin_ctzll(size);
    if (bits == 0) return 0;
    uint64_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((index >> i) & 1ULL) << (bits - 1 - i);
    return r;
}

/* permute — serial fft_inputs.rs:74-84 : swap(i, bitrev(i)) for bitrev(i) > i */
void or_f64_permute(uint64_t *v, uint64_t n, unsigned D) {
    for (uint64_t i = 0; i < n; i++) {
        uint64_t j = or_permute_index(n, i);
        if (j > i) {
            for (unsigned d = 0; d < D; d++) {
                uint64_t t = v[i * D + d];
                v[i * D + d] = v[j * D + d];
                v[j * D + d] = t;
            }
        }
    }
}

/* get_power_series — utils/mod.rs:36-46 */
void or_f64_power_series(uint64_t b, uint64_t *out, uint64_t n) {
    uint64_t cur = f64_new(1);
    for (uint64_t i = 0; i < n; i++) {
        out[i] = cur;
        cur = f64_mul(cur, b);
    }
}

/* get_twiddles — mod.rs:455-468.  out has n/2 entries. */
void or_f64_get_twiddles(uint64_t *out, uint64_t n) {
    unsigned logn = (unsigned)__builtin_ctzll(n);
    uint64_t root = f64_root_of_unity(logn);
    or_f64_power_series(root, out, n / 2);
    or_f64_permute(out, n / 2, 1);
}

/* get_inv_twiddles — mod.rs:491-505 */
void or_f64_get_inv_twiddles(uint64_t *out, uint64_t n) {
    unsigned logn = (unsigned)__builtin_ctzll(n);
    uint64_t root = f64_root_of_unity(logn);
    uint64_t inv_root = f64_exp(root, n - 1);
    or_f64_power_series(inv_root, out, n / 2);
    or_f64_permute(out, n / 2, 1);
}

/* butterfly — fft_inputs.rs:106-113 */
static inline void butterfly(uint64_t *v, unsigned D, uint64_t offset, uint64_t stride) {
    uint64_t *pi = v + offset * D, *pj = v + (offset + stride) * D;
    for (unsigned d = 0; d < D; d++) {
        uint64_t temp = pi[d];
        pi[d] = f64_add(temp, pj[d]);
        pj[d] = f64_sub(temp, pj[d]);
    }
}

/* butterfly_twiddle — fft_inputs.rs:115-123 */
static inline void butterfly_twiddle(uint64_t *v, unsigned D, uint64_t tw, uint64_t offset, uint64_t stride) {
    uint64_t *pi = v + offset * D, *pj = v + (offset + stride) * D;
    for (unsigned d = 0; d < D; d++) {
        uint64_t temp = pi[d];
        uint64_t t = f64_mul(pj[d], tw);
        pi[d] = f64_add(temp, t);
        pj[d] = f64_sub(temp, t);
    }
}

/* fft_in_place — fft_inputs.rs:215-252 (natural-order input, bit-reversed output) */
static void fft_in_place_raw(uint64_t *values, uint64_t len, unsigned D, const uint64_t *twiddles,
                             uint64_t count, uint64_t stride, uint64_t offset) {
    uint64_t size = len / stride;
    if (size > 2) {
        if (stride == count && count < MAX_LOOP) {
            fft_in_place_raw(values, len, D, twiddles, 2 * count, 2 * stride, offset);
        } else {
            fft_in_place_raw(values, len, D, twiddles, count, 2 * stride, offset);
            fft_in_place_raw(values, len, D, twiddles, count, 2 * stride, offset + stride);
        }
    }
    for (uint64_t o = offset; o < offset + count; o++) butterfly(values, D, o, stride);
    uint64_t last_offset = offset + size * stride;
    uint64_t i = 0;
    for (uint64_t o = offset; o < last_offset; o += 2 * stride, i++) {
        if (i == 0) continue; /* .skip(1) */
        for (uint64_t j = o; j < o + count; j++) butterfly_twiddle(values, D, twiddles[i], j, stride);
    }
}

/* FftInputs::fft_in_place — fft_inputs.rs:33-35 */
void or_f64_fft_in_place(uint64_t *values, uint64_t n, unsigned D, const uint64_t *twiddles) {
    if (n < 2) return;
    fft_in_place_raw(values, n, D, twiddles, 1, 1, 0);
}

/* ---------------------------------------------------------------------------------------------- */
/* serial.rs                                                                                      */

/* evaluate_poly — serial.rs:18-25 */
void or_f64_evaluate_poly(uint64_t *p, uint64_t n, unsigned D, const uint64_t *twiddles) {
    or_f64_fft_in_place(p, n, D, twiddles);
    or_f64_permute(p, n, D);
}

/* evaluate_poly_with_offset — serial.rs:29-56.  result has n*blowup elements. */
void or_f64_evaluate_poly_with_offset(const uint64_t *p, uint64_t n, unsigned D, const uint64_t *twiddles,
                                      uint64_t domain_offset, uint64_t blowup, uint64_t *result) {
    uint64_t domain_size = n * blowup;
    uint64_t g = f64_root_of_unity((unsigned)__builtin_ctzll(domain_size));
    for (uint64_t i = 0; i < blowup; i++) {
        uint64_t *chunk = result + i * n * D;
        uint64_t idx = or_permute_index(blowup, i);
        uint64_t offset = f64_mul(f64_exp(g, idx), domain_offset);
        uint64_t factor = f64_new(1);
        for (uint64_t j = 0; j < n; j++) {
            for (unsigned d = 0; d < D; d++) chunk[j * D + d] = f64_mul(p[j * D + d], factor);
            factor = f64_mul(factor, offset);
        }
        or_f64_fft_in_place(chunk, n, D, twiddles);
    }
    or_f64_permute(result, domain_size, D);
}

/* interpolate_poly — serial.rs:66-76 */
void or_f64_interpolate_poly(uint64_t *ev, uint64_t n, unsigned D, const uint64_t *inv_twiddles) {
    uint64_t inv_length = f64_inv(f64_new((uint32_t)n));
    or_f64_fft_in_place(ev, n, D, inv_twiddles);
    for (uint64_t i = 0; i < n * D; i++) ev[i] = f64_mul(ev[i], inv_length); /* shift_by */
    or_f64_permute(ev, n, D);
}

/* interpolate_poly_with_offset — serial.rs:84-101 */
void or_f64_interpolate_poly_with_offset(uint64_t *ev, uint64_t n, unsigned D, const uint64_t *inv_twiddles,
                                         uint64_t domain_offset) {
    or_f64_fft_in_place(ev, n, D, inv_twiddles);
    or_f64_permute(ev, n, D);
    uint64_t inc = f64_inv(domain_offset);
    uint64_t off = f64_inv(f64_new((uint32_t)n));
    for (uint64_t i = 0; i < n; i++) { /* shift_by_series(offset, increment) fft_inputs.rs:129-136 */
        for (unsigned d = 0; d < D; d++) ev[i * D + d] = f64_mul(ev[i * D + d], off);
        off = f64_mul(off, inc);
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* concurrent.rs — Rayon path restated with OpenMP; this is the timed CPU baseline.               */

static unsigned num_threads_pow2(void) {
#ifdef _OPENMP
    unsigned t = (unsigned)omp_get_max_threads();
#else
    unsigned t = 1;
#endif
    unsigned p = 1;
    while (p < t) p <<= 1; /* rayon::current_num_threads().next_power_of_two() */
    return p;
}

/* concurrent::permute — concurrent.rs:104-125 */
void or_f64_permute_par(uint64_t *v, uint64_t n, unsigned D) {
    uint64_t nb = num_threads_pow2();
    if (nb > n) nb = n;
    uint64_t bs = n / nb;
#pragma omp parallel for schedule(static)
    for (uint64_t b = 0; b < nb; b++) {
        for (uint64_t i = b * bs; i < (b + 1) * bs; i++) {
            uint64_t j = or_permute_index(n, i);
            if (j > i)
                for (unsigned d = 0; d < D; d++) {
                    uint64_t t = v[i * D + d];
                    v[i * D + d] = v[j * D + d];
                    v[j * D + d] = t;
                }
        }
    }
}

static inline void swap_el(uint64_t *m, unsigned D, uint64_t i, uint64_t j) {
    for (unsigned d = 0; d < D; d++) {
        uint64_t t = m[i * D + d];
        m[i * D + d] = m[j * D + d];
        m[j * D + d] = t;
    }
}

/* transpose_square_1 / _2 — concurrent.rs:185-218 (single threaded in the reference as well) */
static void transpose_square_stretch(uint64_t *m, unsigned D, uint64_t size, uint64_t stretch) {
    if (stretch == 1) {
        for (uint64_t row = 0; row < size; row += 2) {
            uint64_t i = row * size + row;
            swap_el(m, D, i + 1, i + size);
            for (uint64_t col = row + 2; col < size; col += 2) {
                uint64_t a = row * size + col, b = col * size + row;
                swap_el(m, D, a, b);
                swap_el(m, D, a + 1, b + size);
                swap_el(m, D, a + size, b + 1);
                swap_el(m, D, a + size + 1, b + size + 1);
            }
        }
    } else {
        for (uint64_t row = 0; row < size; row++)
            for (uint64_t col = row + 1; col < size; col++) {
                uint64_t a = (row * size + col) * 2, b = (col * size + row) * 2;
                swap_el(m, D, a, b);
                swap_el(m, D, a + 1, b + 1);
            }
    }
}

/* split_radix_fft — concurrent.rs:132-171 */
void or_f64_split_radix_fft(uint64_t *values, uint64_t n, unsigned D, const uint64_t *twiddles) {
    uint64_t g = twiddles[(n / 2) / 2];
    unsigned logn = (unsigned)__builtin_ctzll(n);
    uint64_t inner_len = 1ULL << (logn / 2);
    uint64_t outer_len = n / inner_len;
    uint64_t stretch = outer_len / inner_len;
    transpose_square_stretch(values, D, inner_len, stretch);
#pragma omp parallel for schedule(dynamic)
    for (uint64_t r = 0; r < n / outer_len; r++)
        fft_in_place_raw(values + r * outer_len * D, outer_len, D, twiddles, stretch, stretch, 0);
    transpose_square_stretch(values, D, inner_len, stretch);
#pragma omp parallel for schedule(dynamic)
    for (uint64_t r = 0; r < n / outer_len; r++) {
        uint64_t *row = values + r * outer_len * D;
        if (r > 0) {
            uint64_t i = or_permute_index(inner_len, r);
            uint64_t inner_tw = f64_exp(g, (uint32_t)i);
            uint64_t outer_tw = inner_tw;
            for (uint64_t k = 1; k < outer_len; k++) {
                for (unsigned d = 0; d < D; d++) row[k * D + d] = f64_mul(row[k * D + d], outer_tw);
                outer_tw = f64_mul(outer_tw, inner_tw);
            }
        }
        fft_in_place_raw(row, outer_len, D, twiddles, 1, 1, 0);
    }
}

/* math/src/fft/mod.rs:34,104-111: the concurrent versions are only dispatched for n >= MIN_CONCURRENT_SIZE */
#define MIN_CONCURRENT_SIZE 1024

/* concurrent::evaluate_poly — concurrent.rs:18-21 */
void or_f64_evaluate_poly_par(uint64_t *p, uint64_t n, unsigned D, const uint64_t *twiddles) {
    if (n < MIN_CONCURRENT_SIZE) { or_f64_evaluate_poly(p, n, D, twiddles); return; }
    or_f64_split_radix_fft(p, n, D, twiddles);
    or_f64_permute_par(p, n, D);
}

/* concurrent::interpolate_poly — concurrent.rs:59-70 */
void or_f64_interpolate_poly_par(uint64_t *v, uint64_t n, unsigned D, const uint64_t *inv_twiddles) {
    if (n < MIN_CONCURRENT_SIZE) { or_f64_interpolate_poly(v, n, D, inv_twiddles); return; }
    or_f64_split_radix_fft(v, n, D, inv_twiddles);
    uint64_t inv_length = f64_inv(f64_new((uint32_t)n));
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n * D; i++) v[i] = f64_mul(v[i], inv_length);
    or_f64_permute_par(v, n, D);
}

/* concurrent::evaluate_poly_with_offset — concurrent.rs:26-49 */
void or_f64_evaluate_poly_with_offset_par(const uint64_t *p, uint64_t n, unsigned D, const uint64_t *twiddles,
                                          uint64_t domain_offset, uint64_t blowup, uint64_t *result) {
    if (n < MIN_CONCURRENT_SIZE) { or_f64_evaluate_poly_with_offset(p, n, D, twiddles, domain_offset, blowup, result); return; }
    uint64_t domain_size = n * blowup;
    uint64_t g = f64_root_of_unity((unsigned)__builtin_ctzll(domain_size));
    for (uint64_t i = 0; i < blowup; i++) { /* par_chunks_mut: nested parallelism lives inside */
        uint64_t *chunk = result + i * n * D;
        uint64_t idx = or_permute_index(blowup, i);
        uint64_t offset = f64_mul(f64_exp(g, idx), domain_offset);
        /* clone_and_shift — concurrent.rs:223-236 */
        uint64_t nb = num_threads_pow2();
        if (nb > n) nb = n;
        uint64_t bs = n / nb;
#pragma omp parallel for schedule(static)
        for (uint64_t b = 0; b < nb; b++) {
            uint64_t factor = f64_exp(offset, b * bs);
            for (uint64_t j = b * bs; j < (b + 1) * bs; j++) {
                for (unsigned d = 0; d < D; d++) chunk[j * D + d] = f64_mul(p[j * D + d], factor);
                factor = f64_mul(factor, offset);
            }
        }
        or_f64_split_radix_fft(chunk, n, D, twiddles);
    }
    or_f64_permute_par(result, domain_size, D);
}

/* ---------------------------------------------------------------------------------------------- */
/* Definitional check helper: polynom::eval (Horner) — math/src/polynom/mod.rs:55-61, base field.  */
uint64_t or_f64_poly_eval(const uint64_t *p, uint64_t n, uint64_t x) {
    uint64_t acc = f64_new(0);
    for (uint64_t i = n; i-- > 0;) acc = f64_add(f64_mul(acc, x), p[i]);
    return acc;
}

/* Conversions used by tests to move between canonical integers and the internal form. */
void or_f64_from_int(const uint64_t *in, uint64_t *out, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) out[i] = f64_new(in[i]);
}
void or_f64_to_int(const uint64_t *in, uint64_t *out, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) out[i] = f64_as_int(in[i]);
}
uint64_t or_f64_mul1(uint64_t a, uint64_t b) { return f64_mul(a, b); }
uint64_t or_f64_add1(uint64_t a, uint64_t b) { return f64_add(a, b); }
uint64_t or_f64_sub1(uint64_t a, uint64_t b) { return f64_sub(a, b); }
uint64_t or_f64_inv1(uint64_t a) { return f64_inv(a); }
uint64_t or_f64_exp1(uint64_t a, uint64_t e) { return f64_exp(a, e); }
uint64_t or_f64_new1(uint64_t a) { return f64_new(a); }
uint64_t or_f64_as_int1(uint64_t a) { return f64_as_int(a); }
uint64_t or_f64_root_of_unity1(unsigned n) { return f64_root_of_unity(n); }
void or_f64_ext_mul(unsigned D, const uint64_t *a, const uint64_t *b, uint64_t *out) { f64_extD_mul(D, a, b, out); }
