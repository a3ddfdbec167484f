/* ORACLE — TEST INFRASTRUCTURE ONLY.  f62 instantiation of field_tmpl.inc (math/src/field/f62/mod.rs).
 * Results are normalised to [0, M) on every operation (the reference's lazy [0, 2M) words are only ever observed
 * through normalize(), see f62.h). */
#include <stdlib.h>
#include "f62.h"
void or_blake3_hash(const uint8_t *in, uint64_t len, uint8_t out[32]);
void or_bytes_hash(int hasher, const uint8_t *in, uint64_t len, uint8_t out[32]);

#define FE uint64_t
#define FN(name) or_f62_##name
static inline uint64_t f62n_add(uint64_t a, uint64_t b) { return f62_normalize(f62_add(a, b)); }
static inline uint64_t f62n_sub(uint64_t a, uint64_t b) { return f62_normalize(f62_sub(a, b)); }
static inline uint64_t f62n_mul(uint64_t a, uint64_t b) { return f62_normalize(f62_mul(a, b)); }
static inline uint64_t f62n_inv(uint64_t a) { return f62_normalize(f62_inv(a)); }
static inline uint64_t f62n_exp(uint64_t a, uint64_t e) { return f62_normalize(f62_exp(a, e)); }
static inline uint64_t f62n_new(uint64_t v) { return f62_normalize(f62_new(v)); }
static inline uint64_t f62n_root(unsigned n) { return f62_normalize(f62_root_of_unity(n)); }
#define F_ADD f62n_add
#define F_SUB f62n_sub
#define F_MUL f62n_mul
#define F_INV f62n_inv
#define F_EXP(b, e) f62n_exp((b), (uint64_t)(e))
#define F_FROM_U64(v) f62n_new((uint64_t)(v))
#define F_ROOT f62n_root
static inline void f62_extD_mul(unsigned D, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    if (D == 1) out[0] = f62_mul(a[0], b[0]);
    else if (D == 2) f62_ext2_mul(a, b, out);
    else f62_ext3_mul(a, b, out);
    for (unsigned d = 0; d < D; d++) out[d] = f62_normalize(out[d]);
}
#define F_EXT_MUL f62_extD_mul
/* Blake3_256<f62>::hash_elements: not IS_CANONICAL => canonical little-endian bytes of as_int() (blake/mod.rs:58-64) */
void or_rp62_hash_elements(const uint64_t *e, uint64_t n, uint64_t digest[4]);
static inline void f62_hash_elems(int hasher, const uint64_t *e, uint64_t n, uint8_t *digest) {
    if (hasher == 4) { or_rp62_hash_elements(e, n, (uint64_t *)digest); return; }   /* Rp62_248 */
    uint64_t stackbuf[768];
    uint64_t *buf = n <= 768 ? stackbuf : (uint64_t *)malloc(n * 8);
    for (uint64_t i = 0; i < n; i++) buf[i] = f62_as_int(e[i]);
    or_bytes_hash(hasher, (const uint8_t *)buf, n * 8, digest);
    if (buf != stackbuf) free(buf);
}
#define F_HASH_ELEMS f62_hash_elems
#include "field_tmpl.inc"
#define F_ONE f62n_new(1)
#include "constraints_tmpl.inc"
uint64_t or_f62_new1(uint64_t v) { return f62n_new(v); }
uint64_t or_f62_as_int1(uint64_t v) { return f62_as_int(v); }
uint64_t or_f62_lazy_mul1(uint64_t a, uint64_t b) { return f62_mul(a, b); }
