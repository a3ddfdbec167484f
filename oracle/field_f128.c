/* ORACLE — TEST INFRASTRUCTURE ONLY.  f128 instantiation of field_tmpl.inc (math/src/field/f128/mod.rs). */
#include "f128.h"
void or_blake3_hash(const uint8_t *in, uint64_t len, uint8_t out[32]);

#define FE u128
#define FN(name) or_f128_##name
#define F_ADD f128_add
#define F_SUB f128_sub
#define F_MUL f128_mul
#define F_INV f128_inv
#define F_EXP(b, e) f128_exp((b), (u128)(e))
#define F_FROM_U64(v) ((u128)(v))
#define F_ROOT f128_root_of_unity
static inline void f128_extD_mul(unsigned D, const u128 *a, const u128 *b, u128 *out) {
    if (D == 1) out[0] = f128_mul(a[0], b[0]);
    else f128_ext2_mul(a, b, out);
}
#define F_EXT_MUL f128_extD_mul
/* Blake3_256<f128>::hash_elements: IS_CANONICAL => raw element bytes (crypto/src/hash/blake/mod.rs:53-57);
 * Rp64_256 is only defined over f64. */
static inline void f128_hash_elems(int hasher, const u128 *e, uint64_t n, uint8_t *digest) {
    (void)hasher;
    or_blake3_hash((const uint8_t *)e, n * 16, digest);
}
#define F_HASH_ELEMS f128_hash_elems
#include "field_tmpl.inc"
