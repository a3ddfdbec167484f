/* ORACLE — TEST INFRASTRUCTURE ONLY.  f64 instantiation of field_tmpl.inc, used to cross-check the template against
 * the hand-restated f64 functions of fft_f64.c / commit.c / fri.c. */
#include "f64.h"
void or_hash_elements(int hasher, const uint64_t *elems, uint64_t n, uint8_t digest[32]);
#define FE uint64_t
#define FN(name) or_f64t_##name
#define F_ADD f64_add
#define F_SUB f64_sub
#define F_MUL f64_mul
#define F_INV f64_inv
#define F_EXP(b, e) f64_exp((b), (uint64_t)(e))
#define F_FROM_U64(v) f64_new((uint64_t)(v))
#define F_ROOT f64_root_of_unity
#define F_EXT_MUL f64_extD_mul
#define F_HASH_ELEMS(h, e, n, d) or_hash_elements((h), (e), (n), (d))
#include "field_tmpl.inc"
#define F_ONE f64_new(1)
#include "constraints_tmpl.inc"
