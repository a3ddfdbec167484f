/* ORACLE — TEST INFRASTRUCTURE ONLY.  f128 instantiation of field_tmpl.inc (math/src/field/f128/mod.rs). */
#include "f128.h"
void or_blake3_hash(const uint8_t *in, uint64_t len, uint8_t out[32]);
void or_bytes_hash(int hasher, const uint8_t *in, uint64_t len, uint8_t out[32]);

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
    or_bytes_hash(hasher, (const uint8_t *)e, n * 16, digest);
}
#define F_HASH_ELEMS f128_hash_elems
#include "field_tmpl.inc"

/* ---- Rescue example AIR over f128 (examples/src/rescue) -------------------------------------------------------- */
#define RESCUE_CONST static const
#include "rescue_f128_constants.h"
#define AIR_HAVE_RESCUE 1
#define F_ONE ((u128)1)
#include "constraints_tmpl.inc"

/* rescue::apply_round — examples/src/rescue/rescue.rs:41-53 (sbox, MDS, first half of ARK; inverse sbox, MDS, second half) */
static void rescue_apply_round(u128 *st, uint64_t step) {
    const u128 *ark = RESCUE_ARK[step % 16];
    for (int i = 0; i < 4; i++) st[i] = f128_exp(st[i], 3);
    or_f128_rescue_mds(1, st, RESCUE_MDS);
    for (int i = 0; i < 4; i++) st[i] = f128_add(st[i], ark[i]);
    for (int i = 0; i < 4; i++) st[i] = f128_exp(st[i], RESCUE_INV_ALPHA);
    or_f128_rescue_mds(1, st, RESCUE_MDS);
    for (int i = 0; i < 4; i++) st[i] = f128_add(st[i], ark[4 + i]);
}
/* RescueProver::build_trace — examples/src/rescue/prover.rs:30-55 (TraceTable::fill: row i+1 = update(i, row i)).
 * trace: 4 columns of n = iterations * 16 elements, column-major. */
void or_f128_rescue_build_trace(const u128 seed[2], uint64_t iterations, u128 *trace) {
    const uint64_t n = iterations * 16;
    u128 st[4] = {seed[0], seed[1], 0, 0};
    for (uint64_t step = 0;; step++) {
        for (int c = 0; c < 4; c++) trace[c * n + step] = st[c];
        if (step + 1 == n) break;
        if (step % 16 < 14) rescue_apply_round(st, step);
        else st[2] = st[3] = 0;
    }
}

/* VdfProver::build_trace — examples/src/vdf/regular/prover.rs:30-41 and vdf/exempt/prover.rs:30-44 (exempt = 1: n - 1 real
 * states, then the garbage value 123 in the last row).  state' = (state - 42)^INV_ALPHA, INV_ALPHA = (2p - 1) / 3
 * (vdf/regular/mod.rs:30-32).  One column of n elements. */
void or_f128_vdf_build_trace(const u128 *seed, uint64_t n, int exempt, u128 *trace) {
    u128 ia = 0;
    {   /* INV_ALPHA = 226854911280625642308916371969163307691 (no 128-bit literals in C) */
        const char *dec = "226854911280625642308916371969163307691";
        for (const char *c = dec; *c; c++) ia = ia * 10 + (u128)(*c - '0');
    }
    u128 st = *seed;
    const uint64_t real = exempt ? n - 1 : n;
    for (uint64_t i = 0; i < real; i++) {
        trace[i] = st;
        st = f128_exp(f128_sub(st, 42), ia);
    }
    if (exempt) trace[n - 1] = 123;
}
