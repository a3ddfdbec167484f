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

/* RescueRapsProver::build_trace — examples/src/rescue_raps/prover.rs:36-94 (RapTraceTable::fill): two hash chains side by side,
 * chain 0 absorbing seeds[k], chain 1 permuted_seeds[k] (both chain_length x 2 elements) at step 14 of every 16-step cycle.
 * trace: 8 columns of n = chain_length * 16 elements, column-major. */
void or_f128_rescue_raps_build_trace(const u128 *seeds, const u128 *permuted, uint64_t chain_length, u128 *trace) {
    const uint64_t n = chain_length * 16;
    u128 st[8] = {seeds[0], seeds[1], 0, 0, permuted[0], permuted[1], 0, 0};
    for (uint64_t step = 0;; step++) {
        for (int c = 0; c < 8; c++) trace[c * n + step] = st[c];
        if (step + 1 == n) break;
        if (step % 16 < 14) {                                  /* apply_rescue_round_parallel, mod.rs:184-190 */
            rescue_apply_round(st, step);
            rescue_apply_round(st + 4, step);
        } else if (step % 16 == 14) {
            const uint64_t idx = step / 16 + 1;
            if (idx < chain_length) {
                st[0] = f128_add(st[0], seeds[2 * idx]);
                st[1] = f128_add(st[1], seeds[2 * idx + 1]);
                st[4] = f128_add(st[4], permuted[2 * idx]);
                st[5] = f128_add(st[5], permuted[2 * idx + 1]);
            }
        }
    }
}

/* QuadExtension::inv (math/src/field/extensions/quadratic.rs:81-94): frobenius(x) / norm(x), frobenius = [x0 + x1, -x1] (mod.rs:280-282) */
static void f128_extD_inv(unsigned D, const u128 *x, u128 *out) {
    if (D == 1) { out[0] = f128_inv(x[0]); return; }
    if (x[0] == 0 && x[1] == 0) { out[0] = out[1] = 0; return; }
    const u128 num[2] = {f128_add(x[0], x[1]), f128_sub(0, x[1])};
    u128 norm[2];
    f128_ext2_mul(x, num, norm);
    const u128 di = f128_inv(norm[0]);
    out[0] = f128_mul(num[0], di);
    out[1] = f128_mul(num[1], di);
}

/* RescueRapsProver::build_aux_trace — rescue_raps/prover.rs:157-205.  trace: the 8 main columns (column-major, n rows);
 * rand: 3 elements of E (D components each); aux: 3 columns of n E elements, column-major (aux[(c * n + i) * D + d]). */
void or_f128_rescue_raps_build_aux(const u128 *trace, uint64_t n, unsigned D, const u128 *rand, u128 *aux) {
    u128 *a0 = aux, *a1 = aux + n * D, *a2 = aux + 2 * n * D;
    memset(aux, 0, 3 * n * D * sizeof(u128));
    u128 t[2], u[2], v[2], lift[2] = {0, 0};
#define RAPS_COMBINE(dst, x0, x1)                                                                  \
    lift[0] = (x0); or_f128_e_mul(D, rand, lift, u);                                                \
    lift[0] = (x1); or_f128_e_mul(D, rand + D, lift, v);                                            \
    or_f128_e_add(D, u, v, (dst));
    RAPS_COMBINE(a0, trace[0], trace[n])                       /* row 0: alpha_0 * state[0] + alpha_1 * state[1] */
    RAPS_COMBINE(a1, trace[4 * n], trace[5 * n])
    a2[0] = 1;
    for (uint64_t i = 1; i < n; i++) {
        if (i % 16 == 14) {                                    /* the absorbed values = next - current on the rate registers */
            RAPS_COMBINE(a0 + i * D, f128_sub(trace[i + 1], trace[i]), f128_sub(trace[n + i + 1], trace[n + i]))
            RAPS_COMBINE(a1 + i * D, f128_sub(trace[4 * n + i + 1], trace[4 * n + i]), f128_sub(trace[5 * n + i + 1], trace[5 * n + i]))
        }
        or_f128_e_add(D, a0 + (i - 1) * D, rand + 2 * D, u);   /* num */
        or_f128_e_add(D, a1 + (i - 1) * D, rand + 2 * D, v);   /* denom */
        f128_extD_inv(D, v, t);
        or_f128_e_mul(D, a2 + (i - 1) * D, u, v);
        or_f128_e_mul(D, v, t, a2 + i * D);
    }
#undef RAPS_COMBINE
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
