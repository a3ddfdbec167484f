// Field policies used by the field-generic kernels (NTT passes, LDE transpose, FRI fold).
//   F::T          element word type as stored in memory (reference's internal representation)
//   F::add/sub/mul
//   F::mul_w16(v, j, w16)   multiply by omega_16^j (j < 8): shifts for f64 (omega_16 = 2^12), table constants otherwise
//   F::ext_mul<D>           full extension-field product (FRI fold / Horner)
// Host halves (HostF64 / HostF128) do the table arithmetic on canonical integers.
#pragma once
#include "f128.cuh"
#include "gl64.cuh"
#include "../../include/winterfell_hip.h"

struct F64 {
    typedef uint64_t T;
    static constexpr int ID = WF_FIELD_F64;
    static constexpr int MAX_EXT = 3;
#ifndef NTT_F64_MAX_LOG_RADIX
#define NTT_F64_MAX_LOG_RADIX 8
#endif
    static constexpr uint32_t MAX_LOG_RADIX = NTT_F64_MAX_LOG_RADIX;
    static constexpr bool SHIFT_TWIDDLES = true;
#ifndef NTT_F64_NO_L24
    static constexpr bool USE_L24 = true;     // NTT passes run their register DFTs on 24-bit limbs (l24.cuh)
#else
    static constexpr bool USE_L24 = false;
#endif
    static __device__ __forceinline__ T add(T a, T b) { return gl::add(a, b); }
    static __device__ __forceinline__ T sub(T a, T b) { return gl::sub(a, b); }
    static __device__ __forceinline__ T mul(T a, T b) { return gl::mul(a, b); }
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ bool is_zero(T a) { return a == 0; }
    static __device__ __forceinline__ T load_norm(T a) { return a; }   // memory words are always canonical
    // NTT pass tables (omega_256^e, omega_16^j, small inter-pass tables): one word per entry
    static constexpr int TAB_WORDS = 1;
    static __device__ __forceinline__ T mul_tab(T v, const T *tab, uint64_t idx) { return gl::mul(v, tab[idx]); }
    static __device__ __forceinline__ T mul_w16_tab(T v, int j, const T *w) { return mul_w16(v, j, w); }
    static __device__ __forceinline__ T mul_w16(T v, int j, const T *) {
        switch (j) {
            case 0: return v;
            case 1: return gl::mul_pow2<12>(v);
            case 2: return gl::mul_pow2<24>(v);
            case 3: return gl::mul_pow2<36>(v);
            case 4: return gl::mul_pow2<48>(v);
            case 5: return gl::mul_pow2<60>(v);
            case 6: return gl::mul_pow2<72>(v);
            default: return gl::mul_pow2<84>(v);
        }
    }
    template <int D>
    static __device__ __forceinline__ void ext_mul(const T (&a)[D], const T (&b)[D], T (&o)[D]) { gl::ext_mul<D>(a, b, o); }
};

struct F128 {
    typedef f128::u128 T;
    static constexpr int ID = WF_FIELD_F128;
    static constexpr int MAX_EXT = 2;   // no cubic extension (math/src/field/f128/mod.rs:288-308)
#ifndef NTT_F128_MAX_LOG_RADIX
#define NTT_F128_MAX_LOG_RADIX 6
#endif
    static constexpr uint32_t MAX_LOG_RADIX = NTT_F128_MAX_LOG_RADIX;
    static constexpr bool SHIFT_TWIDDLES = false;
    static constexpr bool USE_L24 = false;
    static __device__ __forceinline__ T add(T a, T b) { return f128::add(a, b); }
    static __device__ __forceinline__ T sub(T a, T b) { return f128::sub(a, b); }
    static __device__ __forceinline__ T mul(T a, T b) { return f128::mul(a, b); }
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ bool is_zero(T a) { return a == 0; }
    static __device__ __forceinline__ T load_norm(T a) { return a; }
    static __device__ __forceinline__ T mul_w16(T v, int j, const T *w16) { return j == 0 ? v : f128::mul(v, w16[j]); }
    // NTT pass tables hold PAIRS (w, w * 2^64 mod p): a product with a table entry is a_lo w + a_hi (w 2^64), one fold instead of the
    // two of a 256-bit product (f128::mul_tab: ~65 instead of ~78 instructions; round 5)
    static constexpr int TAB_WORDS = 2;
#ifndef F128_PLAIN_TABLE_MUL
    static __device__ __forceinline__ T mul_tab(T v, const T *tab, uint64_t idx) { return f128::mul_tab(v, tab[2 * idx], tab[2 * idx + 1]); }
    static __device__ __forceinline__ T mul_w16_tab(T v, int j, const T *w16) { return j == 0 ? v : f128::mul_tab(v, w16[2 * j], w16[2 * j + 1]); }
#else   // A/B builds (tools/build_variant.sh): the same tables, the general product on the first word of a pair
    static __device__ __forceinline__ T mul_tab(T v, const T *tab, uint64_t idx) { return f128::mul(v, tab[2 * idx]); }
    static __device__ __forceinline__ T mul_w16_tab(T v, int j, const T *w16) { return j == 0 ? v : f128::mul(v, w16[2 * j]); }
#endif
    // quadratic extension x^2 - x - 1 (f128/mod.rs:267-272)
    template <int D>
    static __device__ __forceinline__ void ext_mul(const T (&a)[D], const T (&b)[D], T (&o)[D]) {
        if constexpr (D == 1) {
            o[0] = f128::mul(a[0], b[0]);
        } else {
            const T z = f128::mul(a[0], b[0]);
            const T t = f128::mul(f128::add(a[0], a[1]), f128::add(b[0], b[1]));
            o[0] = f128::add(z, f128::mul(a[1], b[1]));
            o[1] = f128::sub(t, z);
        }
    }
};

// f62: p = 2^62 - 111*2^39 + 1, Montgomery residues (R = 2^64).  The reference keeps lazy values in [0, 2M)
// (math/src/field/f62/mod.rs:61) and only ever observes them through normalize(); we keep every word normalised.
namespace f62 {
constexpr uint64_t M = 4611624995532046337ull;
constexpr uint64_t U = 4611624995532046335ull;   // -M^-1 mod 2^64 (f62/mod.rs:48)
__host__ __device__ __forceinline__ uint64_t norm(uint64_t v) { return v >= M ? v - M : v; }
__host__ __device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) { return norm(a + b); }
__host__ __device__ __forceinline__ uint64_t sub(uint64_t a, uint64_t b) { return a < b ? a + M - b : a - b; }
__host__ __device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b) {   // f62/mod.rs:559-564, then normalize
    const unsigned __int128 z = (unsigned __int128)a * b;
    const uint64_t q = (uint64_t)z * U;
    const unsigned __int128 r = z + (unsigned __int128)q * M;
    return norm((uint64_t)(r >> 64));
}
}  // namespace f62

struct F62 {
    typedef uint64_t T;
    static constexpr int ID = WF_FIELD_F62;
    static constexpr int MAX_EXT = 3;
    static constexpr uint32_t MAX_LOG_RADIX = 8;
    static constexpr bool SHIFT_TWIDDLES = false;
    static constexpr bool USE_L24 = false;
    static __device__ __forceinline__ T add(T a, T b) { return f62::add(a, b); }
    static __device__ __forceinline__ T sub(T a, T b) { return f62::sub(a, b); }
    static __device__ __forceinline__ T mul(T a, T b) { return f62::mul(a, b); }
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ bool is_zero(T a) { return a == 0; }
    static __device__ __forceinline__ T load_norm(T a) { return f62::norm(a); }   // accept the reference's lazy [0, 2M) words
    static __device__ __forceinline__ T mul_w16(T v, int j, const T *w16) { return j == 0 ? v : f62::mul(v, w16[j]); }
    static constexpr int TAB_WORDS = 1;
    static __device__ __forceinline__ T mul_tab(T v, const T *tab, uint64_t idx) { return f62::mul(v, tab[idx]); }
    static __device__ __forceinline__ T mul_w16_tab(T v, int j, const T *w) { return mul_w16(v, j, w); }
    template <int D>
    static __device__ __forceinline__ void ext_mul(const T (&a)[D], const T (&b)[D], T (&o)[D]) {
        if constexpr (D == 1) {
            o[0] = mul(a[0], b[0]);
        } else if constexpr (D == 2) {   // x^2 - x - 1 (f62/mod.rs:321-326)
            const T z = mul(a[0], b[0]);
            const T t = mul(add(a[0], a[1]), add(b[0], b[1]));
            o[0] = add(z, mul(a[1], b[1]));
            o[1] = sub(t, z);
        } else {                          // x^3 + 2x + 2 (f62/mod.rs:347-371)
            const T a0b0 = mul(a[0], b[0]), a1b1 = mul(a[1], b[1]), a2b2 = mul(a[2], b[2]);
            const T s01 = mul(add(a[0], a[1]), add(b[0], b[1]));
            const T m02 = mul(sub(a[0], a[2]), sub(b[2], b[0]));
            const T m12 = mul(sub(a[1], a[2]), sub(b[1], b[2]));
            const T s = add(a0b0, a1b1);
            const T u = sub(sub(m12, a1b1), a2b2);
            const T t = add(u, u);
            o[0] = add(a0b0, t);
            o[1] = sub(sub(add(s01, t), add(a2b2, a2b2)), s);
            o[2] = sub(add(m02, s), a2b2);
        }
    }
};

// ---- host halves: canonical-integer arithmetic for table construction ------------------------------------
struct HostF64 {
    typedef uint64_t T;
    typedef F64 Dev;
    static constexpr uint32_t TWO_ADICITY = 32;
    static T mulmod(T a, T b) { return (T)(((unsigned __int128)a * b) % gl::P); }
    static T from_u64(uint64_t v) { return v % gl::P; }
    static T to_internal(T canon) { return (T)((((unsigned __int128)canon) << 64) % gl::P); }   // Montgomery form
    static T powmod(T a, unsigned __int128 e) { T r = 1; while (e) { if (e & 1) r = mulmod(r, a); a = mulmod(a, a); e >>= 1; } return r; }
    static T invmod(T a) { return powmod(a, (unsigned __int128)gl::P - 2); }
    static T from_internal(T m) { return mulmod(m, invmod(to_internal(1))); }
    static bool valid_internal(T m) { return m < gl::P; }
    static T root_of_unity(uint32_t log_n) { return powmod(7277203076849721926ull, (unsigned __int128)1 << (32 - log_n)); }   // f64/mod.rs:267
};

struct HostF128 {
    typedef f128::u128 T;
    typedef F128 Dev;
    static constexpr uint32_t TWO_ADICITY = 40;
    static T mulmod(T a, T b) { return f128::mul(a, b); }
    static T from_u64(uint64_t v) { return (T)v; }
    static T to_internal(T canon) { return canon; }
    static T from_internal(T m) { return m; }
    static bool valid_internal(T m) { return m < f128::modulus(); }
    static T powmod(T a, unsigned __int128 e) { T r = 1; while (e) { if (e & 1) r = mulmod(r, a); a = mulmod(a, a); e >>= 1; } return r; }
    static T invmod(T a) { return powmod(a, f128::modulus() - 2); }
    static T root_of_unity(uint32_t log_n) {
        const T G = ((T)0x120532e7b364080aull << 64) | 0x86b8723e1920f4aaull;   // f128/mod.rs:43
        return powmod(G, (unsigned __int128)1 << (40 - log_n));
    }
};

struct HostF62 {
    typedef uint64_t T;
    typedef F62 Dev;
    static constexpr uint32_t TWO_ADICITY = 39;
    static T mulmod(T a, T b) { return (T)(((unsigned __int128)a * b) % f62::M); }
    static T from_u64(uint64_t v) { return v % f62::M; }
    static T to_internal(T canon) { return (T)((((unsigned __int128)canon) << 64) % f62::M); }
    static T powmod(T a, unsigned __int128 e) { T r = 1; while (e) { if (e & 1) r = mulmod(r, a); a = mulmod(a, a); e >>= 1; } return r; }
    static T invmod(T a) { return powmod(a, (unsigned __int128)f62::M - 2); }
    static T from_internal(T m) { return mulmod(m % f62::M, invmod(to_internal(1))); }
    static bool valid_internal(T m) { return m < 2 * f62::M; }   // lazy words of the reference are accepted and normalised
    static T root_of_unity(uint32_t log_n) { return powmod(4421547261963328785ull, (unsigned __int128)1 << (39 - log_n)); }   // f62/mod.rs:54
};
