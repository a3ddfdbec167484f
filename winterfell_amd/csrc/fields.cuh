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
    static constexpr bool SHIFT_TWIDDLES = true;
    static __device__ __forceinline__ T add(T a, T b) { return gl::add(a, b); }
    static __device__ __forceinline__ T sub(T a, T b) { return gl::sub(a, b); }
    static __device__ __forceinline__ T mul(T a, T b) { return gl::mul(a, b); }
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ bool is_zero(T a) { return a == 0; }
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
    static constexpr bool SHIFT_TWIDDLES = false;
    static __device__ __forceinline__ T add(T a, T b) { return f128::add(a, b); }
    static __device__ __forceinline__ T sub(T a, T b) { return f128::sub(a, b); }
    static __device__ __forceinline__ T mul(T a, T b) { return f128::mul(a, b); }
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ bool is_zero(T a) { return a == 0; }
    static __device__ __forceinline__ T mul_w16(T v, int j, const T *w16) { return j == 0 ? v : f128::mul(v, w16[j]); }
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
