// Throughput of the Goldilocks primitives (cycles per lane-op) for alternative formulations, gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../winterfell_amd/csrc/gl64.cuh"

#define ITERS 2048
#define CH 8
typedef unsigned __int128 u128;
constexpr uint64_t P = gl::P, EPS = gl::EPS;

__device__ __forceinline__ uint64_t add_a1(uint64_t a, uint64_t b) { uint64_t s = a + b, t = s + EPS; return (s < a || s >= P) ? t : s; }
__device__ __forceinline__ uint64_t add_a2(uint64_t a, uint64_t b) { uint64_t s = a + b, t = s + EPS; return ((s < a) | (t < s)) ? t : s; }
__device__ __forceinline__ uint64_t sub_s1(uint64_t a, uint64_t b) { uint64_t d = a - b; return (a < b) ? d - EPS : d; }
// 32-bit limb versions with explicit carry intrinsics
__device__ __forceinline__ uint64_t sub_s2(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint32_t dl, dh, bo;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n v_subb_co_u32 %1, vcc, %4, %6, vcc\n v_subbrev_co_u32 %2, vcc, 0, 0, vcc"
        : "=&v"(dl), "=&v"(dh), "=&v"(bo) : "v"(al), "v"(ah), "v"(bl), "v"(bh) : "vcc");
    // bo = 0 or 0xFFFFFFFF (= -borrow).  d - EPS*borrow = d + borrow - borrow*2^32 : lo += borrow ; hi = hi - borrow + carry
    uint32_t rl, rh;
    asm("v_sub_co_u32 %0, vcc, %2, %4\n v_addc_co_u32 %1, vcc, %3, %4, vcc" : "=&v"(rl), "=&v"(rh) : "v"(dl), "v"(dh), "v"(bo) : "vcc");
    return ((uint64_t)rh << 32) | rl;
}
__device__ __forceinline__ uint64_t mul_v2(uint64_t a, uint64_t b) {
    uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t p00 = (uint64_t)a0 * b0, p01 = (uint64_t)a0 * b1, p10 = (uint64_t)a1 * b0, p11 = (uint64_t)a1 * b1;
    uint64_t mid = (p00 >> 32) + (uint32_t)p01 + (uint32_t)p10;
    uint64_t lo = (uint32_t)p00 | (mid << 32);
    uint64_t hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
    return gl::mont_red(lo, hi);
}

// row-wise 64 x 64 product: (l0, l1, h1) = a0 * b, (m0, m1, k1) = a1 * b, summed with one carry chain
__device__ __forceinline__ uint64_t mul_v3(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t t = (uint64_t)a0 * b0;
    const uint32_t l0 = (uint32_t)t;
    t = (uint64_t)a0 * b1 + (t >> 32);
    const uint32_t l1 = (uint32_t)t, h1 = (uint32_t)(t >> 32);
    t = (uint64_t)a1 * b0;
    const uint32_t m0 = (uint32_t)t;
    t = (uint64_t)a1 * b1 + (t >> 32);
    const uint32_t m1 = (uint32_t)t, k1 = (uint32_t)(t >> 32);
    uint32_t c;
    const uint32_t p1 = __builtin_addc(l1, m0, 0u, &c);
    const uint32_t p2 = __builtin_addc(h1, m1, c, &c);
    const uint32_t p3 = __builtin_addc(k1, 0u, c, &c);
    return gl::mont_red(gl::join(l0, p1), gl::join(p2, p3));
}

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t seed) {
    uint64_t x[CH], y[CH];
    for (int i = 0; i < CH; i++) { x[i] = (seed * (i + 3) + threadIdx.x) % P; y[i] = (seed * (i + 11) + 7 * threadIdx.x) % P; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) x[i] = gl::mul(x[i], y[i]);
            else if (OP == 1) x[i] = gl::add(x[i], y[i]);
            else if (OP == 2) x[i] = gl::sub(x[i], y[i]);
            else if (OP == 3) x[i] = gl::mul_pow2<36>(x[i]);
            else if (OP == 4) x[i] = add_a1(x[i], y[i]);
            else if (OP == 5) x[i] = add_a2(x[i], y[i]);
            else if (OP == 6) x[i] = sub_s1(x[i], y[i]);
            else if (OP == 7) x[i] = sub_s2(x[i], y[i]);
            else if (OP == 8) x[i] = mul_v2(x[i], y[i]);
            else if (OP == 9) { uint64_t u = x[i], v = y[i]; x[i] = gl::add(u, v); y[i] = gl::sub(u, v); }
            else if (OP == 10) x[i] = gl::mul_pow2<12>(x[i]);
            else if (OP == 11) x[i] = gl::mul_pow2<84>(x[i]);
            else if (OP == 12) x[i] = gl::mul_pow2<48>(x[i]);
            else if (OP == 13) x[i] = mul_v3(x[i], y[i]);
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < CH; i++) s ^= x[i] ^ y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name) {
    const int blocks = 256 * 8, threads = 256;
    uint64_t *d;
    hipMalloc(&d, (size_t)blocks * threads * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x123456789abcdefull);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x123456789abcdefull);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * threads * ITERS * CH;
    // SIMD-cycles per wave-op at 2.4 GHz nominal, 1024 SIMDs
    printf("%-22s %8.3f ms  %7.3f Tlane-op/s  %6.1f cycles/wave-op/SIMD@2.4GHz\n", name, ms, ops / (ms * 1e-3) / 1e12,
           (ms * 1e-3) * 2.4e9 * 1024 / (ops / 64));
    hipFree(d);
}

int main() {
    run<0>("mul (u128)"); run<8>("mul (32-bit limbs)"); run<13>("mul (rows)");
    run<1>("add (ref form)"); run<4>("add a1"); run<5>("add a2");
    run<2>("sub (ref form)"); run<6>("sub s1"); run<7>("sub s2 (asm)");
    run<9>("butterfly add+sub");
    run<10>("mul_pow2<12>"); run<3>("mul_pow2<36>"); run<12>("mul_pow2<48>"); run<11>("mul_pow2<84>");
    return 0;
}
