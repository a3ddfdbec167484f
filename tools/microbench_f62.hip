// Throughput of the f62 field primitives (cycles per wave-op), gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../include/winterfell_hip.h"
#include "../winterfell_amd/csrc/fields.cuh"

#define ITERS 2048
#define CH 8

// candidate: 32-bit-limb Montgomery with q = -t0 (M = M1 * 2^32 + 1  =>  -M^-1 = -1 mod 2^32)
__device__ __forceinline__ uint64_t mul_limb(uint64_t a, uint64_t b) {
    constexpr uint32_t M1 = 0x3FFFC880u;
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t t = (uint64_t)a0 * b0;
    const uint32_t z0 = (uint32_t)t;
    t = (uint64_t)a0 * b1 + (t >> 32);
    const uint32_t l1 = (uint32_t)t, h1 = (uint32_t)(t >> 32);
    t = (uint64_t)a1 * b0;
    const uint32_t m0 = (uint32_t)t;
    t = (uint64_t)a1 * b1 + (t >> 32);
    const uint32_t m1 = (uint32_t)t, k1 = (uint32_t)(t >> 32);
    uint32_t c;
    uint32_t z1 = __builtin_addc(l1, m0, 0u, &c);
    uint32_t z2 = __builtin_addc(h1, m1, c, &c);
    uint32_t z3 = __builtin_addc(k1, 0u, c, &c);
    // step 1: q = -z0; z + q*M: limb 0 becomes 0 with carry (z0 != 0); add q*M1 at limb 1
    uint32_t q = 0u - z0;
    t = (uint64_t)q * M1 + z1;                     // < 2^62 + 2^32
    uint32_t c0 = z0 != 0;
    uint32_t y0 = __builtin_addc((uint32_t)t, c0, 0u, &c);
    uint32_t y1 = __builtin_addc(z2, (uint32_t)(t >> 32), c, &c);
    uint32_t y2 = __builtin_addc(z3, 0u, c, &c);
    // step 2
    q = 0u - y0;
    t = (uint64_t)q * M1 + y1;
    c0 = y0 != 0;
    uint32_t r0 = __builtin_addc((uint32_t)t, c0, 0u, &c);
    uint32_t r1 = __builtin_addc(y2, (uint32_t)(t >> 32), c, &c);
    const uint64_t r = ((uint64_t)r1 << 32) | r0;
    return f62::norm(r);
}

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t seed) {
    uint64_t x[CH], y[CH];
    for (int i = 0; i < CH; i++) { x[i] = (seed * (i + 3) + threadIdx.x) % f62::M; y[i] = (seed * (i + 11) + 7 * threadIdx.x) % f62::M; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) x[i] = f62::mul(x[i], y[i]);
            else if (OP == 1) x[i] = f62::add(x[i], y[i]);
            else if (OP == 2) x[i] = f62::sub(x[i], y[i]);
            else if (OP == 3) x[i] = mul_limb(x[i], y[i]);
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < CH; i++) s ^= x[i] ^ y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void check(uint64_t *bad) {
    uint64_t s = 0x9E3779B97F4A7C15ull + threadIdx.x * 977 + blockIdx.x * 131071;
    for (int i = 0; i < 2000; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint64_t a = s % f62::M;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint64_t b = s % f62::M;
        if (i % 5 == 0) a = f62::M - 1 - (i % 3);
        if (i % 7 == 0) b = (i % 2) ? 0 : 0xFFFFFFFFull;
        if (mul_limb(a, b) != f62::mul(a, b)) atomicAdd((unsigned long long *)bad, 1ull);
    }
}

template <int OP>
void run(const char *name) {
    const int blocks = 256 * 8, threads = 256;
    uint64_t *d;
    (void)hipMalloc(&d, (size_t)blocks * threads * 8);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x123456789abcdefull);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x123456789abcdefull);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * threads * ITERS * CH;
    printf("%-22s %8.3f ms  %6.1f cycles/wave-op/SIMD@2.4GHz\n", name, ms, (ms * 1e-3) * 2.4e9 * 1024 / (ops / 64));
    (void)hipFree(d);
}

int main() {
    uint64_t *bad, h = 0;
    (void)hipMalloc(&bad, 8);
    (void)hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(check, dim3(64), dim3(256), 0, 0, bad);
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("mul_limb mismatches: %llu\n", (unsigned long long)h);
    run<0>("f62 mul"); run<3>("f62 mul (limbs)"); run<1>("f62 add"); run<2>("f62 sub");
    return 0;
}
