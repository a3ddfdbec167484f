// Throughput of the f128 field primitives (cycles per wave-op), gfx950.  Build: see tools/Makefile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../winterfell_amd/csrc/f128.cuh"

#define ITERS 1024
#define CH 4
typedef unsigned __int128 u128;

// -DF128_ALT -include <candidate>.cuh benchmarks an alternative namespace f128alt next to the shipped one

template <int OP>
__global__ __launch_bounds__(256) void k(u128 *out, uint64_t seed) {
    u128 x[CH], y[CH];
    for (int i = 0; i < CH; i++) {
        x[i] = (((u128)(seed * (i + 3) + threadIdx.x)) << 64 | (seed * (i + 5))) % f128::modulus();
        y[i] = (((u128)(seed * (i + 11) + 7 * threadIdx.x)) << 64 | (seed * (i + 13))) % f128::modulus();
    }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (OP == 0) x[i] = f128::mul(x[i], y[i]);
            else if (OP == 1) x[i] = f128::add(x[i], y[i]);
            else if (OP == 2) x[i] = f128::sub(x[i], y[i]);
            else if (OP == 3) { u128 u = x[i], v = y[i]; x[i] = f128::add(u, v); y[i] = f128::sub(u, v); }
#ifdef F128_ALT
            else if (OP == 4) x[i] = f128alt::mul(x[i], y[i]);
            else if (OP == 5) x[i] = f128alt::add(x[i], y[i]);
            else if (OP == 6) x[i] = f128alt::sub(x[i], y[i]);
            else if (OP == 7) { u128 u = x[i], v = y[i]; x[i] = f128alt::add(u, v); y[i] = f128alt::sub(u, v); }
#endif
        }
    }
    u128 s = 0;
    for (int i = 0; i < CH; i++) s ^= x[i] ^ y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name) {
    const int blocks = 256 * 8, threads = 256;
    u128 *d;
    (void)hipMalloc(&d, (size_t)blocks * threads * 16);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x123456789abcdefull);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x123456789abcdefull);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * threads * ITERS * CH;
    printf("%-22s %8.3f ms  %7.3f Tlane-op/s  %6.1f cycles/wave-op/SIMD@2.4GHz\n", name, ms, ops / (ms * 1e-3) / 1e12,
           (ms * 1e-3) * 2.4e9 * 1024 / (ops / 64));
    (void)hipFree(d);
}

int main() {
    run<0>("f128 mul");
    run<1>("f128 add");
    run<2>("f128 sub");
    run<3>("f128 butterfly");
#ifdef F128_ALT
    run<4>("f128 mul (alt)");
    run<5>("f128 add (alt)");
    run<6>("f128 sub (alt)");
    run<7>("f128 butterfly (alt)");
#endif
    return 0;
}
