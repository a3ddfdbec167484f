// Cycle cost of the pieces of an f64 NTT pass, in isolation (no memory traffic): the 16-point register DFT on 24-bit limbs
// (l24.cuh) against the canonical one (dft_regs.cuh), leaving the limb form with and without a table twiddle, the Montgomery
// product.  Each kernel iterates on registers; the time per wave-op is reported in nominal 2.4 GHz cycles per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iwinterfell_amd/csrc tools/microbench_l24.hip -o tools/microbench_l24.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../winterfell_amd/csrc/dft_regs.cuh"
#include "../winterfell_amd/csrc/l24.cuh"

#define ITERS 256

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, const uint64_t *tab, uint64_t seed) {
    extern __shared__ uint32_t dyn[];
    uint64_t x[16];
    for (int i = 0; i < 16; i++) x[i] = (seed * (i + 3) + threadIdx.x * 0x9E3779B97F4A7C15ull) % gl::P;
    if (seed == 1) dyn[threadIdx.x] = 1;
    uint64_t w[4] = {tab[threadIdx.x & 3], tab[4 + (threadIdx.x & 3)], tab[8], tab[12]};
    for (int it = 0; it < ITERS; it++) {
        if constexpr (MODE == 0) {            // canonical 16-point DFT
            dft_dif<F64, 4>(x, nullptr);
        } else if constexpr (MODE == 1) {     // limb DFT + plain conversion back (canonical)
            typedef l24::Dft<4> D;
            int32_t v[D::NV];
#pragma unroll
            for (int e = 0; e < 16; e++) D::load(v, e, x[e]);
            D::run(v);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                uint32_t y[4];
#pragma unroll
                for (int q = 0; q < 4; q++) y[q] = D::limb(v, e, q);
                x[e] = l24::fold(l24::mul4_one(y));
            }
        } else if constexpr (MODE == 2) {     // limb DFT + table multiplication (lazy)
            typedef l24::Dft<4> D;
            int32_t v[D::NV];
#pragma unroll
            for (int e = 0; e < 16; e++) D::load(v, e, x[e]);
            D::run(v);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                uint32_t y[4];
#pragma unroll
                for (int q = 0; q < 4; q++) y[q] = D::limb(v, e, q);
                x[e] = l24::fold_lazy(l24::mul4(y, w[0], w[1], w[2], w[3]));
            }
        } else if constexpr (MODE == 3) {     // 16 Montgomery products
#pragma unroll
            for (int e = 0; e < 16; e++) x[e] = gl::mul(x[e], w[e & 3]);
        } else if constexpr (MODE == 4) {     // limb DFT only (split + butterflies; outputs summed so nothing is dead)
            typedef l24::Dft<4> D;
            int32_t v[D::NV];
#pragma unroll
            for (int e = 0; e < 16; e++) D::load(v, e, x[e]);
            D::run(v);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                uint32_t y[4];
#pragma unroll
                for (int q = 0; q < 4; q++) y[q] = D::limb(v, e, q);
                x[e] = ((uint64_t)(y[0] ^ y[2]) << 32) | (y[1] ^ y[3]);
            }
        } else if constexpr (MODE == 5) {     // leaving the limb form only: 16 x (mul4 + fold_lazy) on given limbs
#pragma unroll
            for (int e = 0; e < 16; e++) {
                uint32_t y[4] = {(uint32_t)x[e] & 0x3fffffffu, (uint32_t)(x[e] >> 32) & 0x3fffffffu, (uint32_t)x[(e + 1) & 15] & 0x3fffffffu,
                                 (uint32_t)(x[(e + 1) & 15] >> 32) & 0x3fffffffu};
                x[e] = l24::fold_lazy(l24::mul4(y, w[0], w[1], w[2], w[3]));
            }
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 16; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, int lds) {
    const int blocks = 256 * 8, threads = 256;
    uint64_t *d, *t;
    hipMalloc(&d, (size_t)blocks * threads * 8);
    hipMalloc(&t, 16 * 8);
    uint64_t h[16];
    for (int i = 0; i < 16; i++) h[i] = 0x123456789abcdefull * (i + 1) % gl::P;
    hipMemcpy(t, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, 0, d, t, 0x1234567ull);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds, 0, d, t, 0x1234567ull);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_ops = (double)blocks * 4 * ITERS * 16;   // per-element operations, per wave
    const double cyc = (ms * 1e-3) * 2.4e9 * 1024 / wave_ops;
    printf("%-44s %8.3f ms  %7.1f cycles per element (wave-op, per SIMD, at 2.4 GHz)\n", name, ms, cyc);
    hipFree(d);
    hipFree(t);
}

int main(int argc, char **argv) {
    const int lds = argc > 1 ? atoi(argv[1]) : 0;
    run<0>("canonical DFT16 (dft_dif<F64,4>)", lds);
    run<4>("limb DFT16 only (split + butterflies)", lds);
    run<1>("limb DFT16 + convert back (canonical)", lds);
    run<2>("limb DFT16 + table multiply (lazy)", lds);
    run<5>("mul4 + fold_lazy only", lds);
    run<3>("Montgomery product", lds);
    return 0;
}
