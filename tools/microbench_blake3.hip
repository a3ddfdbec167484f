// BLAKE3 compression rate in registers (no memory traffic): one merge (64-byte block) per lane per iteration, against the
// rate the Merkle / row-hash kernels reach.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iwinterfell_amd/csrc
// tools/microbench_blake3.hip -o tools/microbench_blake3.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../winterfell_amd/csrc/blake3.cuh"

#define ITERS 512

template <int NH>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    extern __shared__ uint32_t dyn[];
    uint32_t m[NH][16], o[NH][8];
    for (int h = 0; h < NH; h++)
        for (int i = 0; i < 16; i++) m[h][i] = seed * (i + 3 + 17 * h) + threadIdx.x * 0x9E3779B9u;
    if (seed == 1) dyn[threadIdx.x] = 1;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int h = 0; h < NH; h++) {
            b3::merge(m[h], o[h]);
#pragma unroll
            for (int i = 0; i < 8; i++) m[h][i] ^= o[h][i];
        }
    }
    uint32_t s = 0;
    for (int h = 0; h < NH; h++)
        for (int i = 0; i < 16; i++) s ^= m[h][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NH>
static void run(int lds) {
    const int blocks = 256 * 8, threads = 256;
    uint32_t *d;
    hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<NH>, dim3(blocks), dim3(threads), lds, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<NH>, dim3(blocks), dim3(threads), lds, 0, d, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double comps = (double)blocks * threads * ITERS * NH;
    printf("%d independent hashes per lane: %8.3f ms  %6.2f e9 compressions/s  %7.0f nominal cycles per wave-compression per SIMD\n", NH, ms,
           comps / (ms * 1e-3) / 1e9, (ms * 1e-3) * 2.4e9 * 1024 / (comps / 64));
    hipFree(d);
}

int main(int argc, char **argv) {
    const int lds = argc > 1 ? atoi(argv[1]) : 0;
    run<1>(lds);
    run<2>(lds);
    return 0;
}
