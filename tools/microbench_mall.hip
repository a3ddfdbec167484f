// Streaming read-modify-write bandwidth as a function of the working-set size (does a 128 MiB vector that is rewritten
// in place pass after pass run out of the 256 MiB Infinity Cache?), 8-byte and 16-byte accesses per lane, in place and
// ping-pong between two buffers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_mall.hip -o tools/microbench_mall.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <class V>
__global__ __launch_bounds__(256) void rmw(const V *src, V *dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        V v = src[i];
        v.x += 1;
        dst[i] = v;
    }
}
// each workgroup owns a contiguous 32 KiB tile (like an NTT tile): 16 loads per lane, then 16 stores
template <class V>
__global__ __launch_bounds__(256) void rmw_tile(const V *src, V *dst, size_t n) {
    constexpr int PER = 16;
    const size_t base = (size_t)blockIdx.x * 256 * PER;
    V v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) v[k] = src[base + k * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        v[k].x += 1;
        dst[base + k * 256 + threadIdx.x] = v[k];
    }
}

template <class V>
static void run(size_t bytes, bool pingpong, bool tile) {
    V *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipMemset(b, 1, bytes);
    const size_t n = bytes / sizeof(V);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 20;
    auto launch = [&](int r) {
        const V *s = (pingpong && (r & 1)) ? b : a;
        V *d = pingpong ? ((r & 1) ? a : b) : a;
        if (tile) hipLaunchKernelGGL(rmw_tile<V>, dim3((unsigned)(n / (256 * 16))), dim3(256), 0, 0, s, d, n);
        else hipLaunchKernelGGL(rmw<V>, dim3(256 * 16), dim3(256), 0, 0, s, d, n);
    };
    for (int r = 0; r < 4; r++) launch(r);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) launch(r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double tbs = 2.0 * bytes * reps / (ms * 1e-3) / 1e12;
    printf("%4zu MiB  %2zu B/lane  %-9s %-6s  %8.3f us/pass  %6.2f TB/s (read+write)\n", bytes >> 20, sizeof(V), pingpong ? "ping-pong" : "in place",
           tile ? "tile" : "grid", ms * 1e3 / reps, tbs);
    hipFree(a);
    hipFree(b);
}

int main(int argc, char **argv) {
    if (argc > 1) {   // "calib": one known byte count per access width, beyond every cache (for the rocprofv3 counter calibration)
        run<uint2>((size_t)1024 << 20, true, false);
        run<uint4>((size_t)1024 << 20, true, false);
        return 0;
    }
    for (size_t mib : {16, 32, 64, 128, 192, 256, 512, 1024}) {
        for (int tile = 0; tile < 2; tile++) {
            run<uint2>(mib << 20, false, tile);
            run<uint4>(mib << 20, false, tile);
            run<uint2>(mib << 20, true, tile);
            run<uint4>(mib << 20, true, tile);
        }
    }
    return 0;
}
