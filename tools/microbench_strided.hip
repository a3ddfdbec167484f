// Can a 2-pass (4096 x 4096) transform of a 2^24-point f64 vector stream its strided side at HBM speed?  A workgroup owns
// all 4096 rows of a group of C adjacent columns (C x 8 bytes contiguous per row, rows 32 KiB apart) — the access pattern of
// a radix-4096 pass whose tile must fit LDS (C = 2: 64 KiB, C = 4: 128 KiB).  Measured: read-modify-write in place, for
// C = 2 / 4 / 8 / 16, with the column groups handed to workgroups in dispatch order or XCD-aware (adjacent groups on the
// same XCD so that the partial 128-byte lines meet in one L2).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_strided.hip -o tools/microbench_strided.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int LOG_ROWS = 12, ROWS = 1 << LOG_ROWS, ROW_ELEMS = 4096;

// V = per-lane vector (uint2 = 1 element, uint4 = 2 elements), LPR = lanes per row (C = LPR * sizeof(V) / 8)
template <class V, int LPR, bool XCD, int MODE>
__global__ __launch_bounds__(256) void k(V *data, int groups) {
    int g = blockIdx.x;
    if (XCD) {   // block b runs on XCD b % 8: give XCD x the contiguous range of groups [x * groups / 8, (x + 1) * groups / 8)
        g = (blockIdx.x & 7) * (groups >> 3) + (blockIdx.x >> 3);
    }
    constexpr int VPR = ROW_ELEMS * 8 / sizeof(V);        // vectors per matrix row
    constexpr int ROWS_PER_ITER = 256 / LPR;
    const int lane_col = threadIdx.x % LPR, lane_row = threadIdx.x / LPR;
    V *base = data + (size_t)g * LPR + lane_col;
    constexpr int ITERS = ROWS / ROWS_PER_ITER;
    // 16 loads in flight per lane, like an NTT tile
    for (int it0 = 0; it0 < ITERS; it0 += 16) {
        V v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const size_t r = (size_t)(it0 + u) * ROWS_PER_ITER + lane_row;
            if (MODE != 1) v[u] = base[r * VPR];
            else v[u] = V{};
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const size_t r = (size_t)(it0 + u) * ROWS_PER_ITER + lane_row;
            v[u].x += 1;
            if (MODE != 0) base[r * VPR] = v[u];
            else if (v[u].x == 0x12345) base[r * VPR] = v[u];
        }
    }
}

template <class V, int LPR, bool XCD, int MODE>
static void run(const char *what) {
    const size_t bytes = (size_t)ROWS * ROW_ELEMS * 8;   // 128 MiB
    V *d;
    hipMalloc(&d, bytes);
    hipMemset(d, 1, bytes);
    const int C = LPR * sizeof(V) / 8;
    const int groups = ROW_ELEMS / C;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<V, LPR, XCD, MODE>), dim3(groups), dim3(256), 0, 0, d, groups);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(a);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k<V, LPR, XCD, MODE>), dim3(groups), dim3(256), 0, 0, d, groups);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double moved = (MODE == 2 ? 2.0 : 1.0) * bytes * reps;
    printf("C = %2d columns (%3d-byte runs, %zu B/lane)  %-10s %-9s %8.1f us/pass  %5.2f TB/s\n", C, C * 8, sizeof(V), what, XCD ? "xcd-aware" : "dispatch",
           ms * 1e3 / reps, moved / (ms * 1e-3) / 1e12);
    hipFree(d);
}

template <class V, int LPR>
static void all() {
    run<V, LPR, false, 0>("read");
    run<V, LPR, true, 0>("read");
    run<V, LPR, false, 1>("write");
    run<V, LPR, true, 1>("write");
    run<V, LPR, false, 2>("read+write");
    run<V, LPR, true, 2>("read+write");
}

int main() {
    all<uint4, 1>();   // C = 2
    all<uint2, 2>();   // C = 2 with 8-byte lanes
    all<uint4, 2>();   // C = 4
    all<uint4, 4>();   // C = 8
    all<uint2, 8>();   // C = 8 with 8-byte lanes
    all<uint4, 8>();   // C = 16
    all<uint2, 16>();  // C = 16 with 8-byte lanes (the current 3-pass tiles)
    return 0;
}
