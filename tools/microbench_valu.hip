// Throughput microbenchmark for the integer VALU instructions the Goldilocks arithmetic is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o gpurun_out/microbench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 4096
#define CHAINS 8

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint64_t seed) {
    uint64_t x[CHAINS];
    uint32_t a = (uint32_t)seed + threadIdx.x, b = (uint32_t)(seed >> 32) | 1;
    for (int i = 0; i < CHAINS; i++) x[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) {  // v_mad_u64_u32
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
            } else if (OP == 1) {  // v_mul_lo_u32
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b));
                x[i] = lo;
            } else if (OP == 2) {  // v_mul_hi_u32
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(b));
                x[i] = lo;
            } else if (OP == 3) {  // v_add_u32 (full rate reference)
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(b));
                x[i] = lo;
            } else if (OP == 4) {  // v_lshl_add_u64 (64-bit add)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[i]) : "v"(seed));
            } else if (OP == 5) {  // v_mul_u32_u24
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(b));
                x[i] = lo;
            } else if (OP == 6) {  // v_add_co_u32 + v_addc_co_u32 pair
                uint32_t lo = (uint32_t)x[i], hi = (uint32_t)(x[i] >> 32);
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
                x[i] = ((uint64_t)hi << 32) | lo;
            } else if (OP == 7) {  // v_mad_u32_u24
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(lo) : "v"(b));
                x[i] = lo;
            } else if (OP == 8) {  // v_cndmask_b32
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(b) : "vcc");
                x[i] = lo;
            } else if (OP == 10) {  // v_alignbit_b32 (rotate)
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(lo));
                x[i] = lo;
            } else if (OP == 11) {  // v_perm_b32 (byte rotate)
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(lo) : "v"(b));
                x[i] = lo;
            } else if (OP == 12) {  // v_add3_u32
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b));
                x[i] = lo;
            } else if (OP == 13) {  // v_xor_b32
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(lo) : "v"(b));
                x[i] = lo;
            } else if (OP == 14) {  // v_xad_u32 (xor then add)
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b));
                x[i] = lo;
            } else if (OP == 15) {  // v_mov_b32
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "v"(lo));
                x[i] = lo;
            } else if (OP == 16) {  // v_lshlrev_b64
                asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x[i]));
            } else if (OP == 9) {  // v_cmp_lt_u64 + v_cndmask
                uint32_t lo = (uint32_t)x[i];
                asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(lo) : "v"(x[i]), "v"(seed), "v"(b) : "vcc");
                x[i] = (x[i] & 0xffffffff00000000ull) | lo;
            }
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < CHAINS; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(const char *name, int instr_per_iter) {
    const int blocks = 256 * 8, threads = 256;
    uint64_t *d;
    hipMalloc(&d, (size_t)blocks * threads * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x123456789abcdefull);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x123456789abcdefull);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double lane_ops = (double)blocks * threads * ITERS * CHAINS * instr_per_iter;
    double tops = lane_ops / (ms * 1e-3) / 1e12;
    // cycles per wave-instruction per SIMD at 2.4 GHz with 1024 SIMDs
    double wave_instr = lane_ops / 64.0;
    double cyc = (ms * 1e-3) * 2.4e9 * 1024 / wave_instr;
    printf("%-28s %8.3f ms  %7.2f Tlane-op/s  %5.2f cycles/wave-instr/SIMD (at 2.4GHz)\n", name, ms, tops, cyc);
    hipFree(d);
    return tops;
}

int main() {
    run<3>("v_add_u32", 1);
    run<0>("v_mad_u64_u32", 1);
    run<1>("v_mul_lo_u32", 1);
    run<2>("v_mul_hi_u32", 1);
    run<5>("v_mul_u32_u24", 1);
    run<7>("v_mad_u32_u24", 1);
    run<4>("v_lshl_add_u64", 1);
    run<6>("v_add_co+v_addc_co", 2);
    run<8>("v_cndmask_b32", 1);
    run<9>("v_cmp_lt_u64+v_cndmask", 2);
    run<10>("v_alignbit_b32", 1);
    run<11>("v_perm_b32", 1);
    run<12>("v_add3_u32", 1);
    run<13>("v_xor_b32", 1);
    run<14>("v_xad_u32", 1);
    run<15>("v_mov_b32", 1);
    run<16>("v_lshlrev_b64", 1);
    return 0;
}
