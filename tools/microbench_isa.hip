// Issue-rate microbenchmark for the gfx950 integer VALU forms the Goldilocks NTT arithmetic can be written in
// (round 2: which encodings / carry forms are cheap, what an s_nop hazard slot costs).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_isa.hip -o tools/microbench_isa.bin ; run on the GPU box.
// Output: per test the cycles per wave-instruction per SIMD, from the wall time at a nominal 2.4 GHz AND from
// s_memtime ticks measured inside the kernel (independent of the clock the chip actually ran at).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define ITERS 2048

// 8 independent registers r0..r7 (32-bit), two operands a, b; the block is repeated twice per loop iteration
#define R8(OPSTR) OPSTR(%0) OPSTR(%1) OPSTR(%2) OPSTR(%3) OPSTR(%4) OPSTR(%5) OPSTR(%6) OPSTR(%7)

#define DEF_KERNEL32(NAME, BODY)                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint64_t *ticks, uint32_t seed) {          \
        extern __shared__ uint32_t dyn[];                                                                 \
        uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, \
                 r7 = r0 * 19;                                                                            \
        uint32_t a = seed * 31 + threadIdx.x, b = (seed >> 3) | 1;                                        \
        if (seed == 0xdeadbeef) dyn[threadIdx.x] = a;                                                     \
        const uint64_t t0 = __builtin_readcyclecounter();                                                 \
        for (int it = 0; it < ITERS; it++) {                                                              \
            asm volatile(BODY BODY                                                                        \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                         : "v"(a), "v"(b)                                                                 \
                         : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"); \
        }                                                                                                 \
        const uint64_t t1 = __builtin_readcyclecounter();                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;               \
        if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;       \
    }

#define I1(x) "v_add_u32 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_add_u32, R8(I1))
#define I2(x) "v_add_co_u32 " #x ", vcc, " #x ", %8\n"
DEF_KERNEL32(k_add_co_e32, R8(I2))
#define I3(x) "v_addc_co_u32 " #x ", vcc, " #x ", %8, vcc\n"
DEF_KERNEL32(k_addc_co_e32_nonop, R8(I3))
#define I4(x) "v_addc_co_u32 " #x ", vcc, " #x ", %8, vcc\n s_nop 1\n"
DEF_KERNEL32(k_addc_co_e32_nop1, R8(I4))
#define I5(x) "v_add_u32 " #x ", " #x ", %8\n s_nop 0\n"
DEF_KERNEL32(k_add_u32_nop0, R8(I5))
#define I6(x) "v_add_u32 " #x ", " #x ", %8\n s_nop 1\n"
DEF_KERNEL32(k_add_u32_nop1, R8(I6))
// e64 carries into distinct SGPR pairs (no dependence)
#define I7 "v_add_co_u32 %0, s[40:41], %0, %8\n v_add_co_u32 %1, s[42:43], %1, %8\n v_add_co_u32 %2, s[44:45], %2, %8\n v_add_co_u32 %3, s[46:47], %3, %8\n" \
           "v_add_co_u32 %4, s[48:49], %4, %8\n v_add_co_u32 %5, s[50:51], %5, %8\n v_add_co_u32 %6, s[52:53], %6, %8\n v_add_co_u32 %7, s[54:55], %7, %8\n"
DEF_KERNEL32(k_add_co_e64, I7)
// three-limb add chains, four chains interleaved through distinct SGPR pairs: (r0,r1) += (a,b), (r2,r3) += ..., 2 wait states apart
#define I8 "v_add_co_u32 %0, s[40:41], %0, %8\n v_add_co_u32 %2, s[42:43], %2, %8\n v_add_co_u32 %4, s[44:45], %4, %8\n v_add_co_u32 %6, s[46:47], %6, %8\n" \
           "v_addc_co_u32 %1, s[40:41], %1, %9, s[40:41]\n v_addc_co_u32 %3, s[42:43], %3, %9, s[42:43]\n v_addc_co_u32 %5, s[44:45], %5, %9, s[44:45]\n v_addc_co_u32 %7, s[46:47], %7, %9, s[46:47]\n"
DEF_KERNEL32(k_addpair_e64_x4, I8)
// the same pair through vcc with the two hazard slots as s_nop
#define I9 "v_add_co_u32 %0, vcc, %0, %8\n s_nop 1\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_add_co_u32 %2, vcc, %2, %8\n s_nop 1\n v_addc_co_u32 %3, vcc, %3, %9, vcc\n" \
           "v_add_co_u32 %4, vcc, %4, %8\n s_nop 1\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_add_co_u32 %6, vcc, %6, %8\n s_nop 1\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n"
DEF_KERNEL32(k_addpair_vcc_nop, I9)
// the same pair through vcc with NO nops (is the hazard interlocked in hardware? result correctness is not checked here)
#define I10 "v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %9, vcc\n" \
            "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n"
DEF_KERNEL32(k_addpair_vcc_raw, I10)
#define I11(x) "v_lshlrev_b32 " #x ", 3, " #x "\n"
DEF_KERNEL32(k_lshlrev_e32, R8(I11))
#define I12(x) "v_ashrrev_i32 " #x ", 3, " #x "\n"
DEF_KERNEL32(k_ashrrev_e32, R8(I12))
#define I13(x) "v_and_b32 " #x ", %8, " #x "\n"
DEF_KERNEL32(k_and_e32, R8(I13))
#define I14(x) "v_and_b32 " #x ", 0x1fffff, " #x "\n"
DEF_KERNEL32(k_and_literal, R8(I14))
#define I15(x) "v_alignbit_b32 " #x ", " #x ", %8, 7\n"
DEF_KERNEL32(k_alignbit, R8(I15))
#define I16(x) "v_lshl_add_u32 " #x ", " #x ", 5, %8\n"
DEF_KERNEL32(k_lshl_add_u32, R8(I16))
#define I17(x) "v_lshl_or_b32 " #x ", " #x ", 5, %8\n"
DEF_KERNEL32(k_lshl_or_b32, R8(I17))
#define I18(x) "v_and_or_b32 " #x ", " #x ", %9, %8\n"
DEF_KERNEL32(k_and_or_b32, R8(I18))
#define I19(x) "v_add3_u32 " #x ", " #x ", %9, %8\n"
DEF_KERNEL32(k_add3_u32, R8(I19))
#define I20(x) "v_bfe_i32 " #x ", " #x ", 3, 21\n"
DEF_KERNEL32(k_bfe_i32, R8(I20))
#define I21(x) "v_bfe_u32 " #x ", " #x ", 3, 21\n"
DEF_KERNEL32(k_bfe_u32, R8(I21))
#define I22(x) "v_perm_b32 " #x ", " #x ", %8, %9\n"
DEF_KERNEL32(k_perm_b32, R8(I22))
#define I23(x) "v_add_u32_e64 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_add_u32_e64, R8(I23))
#define I24(x) "v_sub_u32 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_sub_u32, R8(I24))
#define I25(x) "v_subrev_u32 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_subrev_u32, R8(I25))
#define I26(x) "v_cndmask_b32 " #x ", " #x ", %8, vcc\n"
DEF_KERNEL32(k_cndmask_e32, R8(I26))
#define I27(x) "v_cndmask_b32 " #x ", " #x ", %8, s[40:41]\n"
DEF_KERNEL32(k_cndmask_e64, R8(I27))
#define I28(x) "v_cmp_lt_u32 vcc, " #x ", %8\n"
DEF_KERNEL32(k_cmp_e32, R8(I28))
#define I29(x) "v_mul_lo_u32 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_mul_lo_u32, R8(I29))
#define I30(x) "v_mul_hi_u32 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_mul_hi_u32, R8(I30))
#define I31(x) "v_mul_u32_u24 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_mul_u32_u24, R8(I31))
#define I32(x) "v_mul_hi_u32_u24 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_mul_hi_u32_u24, R8(I32))
#define I33(x) "v_mad_u32_u24 " #x ", " #x ", %8, %9\n"
DEF_KERNEL32(k_mad_u32_u24, R8(I33))
#define I34(x) "v_xor_b32 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_xor_b32, R8(I34))
#define I35(x) "v_mov_b32 " #x ", %8\n"
DEF_KERNEL32(k_mov_b32, R8(I35))
#define I36(x) "v_pk_add_u16 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_pk_add_u16, R8(I36))
#define I37(x) "v_pk_mul_lo_u16 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_pk_mul_lo_u16, R8(I37))
#define I38(x) "v_fma_f32 " #x ", " #x ", %8, %9\n"
DEF_KERNEL32(k_fma_f32, R8(I38))
#define I39(x) "v_mov_b32_dpp " #x ", " #x " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
DEF_KERNEL32(k_mov_dpp_quad, R8(I39))
#define I40(x) "v_mov_b32_dpp " #x ", " #x " row_ror:4 row_mask:0xf bank_mask:0xf\n"
DEF_KERNEL32(k_mov_dpp_ror, R8(I40))
#define I41(x) "v_add_u32_dpp " #x ", " #x ", %8 row_ror:4 row_mask:0xf bank_mask:0xf\n"
DEF_KERNEL32(k_add_dpp_ror, R8(I41))
#define I42 "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n" \
            "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
DEF_KERNEL32(k_permlane32_swap, I42)
#define I43 "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n" \
            "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
DEF_KERNEL32(k_permlane16_swap, I43)
#define I44(x) "ds_bpermute_b32 " #x ", %8, " #x "\n"
DEF_KERNEL32(k_ds_bpermute, R8(I44) "s_waitcnt lgkmcnt(0)\n")
#define I45(x) "v_sub_co_u32 " #x ", vcc, " #x ", %8\n"
DEF_KERNEL32(k_sub_co_e32, R8(I45))
#define I46(x) "v_mad_u32_u24 " #x ", %8, %9, " #x "\n"
DEF_KERNEL32(k_mad_u32_u24_acc, R8(I46))
#define I47(x) "v_xad_u32 " #x ", " #x ", %9, %8\n"
DEF_KERNEL32(k_xad_u32, R8(I47))
#define I48(x) "v_bfi_b32 " #x ", %9, " #x ", %8\n"
DEF_KERNEL32(k_bfi_b32, R8(I48))
#define I49(x) "v_mul_i32_i24 " #x ", " #x ", %8\n"
DEF_KERNEL32(k_mul_i32_i24, R8(I49))
#define I50(x) "v_dot2_u32_u16 " #x ", %8, %9, " #x "\n"
DEF_KERNEL32(k_dot2_u32_u16, R8(I50))
#define I51(x) "v_dot4_u32_u8 " #x ", %8, %9, " #x "\n"
DEF_KERNEL32(k_dot4_u32_u8, R8(I51))

// 64-bit register forms: 4 independent pairs
#define DEF_KERNEL64(NAME, BODY)                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint64_t *ticks, uint32_t seed) {          \
        extern __shared__ uint32_t dyn[];                                                                 \
        uint64_t r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, \
                 r7 = r0 * 19;                                                                            \
        uint32_t a = seed * 31 + threadIdx.x, b = (seed >> 3) | 1;                                        \
        uint64_t c = ((uint64_t)a << 32) | b;                                                             \
        if (seed == 0xdeadbeef) dyn[threadIdx.x] = a;                                                     \
        const uint64_t t0 = __builtin_readcyclecounter();                                                 \
        for (int it = 0; it < ITERS; it++) {                                                              \
            asm volatile(BODY BODY                                                                        \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                         : "v"(a), "v"(b), "v"(c)                                                         \
                         : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");                        \
        }                                                                                                 \
        const uint64_t t1 = __builtin_readcyclecounter();                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) ^ (uint32_t)((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) >> 32); \
        if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;       \
    }
#define J1(x) "v_mad_u64_u32 " #x ", vcc, %8, %9, " #x "\n"
DEF_KERNEL64(k_mad_u64_u32_vcc, R8(J1))
#define J2(x) "v_mad_u64_u32 " #x ", s[40:41], %8, %9, " #x "\n"
DEF_KERNEL64(k_mad_u64_u32_sgpr, R8(J2))
#define J3(x) "v_mad_u64_u32 " #x ", s[40:41], %8, %9, 0\n"
DEF_KERNEL64(k_mad_u64_u32_zero, R8(J3))
#define J4(x) "v_mad_i64_i32 " #x ", s[40:41], %8, %9, " #x "\n"
DEF_KERNEL64(k_mad_i64_i32, R8(J4))
#define J5(x) "v_lshl_add_u64 " #x ", " #x ", 0, %10\n"
DEF_KERNEL64(k_lshl_add_u64, R8(J5))
#define J6(x) "v_lshlrev_b64 " #x ", 3, " #x "\n"
DEF_KERNEL64(k_lshlrev_b64, R8(J6))
#define J7(x) "v_add_f64 " #x ", " #x ", %10\n"
DEF_KERNEL64(k_add_f64, R8(J7))
#define J8(x) "v_fma_f64 " #x ", " #x ", %10, %10\n"
DEF_KERNEL64(k_fma_f64, R8(J8))
#define J9(x) "v_pk_add_f32 " #x ", " #x ", %10\n"
DEF_KERNEL64(k_pk_add_f32, R8(J9))
#define J10(x) "v_pk_fma_f32 " #x ", " #x ", %10, %10\n"
DEF_KERNEL64(k_pk_fma_f32, R8(J10))
#define J11(x) "v_mov_b64 " #x ", %10\n"
DEF_KERNEL64(k_mov_b64, R8(J11))
#define J12(x) "v_cmp_lt_u64 vcc, " #x ", %10\n"
DEF_KERNEL64(k_cmp_lt_u64, R8(J12))
#define J13(x) "v_ashrrev_i64 " #x ", 3, " #x "\n"
DEF_KERNEL64(k_ashrrev_i64, R8(J13))
#define J14(x) "v_pk_mul_f32 " #x ", " #x ", %10\n"
DEF_KERNEL64(k_pk_mul_f32, R8(J14))

typedef void (*kern_t)(uint32_t *, uint64_t *, uint32_t);

static void run(const char *name, kern_t k, int instr_per_block, int lds_bytes, const char *note) {
    const int blocks = 256 * 8, threads = 256;
    uint32_t *d;
    uint64_t *t;
    hipMalloc(&d, (size_t)blocks * threads * 4);
    hipMalloc(&t, (size_t)blocks * 4 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    if (lds_bytes > 65536) hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_bytes, 0, d, t, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_bytes, 0, d, t, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    static uint64_t h[256 * 8 * 4];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double avg_ticks = 0;
    for (int i = 0; i < blocks * 4; i++) avg_ticks += (double)h[i];
    avg_ticks /= blocks * 4;
    const double wave_instr_total = (double)blocks * 4 * ITERS * 2.0 * instr_per_block;   // the block is repeated twice per iteration
    const double cyc_wall = (ms * 1e-3) * 2.4e9 * 1024 / wave_instr_total;
    // resident waves per SIMD: limited by LDS (160 KiB per CU) or 8
    int wg_per_cu = lds_bytes ? (160 * 1024) / lds_bytes : 8;
    if (wg_per_cu > 8) wg_per_cu = 8;
    const double waves_per_simd = wg_per_cu;   // 4 waves per workgroup over 4 SIMDs
    // s_memtime runs at a constant 100 MHz on this part: report wave-resident time in ns instead of guessing a ratio
    const double per_instr_ticks = avg_ticks / (ITERS * 2.0 * instr_per_block) / waves_per_simd;
    printf("%-26s %8.3f ms  %6.2f cyc/instr/SIMD @2.4GHz   memtime %7.4f ticks/instr/SIMD  (%d waves/SIMD) %s\n", name, ms, cyc_wall, per_instr_ticks,
           (int)waves_per_simd, note);
    hipFree(d);
    hipFree(t);
}

int main(int argc, char **argv) {
    int lds = 0;
    if (argc > 1) lds = atoi(argv[1]);   // e.g. 40960 -> 4 workgroups per CU = 4 waves per SIMD
#define RUN(k, n, note) run(#k, k, n, lds, note)
    RUN(k_add_u32, 8, "VOP2");
    RUN(k_add_u32_e64, 8, "same op, VOP3 encoding");
    RUN(k_sub_u32, 8, "VOP2");
    RUN(k_subrev_u32, 8, "VOP2");
    RUN(k_xor_b32, 8, "VOP2");
    RUN(k_mov_b32, 8, "VOP1");
    RUN(k_add_u32_nop0, 8, "v_add_u32 + s_nop 0 (VALU count only)");
    RUN(k_add_u32_nop1, 8, "v_add_u32 + s_nop 1 (VALU count only)");
    RUN(k_add_co_e32, 8, "VOP2, writes vcc");
    RUN(k_sub_co_e32, 8, "VOP2, writes vcc");
    RUN(k_add_co_e64, 8, "VOP3, writes distinct sgpr pairs");
    RUN(k_addc_co_e32_nonop, 8, "vcc -> vcc back to back, no nop");
    RUN(k_addc_co_e32_nop1, 8, "vcc -> vcc with s_nop 1");
    RUN(k_addpair_vcc_raw, 8, "add_co;addc_co via vcc, no nop");
    RUN(k_addpair_vcc_nop, 8, "add_co;s_nop 1;addc_co via vcc");
    RUN(k_addpair_e64_x4, 8, "4 interleaved pairs via sgpr carries");
    RUN(k_lshlrev_e32, 8, "VOP2");
    RUN(k_ashrrev_e32, 8, "VOP2");
    RUN(k_and_e32, 8, "VOP2");
    RUN(k_and_literal, 8, "VOP2 + 32-bit literal");
    RUN(k_cndmask_e32, 8, "VOP2 reads vcc");
    RUN(k_cndmask_e64, 8, "VOP3 reads sgpr pair");
    RUN(k_cmp_e32, 8, "VOPC writes vcc");
    RUN(k_alignbit, 8, "VOP3");
    RUN(k_lshl_add_u32, 8, "VOP3");
    RUN(k_lshl_or_b32, 8, "VOP3");
    RUN(k_and_or_b32, 8, "VOP3");
    RUN(k_add3_u32, 8, "VOP3");
    RUN(k_xad_u32, 8, "VOP3");
    RUN(k_bfi_b32, 8, "VOP3");
    RUN(k_bfe_i32, 8, "VOP3");
    RUN(k_bfe_u32, 8, "VOP3");
    RUN(k_perm_b32, 8, "VOP3");
    RUN(k_mul_lo_u32, 8, "VOP3");
    RUN(k_mul_hi_u32, 8, "VOP3");
    RUN(k_mul_u32_u24, 8, "VOP2");
    RUN(k_mul_i32_i24, 8, "VOP2");
    RUN(k_mul_hi_u32_u24, 8, "VOP2");
    RUN(k_mad_u32_u24, 8, "VOP3");
    RUN(k_mad_u32_u24_acc, 8, "VOP3 accumulate");
    RUN(k_dot2_u32_u16, 8, "VOP3P");
    RUN(k_dot4_u32_u8, 8, "VOP3P");
    RUN(k_pk_add_u16, 8, "VOP3P");
    RUN(k_pk_mul_lo_u16, 8, "VOP3P");
    RUN(k_fma_f32, 8, "VOP3 (fp reference)");
    RUN(k_mov_dpp_quad, 8, "DPP");
    RUN(k_mov_dpp_ror, 8, "DPP");
    RUN(k_add_dpp_ror, 8, "DPP");
    RUN(k_permlane32_swap, 8, "VOP1");
    RUN(k_permlane16_swap, 8, "VOP1");
    RUN(k_ds_bpermute, 8, "LDS crossbar");
    RUN(k_mad_u64_u32_vcc, 8, "VOP3 64-bit acc");
    RUN(k_mad_u64_u32_sgpr, 8, "VOP3 64-bit acc");
    RUN(k_mad_u64_u32_zero, 8, "VOP3 no addend");
    RUN(k_mad_i64_i32, 8, "VOP3");
    RUN(k_lshl_add_u64, 8, "VOP3 64-bit add");
    RUN(k_lshlrev_b64, 8, "VOP3 64-bit shift");
    RUN(k_ashrrev_i64, 8, "VOP3 64-bit shift");
    RUN(k_mov_b64, 8, "VOP1 64-bit");
    RUN(k_cmp_lt_u64, 8, "VOPC 64-bit");
    RUN(k_add_f64, 8, "fp64");
    RUN(k_fma_f64, 8, "fp64");
    RUN(k_pk_add_f32, 8, "packed fp32");
    RUN(k_pk_mul_f32, 8, "packed fp32");
    RUN(k_pk_fma_f32, 8, "packed fp32");
    return 0;
}
