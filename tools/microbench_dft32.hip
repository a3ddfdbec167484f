// The 32-point limb DFT that round 6 decided NOT to build, compiled for real ("build the one lever left or close the
// file"): the planner of csrc/l24.cuh generalised to omega_N = 2^(192 / N) — sub-limb shifts by 6, 12 or 18 bits for N = 32 — and a
// kernel per size that runs  split -> DFT -> bias -> exit (mul4_one + fold)  on N elements per lane, i.e. one register step without its
// table twiddle.  `hipcc -S` gives the instruction count the compiler actually emits (tools/dft32_budget.py is the model of the same
// plan); on a GPU the binary times both steps per element.
//   F="--offload-arch=gfx950 -O3 -mllvm -pragma-unroll-threshold=4000000 -mllvm -unroll-threshold=4000000 -I winterfell_amd/csrc"
//   hipcc $F tools/microbench_dft32.hip -o tools/microbench_dft32.bin          (the thresholds: a 720-operation plan must unroll fully)
//   hipcc $F -S --cuda-device-only tools/microbench_dft32.hip -o /tmp/dft32.s && python tools/isa_count.py /tmp/dft32.s step
// Compiled (round 6, ROCm 7.2): 16-point step 543 VALU instructions per 16 elements, 92 VGPRs; 32-point step 1896 per 32 elements, 258 VGPRs
// (one wave per SIMD): 33.9 against 59.3 per element and step, i.e. 8.5 against 11.9 per butterfly layer — the sub-limb shifts of
// omega_32 = 2^6 (twelve of the sixteen first-layer twiddles) and the signed split cost more than one exit + one split per element save.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "l24.cuh"

namespace g24 {
using l24::Limb;
using l24::Op;
enum : uint8_t { OP_ADD, OP_SUB, OP_SH_ADD, OP_SH_SUB, OP_SH_LO, OP_SH_HI };     // SH: shift by `s` bits inside a limb (s in Op::b's upper byte)
struct GOp {
    uint8_t kind, s;
    uint16_t a, b;
};
template <int LOGN>
struct GPlan {
    static constexpr int N = 1 << LOGN, NIN = 4 * N, MAX_OPS = 8 * N * LOGN + 4 * N + 1;
    int nops = 0;
    GOp ops[MAX_OPS] = {};
    Limb out[N][4] = {};
    uint64_t max_mag = 0;
};
// SIGNED input split: limbs 0, 1 in [-2^23, 2^23), limb 2 in [-2^15, 2^15], limb 3 in {0, 1}: 32 of them still fit the exit's bias
template <int LOGN, bool SIGNED>
constexpr GPlan<LOGN> make_plan() {
    GPlan<LOGN> p{};
    constexpr int N = 1 << LOGN, UNIT = 192 / N;
    Limb st[N][4] = {};
    for (int e = 0; e < N; e++)
        for (int k = 0; k < 4; k++) {
            st[e][k].id = (uint16_t)(e * 4 + k);
            st[e][k].neg = false;
            st[e][k].zero = !SIGNED && k == 3;
            st[e][k].mag = SIGNED ? (k < 2 ? (1ull << 23) : (k == 2 ? (1ull << 15) + 1 : 1)) : (k < 2 ? 0xffffffull : (k == 2 ? 0xffffull : 0));
        }
    auto emit = [&](uint8_t kind, uint8_t s, uint16_t a, uint16_t b) -> uint16_t {
        p.ops[p.nops] = GOp{kind, s, a, b};
        return (uint16_t)(GPlan<LOGN>::NIN + p.nops++);
    };
    for (int s = 0; s < LOGN; s++) {
        const int half = N >> (s + 1);
        for (int blk = 0; blk < N; blk += 2 * half)
            for (int i = 0; i < half; i++) {
                const int e1 = blk + i, e2 = blk + i + half;
                for (int k = 0; k < 4; k++) {
                    const Limb A = st[e1][k], B = st[e2][k];
                    Limb S{}, D{};
                    if (A.zero && B.zero) S.zero = D.zero = true;
                    else if (B.zero) { S = A; D = A; }
                    else if (A.zero) { S = B; D = B; D.neg = !B.neg; }
                    else {
                        const bool same = A.neg == B.neg;
                        S.id = emit(same ? OP_ADD : OP_SUB, 0, A.id, B.id);
                        D.id = emit(same ? OP_SUB : OP_ADD, 0, A.id, B.id);
                        S.neg = D.neg = A.neg;
                        S.mag = D.mag = A.mag + B.mag;
                    }
                    st[e1][k] = S;
                    st[e2][k] = D;
                }
                const int bits = i * (N / (2 * half)) * UNIT, q = (bits / 24) % 4, sub = bits % 24;
                if (sub) {
                    Limb nw[4] = {};
                    for (int k = 0; k < 4; k++) {
                        const Limb L = st[e2][k], H = st[e2][(k + 3) % 4];
                        const bool hneg = H.neg != (k == 0);
                        Limb R{};
                        const uint64_t lo_mag = ((1ull << (24 - sub)) - 1) << sub;
                        if (L.zero && H.zero) R.zero = true;
                        else if (H.zero) { R.id = emit(OP_SH_LO, (uint8_t)sub, L.id, L.id); R.neg = L.neg; R.mag = lo_mag; }
                        else if (L.zero) { R.id = emit(OP_SH_HI, (uint8_t)sub, H.id, H.id); R.neg = hneg; R.mag = (H.mag >> (24 - sub)) + 1; }
                        else { R.id = emit(L.neg == hneg ? OP_SH_ADD : OP_SH_SUB, (uint8_t)sub, L.id, H.id); R.neg = L.neg; R.mag = lo_mag + (H.mag >> (24 - sub)) + 1; }
                        nw[k] = R;
                    }
                    for (int k = 0; k < 4; k++) st[e2][k] = nw[k];
                }
                if (q) {
                    Limb nw[4] = {};
                    for (int k = 0; k < 4; k++) {
                        nw[k] = st[e2][(k - q + 4) % 4];
                        if (k < q && !nw[k].zero) nw[k].neg = !nw[k].neg;
                    }
                    for (int k = 0; k < 4; k++) st[e2][k] = nw[k];
                }
            }
    }
    for (int e = 0; e < N; e++)
        for (int k = 0; k < 4; k++) {
            p.out[e][k] = st[e][k];
            if (!st[e][k].zero && st[e][k].mag > p.max_mag) p.max_mag = st[e][k].mag;
        }
    return p;
}
template <int LOGN, bool SIGNED>
struct Holder {
    static constexpr GPlan<LOGN> value = make_plan<LOGN, SIGNED>();
    static_assert(value.max_mag <= l24::MAX_MAG, "limb growth exceeds what the bias absorbs");
};
template <int LOGN, bool SIGNED>
struct Dft {
    static constexpr int N = 1 << LOGN, NIN = 4 * N, NV = NIN + Holder<LOGN, SIGNED>::value.nops;
    static __device__ __forceinline__ void load(int32_t (&v)[NV], int e, uint64_t x) {
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        if constexpr (!SIGNED) {
            v[4 * e + 0] = (int32_t)(lo & 0xffffffu);
            v[4 * e + 1] = (int32_t)(((lo >> 24) | (hi << 8)) & 0xffffffu);
            v[4 * e + 2] = (int32_t)(hi >> 16);
            v[4 * e + 3] = 0;
        } else {   // balanced digits: d = ((x + 2^23) & (2^24 - 1)) - 2^23, carry into the next limb
            const uint32_t l0 = lo & 0xffffffu, l1 = ((lo >> 24) | (hi << 8)) & 0xffffffu, l2 = hi >> 16;
            const uint32_t c0 = (l0 + 0x800000u) >> 24;
            const uint32_t t1 = l1 + c0, c1 = (t1 + 0x800000u) >> 24;
            const uint32_t t2 = l2 + c1, c2 = (t2 + 0x8000u) >> 16;
            v[4 * e + 0] = (int32_t)l0 - (int32_t)(c0 << 24);
            v[4 * e + 1] = (int32_t)t1 - (int32_t)(c1 << 24);
            v[4 * e + 2] = (int32_t)t2 - (int32_t)(c2 << 16);
            v[4 * e + 3] = (int32_t)c2;
        }
    }
    static __device__ __forceinline__ void run(int32_t (&v)[NV]) {
        constexpr GPlan<LOGN> pl = Holder<LOGN, SIGNED>::value;
#pragma unroll
        for (int i = 0; i < pl.nops; i++) {
            const int32_t a = v[pl.ops[i].a], b = v[pl.ops[i].b];
            const int s = pl.ops[i].s;
            const uint32_t m = (1u << (24 - (s ? s : 1))) - 1u;
            int32_t r;
            switch (pl.ops[i].kind) {
                case OP_ADD: r = (int32_t)((uint32_t)a + (uint32_t)b); break;
                case OP_SUB: r = (int32_t)((uint32_t)a - (uint32_t)b); break;
                case OP_SH_ADD: r = (int32_t)((((uint32_t)a & m) << s) + (uint32_t)(b >> (24 - s))); break;
                case OP_SH_SUB: r = (int32_t)((((uint32_t)a & m) << s) - (uint32_t)(b >> (24 - s))); break;
                case OP_SH_LO: r = (int32_t)(((uint32_t)a & m) << s); break;
                default: r = b >> (24 - s); break;
            }
            v[NIN + i] = r;
        }
    }
    static __device__ __forceinline__ uint32_t limb(const int32_t (&v)[NV], int e, int k) {
        constexpr GPlan<LOGN> pl = Holder<LOGN, SIGNED>::value;
        if (pl.out[e][k].zero) return l24::BIAS[k];
        return pl.out[e][k].neg ? l24::BIAS[k] - (uint32_t)v[pl.out[e][k].id] : l24::BIAS[k] + (uint32_t)v[pl.out[e][k].id];
    }
};
}  // namespace g24

// one register step on N = 2^LOGN elements per lane: split, DFT, exit (w = 1).  NOTE: with SIGNED digits a limb may keep low bits below
// its sub-limb shift's mask sign-extended; this kernel exists for instruction counts and timing, not for values.
template <int LOGN, bool SIGNED>
__global__ __launch_bounds__(256) void step_kernel(uint64_t *data, int reps) {
    typedef g24::Dft<LOGN, SIGNED> D;
    constexpr int N = 1 << LOGN;
    uint64_t x[N];
    uint64_t *p = data + ((size_t)blockIdx.x * 256 + threadIdx.x) * N;
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = p[i];
    for (int r = 0; r < reps; r++) {
        int32_t v[D::NV];
#pragma unroll
        for (int i = 0; i < N; i++) D::load(v, i, x[i]);
        D::run(v);
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint32_t y[4];
#pragma unroll
            for (int q = 0; q < 4; q++) y[q] = D::limb(v, i, q);
            x[i] = l24::fold_lazy(l24::mul4_one(y));
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = x[i];
}

template <int LOGN, bool SIGNED>
static double time_step(uint64_t *d, int blocks, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((step_kernel<LOGN, SIGNED>), dim3(blocks), dim3(256), 0, 0, d, reps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((step_kernel<LOGN, SIGNED>), dim3(blocks), dim3(256), 0, 0, d, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e9 / ((double)blocks * 256 * (1 << LOGN) * reps);     // ps per element and step (whole device)
}

int main() {
    const int blocks = 4096, reps = 64;
    uint64_t *d;
    hipMalloc(&d, (size_t)blocks * 256 * 32 * 8);
    hipMemset(d, 1, (size_t)blocks * 256 * 32 * 8);
    const double t16 = time_step<4, false>(d, blocks, reps), t32 = time_step<5, true>(d, blocks, reps);
    printf("register step (split + DFT + bias + exit), ps per element on the whole device: 16-point %.3f (%.3f per butterfly layer), 32-point signed %.3f (%.3f per layer)\n",
           t16, t16 / 4, t32, t32 / 5);
    printf("a 2^24 transform is 24 layers: 6 steps of 16 points = %.2f ps per element, 4 x 32 + 1 x 16 = %.2f ps per element (register steps only)\n",
           6 * t16, 4 * t32 + t16);
    return 0;
}
