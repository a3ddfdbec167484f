// Host-side exactness check of winterfell_amd/csrc/l24.cuh (the carry-free limb arithmetic of the f64 NTT passes):
// compiled with g++ (no GPU needed), compared against direct big-integer DFTs over p = 2^64 - 2^32 + 1.
//   g++ -O2 -std=c++17 tests/cpp/l24_host_test.cpp -o /tmp/l24_host_test && /tmp/l24_host_test
#include <stdio.h>
#include <stdlib.h>

#include "../../winterfell_amd/csrc/l24.cuh"

typedef unsigned __int128 u128;
static const uint64_t P = l24::P;

static uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % P); }
static uint64_t addmod(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + b) % P); }
static uint64_t powmod(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mulmod(r, a);
        a = mulmod(a, a);
        e >>= 1;
    }
    return r;
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int brev(int i, int bits) {
    int r = 0;
    for (int k = 0; k < bits; k++) r |= ((i >> k) & 1) << (bits - 1 - k);
    return r;
}

static int failures = 0;

template <int LOGN>
static void check(int rounds) {
    typedef l24::Dft<LOGN> D;
    constexpr int N = D::N;
    constexpr auto pl = l24::PlanHolder<LOGN>::value;
    int nhs = 0;
    for (int i = 0; i < pl.nops; i++) nhs += pl.ops[i].kind >= l24::OP_HS_ADD;
    printf("N = %2d: %3d operations (%d half-limb shifts), max limb magnitude 2^%.2f\n", N, pl.nops, nhs, __builtin_log2((double)pl.max_mag));
    const uint64_t wN = powmod(1ull << 12, 16 / N);   // omega_N = omega_16^(16/N), omega_16 = 2^12
    for (int r = 0; r < rounds; r++) {
        uint64_t x[N];
        for (int j = 0; j < N; j++) {
            switch (r % 7) {
                case 0: x[j] = rnd(); break;                      // any 64-bit word (lazy inputs included)
                case 1: x[j] = P - 1; break;
                case 2: x[j] = ~0ull; break;
                case 3: x[j] = 0; break;
                case 4: x[j] = (j & 1) ? P - 1 : ~0ull; break;
                case 5: x[j] = rnd() % P; break;
                default: x[j] = (rnd() & 1) ? 0xffffffff00000000ull : 0x00000000ffffffffull; break;
            }
        }
        int32_t v[D::NV];
        for (int e = 0; e < N; e++) D::load(v, e, x[e]);
        D::run(v);
        const uint64_t w = r % 3 == 0 ? 1 : rnd() % P;
        uint64_t W[4];
        for (int k = 0; k < 4; k++) W[k] = mulmod(w, powmod(1ull << 24, k));
        for (int k = 0; k < N; k++) {
            uint64_t want = 0;
            for (int j = 0; j < N; j++) want = addmod(want, mulmod(x[j] % P, powmod(wN, (uint64_t)j * k)));
            const int e = brev(k, LOGN);
            uint32_t y[4];
            for (int q = 0; q < 4; q++) {
                y[q] = D::limb(v, e, q);
                if (y[q] == 0 || y[q] >= (1u << 30)) {
                    printf("limb out of range\n");
                    failures++;
                }
            }
            const uint64_t got1 = l24::fold(l24::mul4_one(y));
            const uint64_t gotw = l24::fold(l24::mul4(y, W[0], W[1], W[2], W[3]));
            const uint64_t lazy = l24::fold_lazy(l24::mul4(y, W[0], W[1], W[2], W[3]));
            if (got1 != want || gotw != mulmod(want, w) || lazy % P != mulmod(want, w)) {
                if (failures < 10)
                    printf("MISMATCH N=%d round %d k=%d: got %llx / %llx want %llx\n", N, r, k, (unsigned long long)got1, (unsigned long long)gotw,
                           (unsigned long long)want);
                failures++;
            }
        }
    }
}

// fold() on extreme accumulator values (every carry path)
static void check_fold() {
    const uint64_t ext[] = {0, 1, 0xffffffffull, 0x100000000ull, P - 1, P, P + 1, ~0ull, ~0ull - 0xffffffffull, 0xfffffffeffffffffull, 0xffffffff00000000ull};
    const int ne = sizeof(ext) / sizeof(ext[0]);
    for (int a = 0; a < ne; a++)
        for (int b = 0; b < ne; b++) {
            l24::LM x{ext[a], ext[b]};
            if ((uint32_t)(x.m >> 32) > 0xfffffffbu) continue;   // M never reaches this range (l24.cuh)
            const uint64_t want = (uint64_t)(((u128)(x.l % P) + (u128)(x.m % P) * (1ull << 32)) % P);
            const uint64_t got = l24::fold(x);
            if (got != want || l24::fold_lazy(x) % P != want) {
                if (failures < 10)
                    printf("fold mismatch l=%llx m=%llx got %llx want %llx\n", (unsigned long long)x.l, (unsigned long long)x.m, (unsigned long long)got,
                           (unsigned long long)want);
                failures++;
            }
        }
    for (int i = 0; i < 200000; i++) {
        l24::LM x{rnd(), rnd()};
        if ((i & 3) == 0) x.l |= 0xffffffff00000000ull;
        if ((i & 7) == 1) x.m |= 0xfffffff000000000ull;
        if ((uint32_t)(x.m >> 32) > 0xfffffffbu) x.m &= 0xfffffffbffffffffull;
        const uint64_t want = (uint64_t)(((u128)(x.l % P) + (u128)(x.m % P) * (1ull << 32)) % P);
        if (l24::fold(x) != want) failures++;
    }
}

int main() {
    u128 bias = 0;
    for (int k = 3; k >= 0; k--) bias = ((bias << 24) + l24::BIAS[k]);
    if ((uint64_t)(bias % P) != 0) {
        printf("bias does not represent 0\n");
        failures++;
    }
    check_fold();
    check<1>(300);
    check<2>(300);
    check<3>(300);
    check<4>(700);
    printf(failures ? "FAILED (%d)\n" : "ok\n", failures);
    return failures != 0;
}
