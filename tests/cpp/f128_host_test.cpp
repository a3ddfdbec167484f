// Host-side exactness check of winterfell_amd/csrc/f128.cuh: mul (two folds of the 256-bit product) and mul_tab (the table form
// a_lo w + a_hi (w 2^64), one fold) against an independent double-and-add over p = 2^128 - 45 * 2^40 + 1
// (math/src/field/f128/mod.rs:429-466 is the reference's multiplication; any exact one gives the same canonical value).
//   /opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -I tests/cpp/stub tests/cpp/f128_host_test.cpp -o /tmp/f128_host_test && /tmp/f128_host_test
#include <stdio.h>
#include <stdlib.h>

#include "../../winterfell_amd/csrc/f128.cuh"

typedef unsigned __int128 u128;
static const u128 P = f128::modulus();

static u128 addmod(u128 a, u128 b) {        // a, b < p
    const u128 s = a + b;                   // may wrap 2^128
    if (s < a) return s - P;                // wrapped: s + 2^128 - p
    return s >= P ? s - P : s;
}
static u128 mulmod_ref(u128 a, u128 b) {    // double-and-add, most significant bit first
    u128 r = 0;
    for (int i = 127; i >= 0; i--) {
        r = addmod(r, r);
        if ((b >> i) & 1) r = addmod(r, a);
    }
    return r;
}
static uint64_t st = 0x243F6A8885A308D3ull;
static uint64_t rnd64() {
    uint64_t z = (st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static u128 rnd() { return (((u128)rnd64() << 64) | rnd64()) % P; }

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 200000;
    const u128 two64 = (u128)1 << 64;
    const u128 edge[] = {0, 1, 2, P - 1, P - 2, two64, two64 - 1, two64 + 1, (u128)45 << 40, ((u128)45 << 40) - 1, P >> 1, (P >> 1) + 1,
                         ~(u128)0 % P, ((u128)0xFFFFFFFFFFFFFFFFull << 64) % P, (u128)0xFFFFFFFFFFFFFFFFull};
    const int ne = sizeof(edge) / sizeof(edge[0]);
    int bad = 0;
    for (int i = 0; i < n + ne * ne && bad < 5; i++) {
        const u128 a = i < ne * ne ? edge[i / ne] : rnd(), w = i < ne * ne ? edge[i % ne] : rnd();
        const u128 want = mulmod_ref(a, w);
        const u128 w64 = mulmod_ref(w, two64);
        const u128 got = f128::mul(a, w), got_tab = f128::mul_tab(a, w, w64);
        if (got != want || got_tab != want) {
            printf("FAIL at %d: mul %s, mul_tab %s\n", i, got == want ? "ok" : "WRONG", got_tab == want ? "ok" : "WRONG");
            bad++;
        }
    }
    printf(bad ? "FAILED\n" : "%d products: mul and mul_tab ok\n", n + ne * ne);
    return bad != 0;
}
