// Standalone timing harness over the C ABI (no Python): per-kernel HIP-event times for the f64 NTT and the
// trace LDE + commit pipeline.  Build: see tools/Makefile.  Usage: ntt_bench [log_n] [reps]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/winterfell_hip.h"

#define CK(x) do { int _s = (x); if (_s) { printf("FAIL %s -> %d (%s)\n", #x, _s, wf_strerror(_s)); return 1; } } while (0)

int main(int argc, char **argv) {
    int log_n = argc > 1 ? atoi(argv[1]) : 24;
    int reps = argc > 2 ? atoi(argv[2]) : 20;
    wf_ctx *ctx;
    CK(wf_ctx_create(0, &ctx));
    const size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> h(n);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x % 0xffffffff00000001ull; }
    void *d;
    CK(wf_malloc(ctx, n * 8, &d));
    CK(wf_memcpy_h2d(ctx, d, h.data(), n * 8));
    CK(wf_fft_evaluate_poly(ctx, WF_FIELD_F64, 1, d, log_n, 1));
    CK(wf_fft_interpolate_poly(ctx, WF_FIELD_F64, 1, d, log_n, 1));
    CK(wf_ctx_sync(ctx));
    CK(wf_prof_enable(ctx, 1));
    for (int r = 0; r < reps; r++) CK(wf_fft_evaluate_poly(ctx, WF_FIELD_F64, 1, d, log_n, 1));
    char buf[4096];
    CK(wf_prof_collect(ctx, buf, sizeof buf));
    printf("== forward NTT 2^%d x %d reps (name launches total_ms) ==\n%s", log_n, reps, buf);
    double tot = 0; { char *p = buf; while (*p) { char nm[64]; unsigned long long c; double ms; if (sscanf(p, "%63s %llu %lf", nm, &c, &ms) == 3) tot += ms; p = strchr(p, '\n'); if (!p) break; p++; } }
    printf("per transform: %.2f us  -> %.1f GB/s algorithmic (%.1f%% of 8 TB/s)\n", tot * 1e3 / reps, 2.0 * n * 8 / (tot * 1e-3 / reps) / 1e9,
           2.0 * n * 8 / (tot * 1e-3 / reps) / 1e9 / 80.0);
    for (int r = 0; r < reps; r++) CK(wf_fft_interpolate_poly(ctx, WF_FIELD_F64, 1, d, log_n, 1));
    CK(wf_prof_collect(ctx, buf, sizeof buf));
    printf("== inverse NTT ==\n%s", buf);
    // round trip check
    std::vector<uint64_t> back(n);
    CK(wf_memcpy_d2h(ctx, back.data(), d, n * 8));
    printf("roundtrip %s\n", memcmp(back.data(), h.data(), n * 8) == 0 ? "OK" : "MISMATCH");
    // trace LDE + commit: 2^20 x 4, blowup 8
    {
        const uint32_t ln = 20, lb = 3, c = 4;
        const size_t tn = (size_t)1 << ln, N = tn << lb;
        void *tr, *lde, *leaves, *nodes;
        CK(wf_malloc(ctx, c * tn * 8, &tr));
        CK(wf_malloc(ctx, N * 8 * 8, &lde));
        CK(wf_malloc(ctx, N * 32, &leaves));
        CK(wf_malloc(ctx, N * 32, &nodes));
        CK(wf_memcpy_h2d(ctx, tr, h.data(), c * tn * 8));
        uint64_t off = 7ull * 0xffffffffull;  // new(7) = 7 * 2^64 mod p = 7 * (2^32 - 1)
        uint8_t root[32];
        for (int hash = 0; hash < 2; hash++) {
            CK(wf_prof_enable(ctx, 0));
            CK(wf_build_trace_commitment(ctx, hash, WF_FIELD_F64, 1, tr, c, tn, ln, lb, &off, 1, 1, 0, lde, leaves, nodes, root));
            CK(wf_prof_enable(ctx, 1));
            int r2 = hash == 0 ? 10 : 2;
            for (int r = 0; r < r2; r++)
                CK(wf_build_trace_commitment(ctx, hash, WF_FIELD_F64, 1, tr, c, tn, ln, lb, &off, 1, 1, 0, lde, leaves, nodes, root));
            CK(wf_prof_collect(ctx, buf, sizeof buf));
            printf("== trace LDE+commit 2^20 x 4, blowup 8, hash %d, %d reps ==\n%s", hash, r2, buf);
        }
    }
    wf_ctx_destroy(ctx);
    return 0;
}
