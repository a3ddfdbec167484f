// What Prover::new_trace_lde (prover/src/lib.rs:182-190) costs a HOST caller, PCIe included: host column Vecs in, host
// TracePolyTable + commitment out, the LDE matrix, leaves and nodes left in HBM for the query phase.
//
//   serial     INTEGRATION.md section 2 as first written: upload every column, ONE wf_build_trace_commitment, download the polynomials
//              (and, for a host-side MerkleTree::from_raw_parts, leaves + nodes).  Each leg is timed on its own.
//   pipelined  wf::new_trace_lde_from_host (include/winterfell_hip.hpp): uploads, group-wise interpolation, polynomial downloads and the
//              LDE + commit overlap on three contexts / threads; leaves + nodes stay on the device.
// Both produce the same polynomials (compared word for word) and the same root.  Host buffers are page-locked with wf_host_register
// first (what a prover that reuses its buffers does; the registration time is printed separately) — pass pin=0 for pageable memory
// (the library then stages through its own page-locked bounce buffers).
//
//   g++ -O2 -std=c++17 -pthread -Iinclude tools/host_pipeline_bench.cpp -Lwinterfell_amd -lwinterfell_hip -Wl,-rpath,$PWD/winterfell_amd -o tools/host_pipeline_bench.bin
//   tools/host_pipeline_bench.bin [field: 0 f64 | 1 f128] [log_n=20] [cols=4] [partitions=1] [reps=5] [pin=1] [group=8]
// Prints one JSON object (last line).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "winterfell_hip.hpp"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv) {
    const int fid = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t log_n = argc > 2 ? atoi(argv[2]) : 20, cols = argc > 3 ? atoi(argv[3]) : 4, parts = argc > 4 ? atoi(argv[4]) : 1;
    const int reps = argc > 5 ? atoi(argv[5]) : 5, pin = argc > 6 ? atoi(argv[6]) : 1;
    const uint32_t group = argc > 7 ? atoi(argv[7]) : 8;
    const wf::Field f = fid == 1 ? wf::Field::F128 : wf::Field::F64;
    const uint32_t W = wf::words(f);
    const uint64_t n = 1ull << log_n, blowup = 8, N = n * blowup;
    const size_t col_words = n * W, col_bytes = col_words * 8;
    try {
        wf::Context ctx(0), up(0), down(0);
        // ---- host side: c column "Vecs" of uniform field elements (f64: Montgomery residues < p; f128: both words < 2^62 => canonical)
        std::vector<std::vector<uint64_t>> trace(cols, std::vector<uint64_t>(col_words)), polys_a(cols, std::vector<uint64_t>(col_words)),
            polys_b(cols, std::vector<uint64_t>(col_words));
        uint64_t x = 0x5EED0000 + cols;
        for (auto &col : trace)
            for (auto &w : col) {
                x += 0x9E3779B97F4A7C15ull;
                uint64_t z = x;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                z ^= z >> 31;
                w = f == wf::Field::F64 ? z % 0xffffffff00000001ull : z >> 2;
            }
        std::vector<uint8_t> h_leaves(N * 32), h_nodes(N * 32);
        double pin_ms = 0;
        if (pin) {
            const double t0 = now_ms();
            for (auto *set : {&trace, &polys_a, &polys_b})
                for (auto &col : *set) wf::check(wf_host_register(ctx.handle(), col.data(), col_bytes), "wf_host_register");
            wf::check(wf_host_register(ctx.handle(), h_leaves.data(), h_leaves.size()), "wf_host_register");
            wf::check(wf_host_register(ctx.handle(), h_nodes.data(), h_nodes.size()), "wf_host_register");
            pin_ms = now_ms() - t0;
        }
        const uint64_t offset[2] = {f == wf::Field::F64 ? (uint64_t)(((unsigned __int128)7 << 64) % 0xffffffff00000001ull) : 3ull, 0};
        const wf::PartitionOptions po{parts, 1};
        const wf::Hash h = wf::Hash::Blake3_256;
        std::vector<const uint64_t *> in;
        std::vector<uint64_t *> out_b;
        for (uint32_t k = 0; k < cols; k++) {
            in.push_back(trace[k].data());
            out_b.push_back(polys_b[k].data());
        }
        std::vector<double> t_h2d, t_kern, t_d2h_polys, t_d2h_tree, t_serial, t_pipe;
        std::vector<uint8_t> root_a, root_b;
        for (int r = 0; r < reps + 1; r++) {            // repetition 0 warms up (tables, pools, clocks) and is dropped
            // ---- serial
            {
                const double t0 = now_ms();
                wf::ColMatrix tr{wf::DeviceBuffer(ctx, (size_t)cols * col_bytes), f, cols, 1, n};
                for (uint32_t k = 0; k < cols; k++)
                    wf::check(wf_memcpy_h2d(ctx.handle(), (uint8_t *)tr.data.data() + (size_t)k * col_bytes, trace[k].data(), col_bytes), "h2d");
                const double t1 = now_ms();
                wf::RowMatrix lde;
                lde.field = f;
                lde.num_rows = N;
                lde.row_width = wf_row_width(cols, 1);
                lde.elements_per_row = cols;
                lde.data = wf::DeviceBuffer(ctx, N * lde.row_width * 8 * W);
                wf::DeviceBuffer leaves(ctx, N * 32), nodes(ctx, N * 32);
                uint8_t root[32];
                wf::check(wf_build_trace_commitment(ctx.handle(), (int)h, (int)f, 1, tr.data.data(), cols, n, log_n, 3, offset, po.num_partitions, po.hash_rate,
                                                    0, lde.data.data(), leaves.data(), nodes.data(), root), "wf_build_trace_commitment");
                ctx.sync();
                const double t2 = now_ms();
                for (uint32_t k = 0; k < cols; k++) tr.data.download(polys_a[k].data(), col_bytes, (size_t)k * col_bytes);
                const double t3 = now_ms();
                leaves.download(h_leaves.data(), h_leaves.size());
                nodes.download(h_nodes.data(), h_nodes.size());
                const double t4 = now_ms();
                root_a.assign(root, root + 32);
                if (r) {
                    t_h2d.push_back(t1 - t0);
                    t_kern.push_back(t2 - t1);
                    t_d2h_polys.push_back(t3 - t2);
                    t_d2h_tree.push_back(t4 - t3);
                    t_serial.push_back(t3 - t0);
                }
            }
            // ---- pipelined
            {
                const double t0 = now_ms();
                wf::TraceCommitment tc = wf::new_trace_lde_from_host(ctx, up, down, h, f, in, n, blowup, offset, out_b, po, 1, group);
                root_b = tc.tree.root();
                const double t1 = now_ms();
                if (r) t_pipe.push_back(t1 - t0);
            }
        }
        bool same = root_a == root_b;
        for (uint32_t k = 0; k < cols && same; k++) same = polys_a[k] == polys_b[k];
        if (pin) {
            for (auto *set : {&trace, &polys_a, &polys_b})
                for (auto &col : *set) wf_host_unregister(ctx.handle(), col.data());
            wf_host_unregister(ctx.handle(), h_leaves.data());
            wf_host_unregister(ctx.handle(), h_nodes.data());
        }
        const double trace_gb = (double)cols * col_bytes / 1e9, tree_gb = 2.0 * N * 32 / 1e9;
        const double h2d = median(t_h2d), kern = median(t_kern), d2hp = median(t_d2h_polys), d2ht = median(t_d2h_tree);
        const double larger = std::max(h2d, d2hp);
        printf("{\"shape\": \"2^%u x %u %s, blowup 8, Blake3_256, %u partition(s)\", \"pinned\": %s, \"host_register_ms\": %.2f, "
               "\"h2d_trace_ms\": %.3f, \"h2d_GBps\": %.1f, \"kernels_ms\": %.3f, \"d2h_polys_ms\": %.3f, \"d2h_polys_GBps\": %.1f, "
               "\"d2h_leaves_nodes_ms\": %.3f, \"d2h_leaves_nodes_GBps\": %.1f, \"serial_total_ms_polys_only\": %.3f, "
               "\"serial_total_ms_with_leaves_nodes\": %.3f, \"pipelined_total_ms\": %.3f, \"bound_kernels_plus_1p3_larger_leg_ms\": %.3f, "
               "\"pipelined_within_bound\": %s, \"same_polys_and_root\": %s, \"reps\": %d, \"group\": %u}\n",
               log_n, cols, fid == 1 ? "f128" : "f64", parts, pin ? "true" : "false", pin_ms, h2d, trace_gb / (h2d * 1e-3), kern, d2hp, trace_gb / (d2hp * 1e-3),
               d2ht, tree_gb / (d2ht * 1e-3), median(t_serial), median(t_serial) + d2ht, median(t_pipe), kern + 1.3 * larger,
               median(t_pipe) <= kern + 1.3 * larger ? "true" : "false", same ? "true" : "false", reps, group);
        return same ? 0 : 1;
    } catch (const std::exception &e) {
        fprintf(stderr, "host_pipeline_bench: %s\n", e.what());
        return 2;
    }
}
