// The reference's `fib-small` example (examples/src/fibonacci/fib_small) end to end through the C++ host layer
// (include/winterfell_hip.hpp): the steps of Prover::generate_proof (prover/src/lib.rs:275-492) with every data-parallel step
// on the device and the Fiat-Shamir coin / channel as host logic.  The transcript is the reference's (coin seeded with
// Context::to_elements() ++ public inputs, same draw order) and the one winterfell_amd.prover.prove() produces, so for equal inputs both host layers print the same roots, nonce and
// query positions — tests/test_gpu_cpp_host.py checks exactly that.
//
//   g++ -O2 -std=c++17 -Iinclude examples/fib_small.cpp -Lwinterfell_amd -lwinterfell_hip -o examples/fib_small.bin
//   examples/fib_small.bin [log_n=16] [hash: 0 blake3_256 | 1 rp64_256 | 2 sha3_256] [ext_degree=2] [repeat=3]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>

#include "winterfell_hip.hpp"

typedef unsigned __int128 u128;
static const uint64_t P = 0xffffffff00000001ull;
static uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % P); }
static uint64_t powmod(uint64_t a, u128 e) {
    uint64_t r = 1;
    for (; e; e >>= 1, a = mulmod(a, a))
        if (e & 1) r = mulmod(r, a);
    return r;
}
static uint64_t to_mont(uint64_t canon) { return (uint64_t)(((u128)(canon % P) << 64) % P); }             // BaseElement::new
static uint64_t from_mont(uint64_t m) { static const uint64_t rinv = powmod(to_mont(1), (u128)P - 2); return mulmod(m, rinv); }
static uint64_t root_of_unity(uint32_t log_n) { return powmod(7277203076849721926ull, (u128)1 << (32 - log_n)); }   // f64/mod.rs:267

// crypto::DefaultRandomCoin (crypto/src/random/default.rs) over the library's hashers
struct Coin {
    wf::Context &ctx;
    wf::Hash h;
    uint8_t seed[32];
    uint64_t counter = 0;
    Coin(wf::Context &c, wf::Hash hash, const std::vector<uint64_t> &seed_elements) : ctx(c), h(hash) {
        hash_elements(seed_elements, seed);
    }
    void hash_elements(const std::vector<uint64_t> &e, uint8_t out[32]) {
        wf::DeviceBuffer d_in(ctx, e), d_out(ctx, 32);
        wf::check(wf_hash_elements_batch(ctx.handle(), (int)h, WF_FIELD_F64, d_in.data(), 1, e.size(), (uint32_t)e.size(), d_out.data()), "wf_hash_elements_batch");
        d_out.download(out, 32);
    }
    void reseed(const uint8_t data[32]) {                                      // seed = merge(seed, data), default.rs:150-153
        std::vector<uint8_t> two(64);
        std::memcpy(two.data(), seed, 32);
        std::memcpy(two.data() + 32, data, 32);
        wf::DeviceBuffer d_in(ctx, two), d_out(ctx, 32);
        wf::check(wf_hash_merge_batch(ctx.handle(), (int)h, d_in.data(), 1, d_out.data()), "wf_hash_merge_batch");
        d_out.download(seed, 32);
        counter = 0;
    }
    // the next `count` values of next() = merge_with_int(seed, ++counter) as Digest::as_bytes, one device batch
    std::vector<std::vector<uint8_t>> next(uint32_t count) {
        wf::DeviceBuffer d_out(ctx, (size_t)count * 32);
        wf::check(wf_hash_merge_with_int_batch(ctx.handle(), (int)h, seed, counter + 1, count, d_out.data()), "wf_hash_merge_with_int_batch");
        std::vector<uint8_t> raw = d_out.to_host<uint8_t>();
        counter += count;
        std::vector<std::vector<uint8_t>> out;
        for (uint32_t i = 0; i < count; i++) {
            std::vector<uint8_t> d(raw.begin() + 32 * i, raw.begin() + 32 * (i + 1));
            if (h == wf::Hash::Rp64_256)                                          // ElementDigest::as_bytes: canonical little-endian
                for (int k = 0; k < 4; k++) {
                    uint64_t w;
                    std::memcpy(&w, &d[8 * k], 8);
                    w = from_mont(w);
                    std::memcpy(&d[8 * k], &w, 8);
                }
            out.push_back(d);
        }
        return out;
    }
    // draw::<E> (default.rs:185-199) `count` times; the hashes of a run come in one batch, a rejected value costs one more
    std::vector<uint64_t> draw(uint32_t count, uint32_t D) {
        std::vector<uint64_t> out;
        std::vector<std::vector<uint8_t>> ahead = next(count);
        size_t pos = 0;
        for (uint32_t k = 0; k < count; k++) {
            for (int tries = 0;; tries++) {
                if (tries == 1000) throw std::runtime_error("FailedToDrawFieldElement(1000)");
                if (pos == ahead.size()) { ahead = next(1); pos = 0; }
                const std::vector<uint8_t> &b = ahead[pos++];
                uint64_t v[3];
                bool ok = true;
                for (uint32_t d = 0; d < D; d++) {
                    std::memcpy(&v[d], &b[8 * d], 8);
                    ok = ok && v[d] < P;
                }
                if (!ok) continue;
                for (uint32_t d = 0; d < D; d++) out.push_back(to_mont(v[d]));
                break;
            }
        }
        return out;
    }
};

struct Channel : wf::ProverChannel {                                           // fri::ProverChannel over the coin
    Coin &coin;
    uint32_t D;
    std::vector<std::vector<uint8_t>> commitments;
    Channel(Coin &c, uint32_t d) : coin(c), D(d) {}
    void commit_fri_layer(const uint8_t root[32]) override {
        commitments.emplace_back(root, root + 32);
        coin.reseed(root);
    }
    std::vector<uint64_t> draw_fri_alpha() override { return coin.draw(1, D); }
};

static std::string hex(const uint8_t *p, size_t n) {
    static const char *d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
    return s;
}

int main(int argc, char **argv) {
    const uint32_t log_n = argc > 1 ? atoi(argv[1]) : 16;
    const wf::Hash hash = (wf::Hash)(argc > 2 ? atoi(argv[2]) : 0);
    const uint32_t D = argc > 3 ? atoi(argv[3]) : 2;
    const int repeat = argc > 4 ? atoi(argv[4]) : 3;
    const uint64_t n = 1ull << log_n, blowup = 8, ce_blowup = 2, N = n * blowup;
    const uint32_t num_queries = 28, grinding = 16, folding = 8, rem_deg = 127;
    const wf::Field F = wf::Field::F64;
    // trace (prover.rs:30-50): row i = (f(2i), f(2i+1)), f(0) = f(1) = 1; sequential host work, as in the reference
    std::vector<uint64_t> trace(2 * n);
    uint64_t a = 1, b = 1;
    for (uint64_t i = 0; i < n; i++) {
        trace[i] = to_mont(a);
        trace[n + i] = to_mont(b);
        a = (uint64_t)(((u128)a + b) % P);
        b = (uint64_t)(((u128)a + b) % P);
    }
    const uint64_t result = trace[2 * n - 1], one = to_mont(1), offset = to_mont(7);
    wf::Context ctx(0);
    double best = 1e30;
    std::string summary;
    for (int rep = 0; rep < repeat; rep++) {
        wf::ColMatrix cm{wf::DeviceBuffer(ctx, trace), F, 2, 1, n};
        ctx.sync();
        const auto t0 = std::chrono::steady_clock::now();
        // coin seed = hash_elements(Context::to_elements() ++ PublicInputs::to_elements()), the reference's encoding
        // (prover/src/channel.rs:57-75; fib_small's public input is the result element, 3 assertions + 2 transition constraints)
        std::vector<uint64_t> seed_e;
        for (uint64_t v : wf::context_to_elements_f64(2, n, P, 5, D, folding, rem_deg, (uint32_t)blowup, grinding, num_queries)) seed_e.push_back(to_mont(v));
        seed_e.push_back(result);
        Coin coin(ctx, hash, seed_e);
        // 1. main trace commitment
        wf::TraceCommitment tc = wf::build_trace_commitment(hash, cm, blowup, &offset);
        const std::vector<uint8_t> trace_root = tc.tree.root();
        coin.reseed(trace_root.data());
        // 2. constraint evaluation
        const std::vector<uint64_t> cc = coin.draw(5, D);                       // 2 transition, then 3 boundary coefficients
        std::vector<uint64_t> cc_t(cc.begin(), cc.begin() + 2 * D), cc_b(cc.begin() + 2 * D, cc.end());
        std::vector<wf::Assertion> as{{0, 0, {one}}, {1, 0, {one}}, {1, n - 1, {result}}};
        wf::DeviceBuffer ev = wf::evaluate_constraints(WF_AIR_FIB_SMALL, tc.lde, n, blowup, ce_blowup, &offset, D, cc_t, as, cc_b);
        // 3. composition polynomial (1 column) and its commitment
        wf::fft::interpolate_poly_with_offset(ev, F, n * ce_blowup, &offset, D);
        wf::DeviceBuffer col(ctx, n * D * 8);
        wf::check(wf_memcpy_d2d(ctx.handle(), col.data(), ev.data(), n * D * 8), "wf_memcpy_d2d");
        wf::ColMatrix quot{std::move(col), F, 1, D, n};
        wf::TraceCommitment qc = wf::build_trace_commitment(hash, quot, blowup, &offset, {}, true);
        const std::vector<uint8_t> constraint_root = qc.tree.root();
        coin.reseed(constraint_root.data());
        // 4. out-of-domain frames, DEEP composition and its evaluation
        const std::vector<uint64_t> z = coin.draw(1, D);
        std::vector<uint64_t> pts(z);
        const uint64_t g = root_of_unity(log_n);
        for (uint32_t d = 0; d < D; d++) pts.push_back(to_mont(mulmod(from_mont(z[d]), g)));
        const std::vector<uint64_t> tf = wf::evaluate_columns_at(tc.polys, pts, 2, D);      // [point][column][D]
        const std::vector<uint64_t> qf = wf::evaluate_columns_at(quot, pts, 2, D);
        std::vector<uint64_t> ood;                                                         // current rows (trace, quotient), then next rows
        ood.insert(ood.end(), tf.begin(), tf.begin() + 2 * D);
        ood.insert(ood.end(), qf.begin(), qf.begin() + D);
        ood.insert(ood.end(), tf.begin() + 2 * D, tf.end());
        ood.insert(ood.end(), qf.begin() + D, qf.end());
        uint8_t ood_digest[32];
        coin.hash_elements(ood, ood_digest);
        coin.reseed(ood_digest);
        const std::vector<uint64_t> dc = coin.draw(3, D);                                  // 2 trace + 1 constraint coefficients
        std::vector<uint64_t> dc_t(dc.begin(), dc.begin() + 2 * D), dc_c(dc.begin() + 2 * D, dc.end());
        wf::DeviceBuffer deep = wf::deep_compose(tc.polys, nullptr, quot, D, z, dc_t, dc_c);
        wf::DeviceBuffer deep_ev = wf::fft::evaluate_poly_with_offset(deep, F, n, &offset, blowup, D);
        // 5. FRI commit phase
        Channel chan(coin, D);
        wf::FriProver fri(wf::FriOptions{blowup, folding, rem_deg}, hash, F, D, {offset});
        fri.build_layers(chan, std::move(deep_ev), N);
        const size_t fri_layers = fri.layers().size();
        // 6. proof of work, query positions (prover/src/channel.rs:146-185)
        const uint64_t nonce = wf::grind_query_seed(ctx, hash, coin.seed, grinding);
        {
            wf::DeviceBuffer d_out(ctx, 32);
            wf::check(wf_hash_merge_with_int_batch(ctx.handle(), (int)hash, coin.seed, nonce, 1, d_out.data()), "wf_hash_merge_with_int_batch");
            d_out.download(coin.seed, 32);
            coin.counter = 0;
        }
        std::set<uint64_t> pos_set;
        for (const auto &d : coin.next(num_queries)) {
            uint64_t v;
            std::memcpy(&v, d.data(), 8);
            pos_set.insert(v & (N - 1));
        }
        const std::vector<uint64_t> positions(pos_set.begin(), pos_set.end());
        // 7. openings
        const std::vector<uint64_t> t_rows = tc.lde.rows(positions), c_rows = qc.lde.rows(positions);
        auto t_open = tc.tree.prove_batch(positions), c_open = qc.tree.prove_batch(positions);
        wf::FriProver::Proof fp = fri.build_proof(positions);
        ctx.sync();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms < best) best = ms;
        char buf[1024];
        snprintf(buf, sizeof buf,
                 "\"trace_length\": %llu, \"hash\": %d, \"ext_degree\": %u, \"trace_root\": \"%s\", \"constraint_root\": \"%s\", \"pow_nonce\": %llu, "
                 "\"num_unique_queries\": %zu, \"first_position\": %llu, \"fri_layers\": %zu, \"fri_remainder_len\": %zu, \"last_fri_commitment\": \"%s\"",
                 (unsigned long long)n, (int)hash, D, hex(trace_root.data(), 32).c_str(), hex(constraint_root.data(), 32).c_str(), (unsigned long long)nonce,
                 positions.size(), (unsigned long long)positions[0], fri_layers, fp.remainder.size() / D, hex(chan.commitments.back().data(), 32).c_str());
        summary = buf;
        (void)t_rows; (void)c_rows; (void)t_open; (void)c_open;
    }
    printf("{\"example\": \"fib_small (C++ host layer)\", %s, \"prove_ms\": %.3f}\n", summary.c_str(), best);
    return 0;
}
