// Parity of the C++ host layer (include/winterfell_hip.hpp) against the CPU oracle (liboracle.so, test infrastructure):
// the compiled-language counterpart of tests/test_gpu_*.py.  Built and run by tests/test_gpu_cpp_host.py on a GPU box.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/winterfell_hip.hpp"

extern "C" {
// field-generic oracle instantiation for f64 (oracle/field_f64t.c) and the f64 helpers
uint64_t or_f64_new1(uint64_t a);
void or_f64t_evaluate_poly(uint64_t *p, uint64_t n, unsigned D);
void or_f64t_interpolate_poly(uint64_t *p, uint64_t n, unsigned D);
void or_f64t_evaluate_poly_with_offset(const uint64_t *p, uint64_t n, unsigned D, const uint64_t *offset, uint64_t blowup, uint64_t *out);
int or_f64t_build_trace_commitment(int hasher, uint64_t *trace, uint64_t c, uint64_t n, unsigned D, uint64_t blowup, const uint64_t *offset,
                                   uint64_t num_partitions, uint64_t hash_rate, uint64_t *lde, uint8_t *leaves, uint8_t *nodes);
void or_f64t_transpose_slice(const uint64_t *src, uint64_t len, unsigned D, uint64_t N, uint64_t *dst);
int or_f64t_fri_layer_commit(int hasher, const uint64_t *tr, uint64_t rows, unsigned D, uint64_t N, uint8_t *leaves, uint8_t *nodes);
void or_f64t_apply_drp(const uint64_t *values, uint64_t rows, unsigned D, uint64_t N, const uint64_t *offset, const uint64_t *alpha, uint64_t *out);
void or_fri_remainder(int hasher, uint64_t *evals, uint64_t len, unsigned D, uint64_t offset, uint64_t blowup, uint64_t *rem, uint8_t com[32]);
void or_f64t_fib_small_build_trace(uint64_t n, uint64_t *trace);
int or_f64t_evaluate_constraints(int air, const uint64_t *lde, uint64_t row_width, uint64_t n, uint64_t lde_blowup, uint64_t ce_blowup,
                                 const uint64_t *offset, unsigned D, const uint64_t *cc_t, uint64_t num_assert, const uint64_t *a_col,
                                 const uint64_t *a_step, const uint64_t *a_val, const uint64_t *cc_b, uint64_t *out);
void or_f64t_evaluate_columns_at(const uint64_t *polys, uint64_t c, uint64_t n, unsigned pD, const uint64_t *x, unsigned D, uint64_t *out);
void or_f64t_deep_compose(const uint64_t *main_polys, uint64_t c_main, const uint64_t *aux, uint64_t c_aux, const uint64_t *quot, uint64_t c_q,
                          uint64_t n, unsigned D, const uint64_t *z, const uint64_t *cc_t, const uint64_t *cc_c, const uint64_t *otc,
                          const uint64_t *otn, const uint64_t *oqc, const uint64_t *oqn, uint64_t *out);
uint64_t or_row_width(uint64_t base_cols);
// DefaultRandomCoin restatement (oracle/fri.c)
uint64_t or_coin_sizeof(void);
void or_coin_new(void *c, int hasher, const uint64_t *seed, uint64_t n);
void or_coin_reseed(void *c, const uint8_t data[32]);
int or_coin_draw(void *c, unsigned D, uint64_t *out);
void or_coin_seed(const void *c, uint8_t out[32]);
uint64_t or_coin_grind(const void *c, uint32_t factor, uint64_t limit);
}

static int failures = 0;
#define EXPECT(cond, what)                                  \
    do {                                                    \
        if (!(cond)) {                                      \
            printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); \
            failures++;                                     \
        } else {                                            \
            printf("ok   %s\n", what);                      \
        }                                                   \
    } while (0)

static const uint64_t P = 0xffffffff00000001ull;
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state % P;   // every canonical u64 < p is a valid internal (Montgomery) word
}
static std::vector<uint64_t> rand_vec(size_t n) {
    std::vector<uint64_t> v(n);
    for (auto &x : v) x = rnd();
    return v;
}

// oracle-backed Fiat-Shamir channel (fri::DefaultProverChannel)
struct OracleChannel : wf::ProverChannel {
    std::vector<uint8_t> coin;
    unsigned D;
    std::vector<std::vector<uint8_t>> commitments;
    OracleChannel(int hasher, unsigned D_) : coin(or_coin_sizeof()), D(D_) { or_coin_new(coin.data(), hasher, nullptr, 0); }
    void commit_fri_layer(const uint8_t root[32]) override {
        commitments.emplace_back(root, root + 32);
        or_coin_reseed(coin.data(), root);
    }
    std::vector<uint64_t> draw_fri_alpha() override {
        std::vector<uint64_t> a(D);
        if (or_coin_draw(coin.data(), D, a.data())) abort();
        return a;
    }
};

int main() {
    wf::Context ctx(0);
    const wf::Field F = wf::Field::F64;
    const uint64_t offset = or_f64_new1(7);

    // ---- math::fft -----------------------------------------------------------------------------------------------------
    {
        const uint64_t n = 1 << 10;
        const unsigned D = 2;
        std::vector<uint64_t> p = rand_vec(n * D), want = p;
        wf::DeviceBuffer d(ctx, p);
        wf::fft::evaluate_poly(d, F, n, D);
        or_f64t_evaluate_poly(want.data(), n, D);
        EXPECT(d.to_host<uint64_t>() == want, "fft::evaluate_poly (quadratic extension, 2^10)");
        wf::fft::interpolate_poly(d, F, n, D);
        EXPECT(d.to_host<uint64_t>() == p, "fft::interpolate_poly inverts it");
        std::vector<uint64_t> lde(n * 8 * D);
        or_f64t_evaluate_poly_with_offset(p.data(), n, D, &offset, 8, lde.data());
        wf::DeviceBuffer e = wf::fft::evaluate_poly_with_offset(d, F, n, &offset, 8, D);
        EXPECT(e.to_host<uint64_t>() == lde, "fft::evaluate_poly_with_offset (blowup 8)");
        bool threw = false;
        try {
            wf::fft::evaluate_poly(d, F, 1000, D);
        } catch (const std::invalid_argument &) {
            threw = true;
        }
        EXPECT(threw, "non power-of-two length is rejected (fft/mod.rs:90-93)");
    }

    // ---- wf_malloc / wf_free: the stream-ordered pool hands a freed block to the next request of that size class, trims on demand ----
    {
        void *a = nullptr, *b = nullptr, *c = nullptr;
        wf::check(wf_malloc(ctx.handle(), 3u << 20, &a), "wf_malloc");
        wf::check(wf_free(ctx.handle(), a), "wf_free");
        wf::check(wf_malloc(ctx.handle(), (3u << 20) - 4096, &b), "wf_malloc");          // same 2 MiB size class: the cached block
        wf::check(wf_malloc(ctx.handle(), 3u << 20, &c), "wf_malloc");                   // pool empty again: a fresh block
        EXPECT(a == b && c != b, "freed block is reused for the next request of its size class");
        wf::check(wf_free(ctx.handle(), b), "wf_free");
        wf::check(wf_free(ctx.handle(), c), "wf_free");
        wf::check(wf_ctx_trim(ctx.handle()), "wf_ctx_trim");
        void *d = nullptr;
        wf::check(wf_malloc(ctx.handle(), 100, &d), "wf_malloc");
        wf::check(wf_free(ctx.handle(), d), "wf_free");
        wf::check(wf_free(ctx.handle(), nullptr), "wf_free(nullptr)");
        // a freed buffer may be reused while work that read it is still queued: results must not change
        std::vector<uint64_t> p = rand_vec(1 << 12), want = p;
        or_f64t_evaluate_poly(want.data(), 1 << 12, 1);
        bool ok = true;
        for (int it = 0; it < 8; it++) {
            wf::DeviceBuffer x(ctx, p);
            wf::fft::evaluate_poly(x, F, 1 << 12);
            wf::DeviceBuffer y = x.clone();          // y's block = whatever the previous iteration freed
            ok = ok && y.to_host<uint64_t>() == want;
        }
        EXPECT(ok, "buffers recycled without synchronisation keep their contents (stream order)");
    }

    // ---- wf::Comm: the column-sharded commitment, 4 ranks as threads over the loopback transport, vs the oracle's partitioned one ----
    {
        const int G = 4;
        const uint64_t n = 1 << 7, cps = 3, c = G * cps, blowup = 8, N = n * blowup, per = N / G;
        std::vector<uint64_t> trace = rand_vec(c * n), o_trace = trace;
        const uint64_t rw = or_row_width(c);
        std::vector<uint64_t> o_lde(N * rw);
        std::vector<uint8_t> o_leaves(N * 32), o_nodes(N * 32);
        or_f64t_build_trace_commitment(0, o_trace.data(), c, n, 1, blowup, &offset, G, 1, o_lde.data(), o_leaves.data(), o_nodes.data());
        std::vector<std::unique_ptr<wf::Context>> rctx;
        std::vector<wf::Context *> rptr;
        for (int r = 0; r < G; r++) {
            rctx.emplace_back(new wf::Context(0));
            rptr.push_back(rctx.back().get());
        }
        std::vector<wf::Comm> comms = wf::Comm::loopback(rptr);
        std::vector<int> ok(G, 0);
        std::vector<std::thread> th;
        for (int r = 0; r < G; r++)
            th.emplace_back([&, r]() {
                std::vector<uint64_t> sh(trace.begin() + r * cps * n, trace.begin() + (r + 1) * cps * n);
                wf::ColMatrix cm{wf::DeviceBuffer(*rptr[r], sh), F, (uint32_t)cps, 1, n};
                wf::ShardedCommitment sc = comms[r].sharded_commit(wf::Hash::Blake3_256, cm, blowup, &offset);
                const std::vector<uint8_t> leaves = sc.leaves.to_host<uint8_t>(), nodes = sc.nodes.to_host<uint8_t>(), top = sc.top.to_host<uint8_t>();
                bool good = std::memcmp(leaves.data(), &o_leaves[r * per * 32], per * 32) == 0 && std::memcmp(sc.root.data(), &o_nodes[32], 32) == 0 &&
                            std::memcmp(&top[32], &o_nodes[32], (G - 1) * 32) == 0;
                for (uint64_t j = 1; j < per && good; j++) {      // local heap index j -> global ((G + r) << depth) + offset in level
                    uint32_t depth = 0;
                    while ((2ull << depth) <= j) depth++;
                    good = std::memcmp(&nodes[j * 32], &o_nodes[((((uint64_t)G + r) << depth) + (j - (1ull << depth))) * 32], 32) == 0;
                }
                ok[r] = good;
            });
        for (auto &t : th) t.join();
        bool all = true;
        for (int r = 0; r < G; r++) all = all && ok[r];
        EXPECT(all, "wf::Comm::sharded_commit over 4 loopback ranks == PartitionOptions(4, 1) commitment (leaves, subtrees, top tree, root)");
    }

    // ---- build_trace_commitment + MerkleTree (Blake3_256 and Rp64_256, partitions) ---------------------------------------------
    for (int hasher = 0; hasher < 2; hasher++) {
        const uint64_t n = 1 << 8, c = 12, blowup = 8, N = n * blowup, parts = hasher == 0 ? 4 : 1;
        std::vector<uint64_t> trace = rand_vec(c * n), o_trace = trace;
        const uint64_t rw = or_row_width(c);
        std::vector<uint64_t> o_lde(N * rw);
        std::vector<uint8_t> o_leaves(N * 32), o_nodes(N * 32);
        or_f64t_build_trace_commitment(hasher, o_trace.data(), c, n, 1, blowup, &offset, parts, 1, o_lde.data(), o_leaves.data(), o_nodes.data());
        wf::ColMatrix cm{wf::DeviceBuffer(ctx, trace), F, (uint32_t)c, 1, n};
        wf::TraceCommitment tc = wf::build_trace_commitment((wf::Hash)hasher, cm, blowup, &offset, wf::PartitionOptions{(uint32_t)parts, 1});
        EXPECT(tc.polys.data.to_host<uint64_t>() == o_trace, hasher ? "trace polys (Rp64_256 run)" : "trace polys (Blake3_256 run)");
        EXPECT(tc.lde.data.to_host<uint64_t>() == o_lde, "row-major trace LDE");
        EXPECT(tc.tree.leaves() == o_leaves, "row digests");
        EXPECT(tc.tree.nodes() == o_nodes, "Merkle nodes (heap layout)");
        auto proof = tc.tree.prove(5);
        EXPECT(proof.size() == tc.tree.depth() + 1 && std::memcmp(proof[1].data(), &o_leaves[4 * 32], 32) == 0 &&
                   std::memcmp(proof.back().data(), &o_nodes[3 * 32], 32) == 0, "MerkleTree::prove path");
        {
            // prove_batch vs a set-based restatement: at every level a sibling goes into the proof iff it is not itself on
            // the path of an opened leaf; nodes are filed per pair of sibling leaf positions, in level order
            const std::vector<uint64_t> idx = {700, 3, 2, 1500, 1501, 64};
            auto pb = tc.tree.prove_batch(idx);
            bool ok = pb.second.depth == tc.tree.depth() && pb.first.size() == idx.size();
            for (size_t k = 0; k < idx.size() && ok; k++) ok = std::memcmp(pb.first[k].data(), &o_leaves[idx[k] * 32], 32) == 0;
            std::vector<uint64_t> sorted = idx;
            std::sort(sorted.begin(), sorted.end());
            std::vector<uint64_t> pairs;
            for (uint64_t v : sorted)
                if (pairs.empty() || pairs.back() != (v & ~1ull)) pairs.push_back(v & ~1ull);
            ok = ok && pb.second.nodes.size() == pairs.size();
            size_t total = 0, want_total = 0;
            for (const auto &lst : pb.second.nodes) total += lst.size();
            // expected multiset of proof digests
            std::vector<std::vector<uint8_t>> want;
            std::vector<uint64_t> known(sorted);          // leaf level: heap index = N + leaf
            for (auto &v : known) v += N;
            for (uint32_t d = 0; d < tc.tree.depth(); d++) {
                std::vector<uint64_t> nxt;
                for (uint64_t node : known) {
                    const uint64_t sib = node ^ 1;
                    if (!std::binary_search(known.begin(), known.end(), sib)) {
                        const uint8_t *src = d == 0 ? &o_leaves[(sib - N) * 32] : &o_nodes[sib * 32];
                        want.emplace_back(src, src + 32);
                        want_total++;
                    }
                    if (nxt.empty() || nxt.back() != node >> 1) nxt.push_back(node >> 1);
                }
                known.swap(nxt);
            }
            std::vector<std::vector<uint8_t>> got;
            for (const auto &lst : pb.second.nodes)
                for (const auto &dg : lst) got.push_back(dg);
            std::sort(got.begin(), got.end());
            std::sort(want.begin(), want.end());
            EXPECT(ok && total == want_total && got == want, "MerkleTree::prove_batch (leaves in caller order, exactly the non-recomputable siblings)");
        }
        std::vector<uint64_t> rows = tc.lde.rows({3, 700});
        EXPECT(std::memcmp(rows.data(), &o_lde[3 * rw], c * 8) == 0 && std::memcmp(&rows[c], &o_lde[700 * rw], c * 8) == 0, "TraceLde::query rows");
        // two-step path: evaluate_polys_over + commit_to_rows == the fused call
        wf::RowMatrix lde2 = wf::RowMatrix::evaluate_polys_over(tc.polys, blowup, &offset);
        wf::MerkleTree t2 = lde2.commit_to_rows((wf::Hash)hasher, wf::PartitionOptions{(uint32_t)parts, 1});
        EXPECT(t2.nodes() == o_nodes, "RowMatrix::commit_to_rows");
        // column-major path of the reference's benches/row_matrix.rs: ColMatrix::evaluate_columns_over holds the same values
        // as the row-major LDE, and with one partition ColMatrix::commit_to_rows commits to the same rows
        wf::ColMatrix ev = tc.polys.evaluate_columns_over(blowup, &offset);
        std::vector<uint64_t> evh = ev.data.to_host<uint64_t>();
        bool same = ev.num_rows == N;
        for (uint64_t r = 0; r < N && same; r += 37)
            for (uint64_t k = 0; k < c; k++) same = same && evh[k * N + r] == o_lde[r * rw + k];
        EXPECT(same, "ColMatrix::evaluate_columns_over == transposed RowMatrix");
        if (parts == 1) EXPECT(ev.commit_to_rows((wf::Hash)hasher).nodes() == o_nodes, "ColMatrix::commit_to_rows");
    }
    {
        bool threw = false;
        try {
            wf::MerkleTree t(wf::Hash::Blake3_256, wf::DeviceBuffer(ctx, std::vector<uint8_t>(3 * 32)), 3);
        } catch (const wf::Error &e) {
            threw = e.status() == WF_ERR_NOT_POWER_OF_TWO;
        }
        EXPECT(threw, "MerkleTree: NumberOfLeavesNotPowerOfTwo");
    }

    // ---- FriProver::build_layers with the oracle's DefaultProverChannel -----------------------------------------------------------
    {
        const unsigned D = 2;
        const uint64_t len = 1 << 12, N = 4, blowup = 8;
        std::vector<uint64_t> poly = rand_vec((len / blowup) * D), ev(len * D);
        or_f64t_evaluate_poly_with_offset(poly.data(), len / blowup, D, &offset, blowup, ev.data());
        OracleChannel chan(0, D), ochan(0, D);
        wf::FriProver prover(wf::FriOptions{blowup, N, 7}, wf::Hash::Blake3_256, F, D, {offset});
        prover.build_layers(chan, wf::DeviceBuffer(ctx, ev), len);
        std::vector<uint64_t> cur = ev;
        uint64_t length = len;
        bool ok = true;
        for (size_t k = 0; k < prover.layers().size(); k++) {
            const uint64_t rows = length / N;
            std::vector<uint64_t> tr(length * D), folded(rows * D);
            std::vector<uint8_t> leaves(rows * 32), nodes(rows * 32);
            or_f64t_transpose_slice(cur.data(), length, D, N, tr.data());
            or_f64t_fri_layer_commit(0, tr.data(), rows, D, N, leaves.data(), nodes.data());
            ochan.commit_fri_layer(&nodes[32]);
            std::vector<uint64_t> alpha = ochan.draw_fri_alpha();
            or_f64t_apply_drp(tr.data(), rows, D, N, &offset, alpha.data(), folded.data());
            ok = ok && prover.layers()[k].commitment.nodes() == nodes && prover.layers()[k].evaluations.to_host<uint64_t>() == tr;
            cur = folded;
            length = rows;
        }
        std::vector<uint64_t> rem((length / blowup) * D);
        uint8_t com[32];
        or_fri_remainder(0, cur.data(), length, D, offset, blowup, rem.data(), com);
        EXPECT(ok && prover.layers().size() == 3, "FriProver layers: nodes and transposed evaluations");
        EXPECT(prover.remainder_poly() == rem && std::memcmp(chan.commitments.back().data(), com, 32) == 0, "FRI remainder polynomial and its commitment");
        // the same loop with the coin on the device (wf::DeviceCoin + wf_fri_build_layers): same layers, and the transcript the
        // oracle channel recorded — every layer root, the remainder commitment, every alpha, the coin's final seed
        {
            OracleChannel fresh(0, D);
            uint8_t seed0[32], seed_end[32];
            or_coin_seed(fresh.coin.data(), seed0);
            wf::DeviceCoin dcoin(ctx, wf::Hash::Blake3_256, F, seed0);
            wf::FriProver fused(wf::FriOptions{blowup, N, 7}, wf::Hash::Blake3_256, F, D, {offset});
            wf::FriProver::Transcript tr = fused.build_layers(dcoin, wf::DeviceBuffer(ctx, ev), len);
            bool fok = tr.roots.size() == chan.commitments.size() && fused.layers().size() == prover.layers().size() &&
                       fused.remainder_poly() == prover.remainder_poly();
            for (size_t k = 0; fok && k < tr.roots.size(); k++) fok = std::memcmp(tr.roots[k].data(), chan.commitments[k].data(), 32) == 0;
            std::vector<uint64_t> alphas;
            OracleChannel replay(0, D);
            for (size_t k = 0; k + 1 < chan.commitments.size(); k++) {
                replay.commit_fri_layer(chan.commitments[k].data());
                const std::vector<uint64_t> a = replay.draw_fri_alpha();
                alphas.insert(alphas.end(), a.begin(), a.end());
            }
            replay.commit_fri_layer(chan.commitments.back().data());
            or_coin_seed(replay.coin.data(), seed_end);
            for (size_t k = 0; fok && k < fused.layers().size(); k++)
                fok = fused.layers()[k].commitment.nodes() == prover.layers()[k].commitment.nodes() &&
                      fused.layers()[k].evaluations.to_host<uint64_t>() == prover.layers()[k].evaluations.to_host<uint64_t>();
            const auto st = dcoin.read();
            EXPECT(fok && tr.alphas == alphas && std::memcmp(st.first.data(), seed_end, 32) == 0 && st.second == 0,
                   "FriProver::build_layers with a device coin: layers, roots, alphas, coin state");
        }
        // query phase: rows of every layer at the folded positions, in fold_positions order; build_proof resets the prover
        std::vector<std::vector<uint64_t>> layer_rows;
        for (const auto &l : prover.layers()) layer_rows.push_back(l.evaluations.to_host<uint64_t>());
        const std::vector<uint64_t> positions = {5, 1029, 4000, 2053, 77};
        EXPECT((wf::FriProver::fold_positions(positions, len, N) == std::vector<uint64_t>{5, 928, 77}), "fold_positions (duplicates dropped, first-seen order)");
        wf::FriProver::Proof fp = prover.build_proof(positions);
        bool qok = fp.layers.size() == layer_rows.size() && fp.remainder == rem && prover.layers().empty();
        std::vector<uint64_t> pos = positions;
        uint64_t dom = len;
        for (size_t k = 0; k < fp.layers.size() && qok; k++) {
            pos = wf::FriProver::fold_positions(pos, dom, N);
            const size_t rw = N * D;
            qok = fp.layers[k].values.size() == pos.size() * rw && fp.layers[k].proof.depth == wf::log2_exact(dom / N, "rows");
            for (size_t q = 0; q < pos.size() && qok; q++) qok = std::memcmp(&fp.layers[k].values[q * rw], &layer_rows[k][pos[q] * rw], rw * 8) == 0;
            dom /= N;
        }
        EXPECT(qok, "FriProver::build_proof: queried rows per layer, remainder, reset");
    }

    // ---- fib_small: constraint evaluation, OOD frame, DEEP composition, grinding ------------------------------------------------------
    {
        const unsigned D = 2;
        const uint64_t n = 1 << 8, blowup = 8, ce_blowup = 2, N = n * blowup;
        std::vector<uint64_t> trace(2 * n);
        or_f64t_fib_small_build_trace(n, trace.data());
        wf::ColMatrix cm{wf::DeviceBuffer(ctx, trace), F, 2, 1, n};
        wf::TraceCommitment tc = wf::build_trace_commitment(wf::Hash::Blake3_256, cm, blowup, &offset);
        std::vector<uint64_t> cc_t = rand_vec(2 * D), cc_b = rand_vec(3 * D);
        const uint64_t one = or_f64_new1(1), result = trace[2 * n - 1];
        std::vector<wf::Assertion> as{{0, 0, {one}}, {1, 0, {one}}, {1, n - 1, {result}}};
        wf::DeviceBuffer ev = wf::evaluate_constraints(WF_AIR_FIB_SMALL, tc.lde, n, blowup, ce_blowup, &offset, D, cc_t, as, cc_b);
        std::vector<uint64_t> o_lde = tc.lde.data.to_host<uint64_t>(), want(n * ce_blowup * D);
        const uint64_t a_col[3] = {0, 1, 1}, a_step[3] = {0, 0, n - 1}, a_val[3] = {one, one, result};
        or_f64t_evaluate_constraints(0, o_lde.data(), tc.lde.row_width, n, blowup, ce_blowup, &offset, D, cc_t.data(), 3, a_col, a_step, a_val,
                                     cc_b.data(), want.data());
        EXPECT(ev.to_host<uint64_t>() == want, "evaluate_constraints (FibSmall, quadratic extension)");
        // composition poly (1 column) + OOD frames + DEEP
        wf::fft::interpolate_poly_with_offset(ev, F, n * ce_blowup, &offset, D);
        std::vector<uint64_t> comp = ev.to_host<uint64_t>();
        comp.resize(n * D);
        wf::ColMatrix quot{wf::DeviceBuffer(ctx, comp), F, 1, D, n};
        std::vector<uint64_t> z = rand_vec(D), cct = rand_vec(2 * D), ccq = rand_vec(1 * D);
        std::vector<uint64_t> frame = wf::evaluate_columns_at(tc.polys, z, 1, D), o_frame(2 * D);
        std::vector<uint64_t> polys_h = tc.polys.data.to_host<uint64_t>();
        or_f64t_evaluate_columns_at(polys_h.data(), 2, n, 1, z.data(), D, o_frame.data());
        EXPECT(frame == o_frame, "out-of-domain trace frame (evaluate_columns_at)");
        wf::DeviceBuffer deep = wf::deep_compose(tc.polys, nullptr, quot, D, z, cct, ccq);
        std::vector<uint64_t> o_deep(n * D), zeros(4 * D, 0);
        or_f64t_deep_compose(polys_h.data(), 2, nullptr, 0, comp.data(), 1, n, D, z.data(), cct.data(), ccq.data(), zeros.data(), zeros.data(), zeros.data(),
                             zeros.data(), o_deep.data());
        EXPECT(deep.to_host<uint64_t>() == o_deep, "DEEP composition polynomial");
        // grinding against the oracle coin
        std::vector<uint8_t> coin(or_coin_sizeof());
        or_coin_new(coin.data(), 0, cc_t.data(), 2);
        uint8_t seed[32];
        or_coin_seed(coin.data(), seed);
        EXPECT(wf::grind_query_seed(ctx, wf::Hash::Blake3_256, seed, 12) == or_coin_grind(coin.data(), 12, 1ull << 30), "grind_query_seed (factor 12)");
        (void)N;
    }

    printf(failures ? "FAILED: %d check(s)\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
