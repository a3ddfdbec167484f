// C++ host layer over the C ABI of libwinterfell_hip.so, mirroring the reference's (Rust) interfaces for the hot path:
// the reference is compiled code and its toolchain is absent from the build image, so this header is the host side a
// compiled caller links against (the Python package winterfell_amd/ is the same mirror for the test-suite).  It is
// header only, needs nothing but <winterfell_hip.h>, and owns device memory through RAII buffers.
//
//   wf::Context                        one GPU + stream                                   (no reference counterpart)
//   wf::DeviceBuffer                   HBM allocation with upload / download
//   wf::fft::*                         math::fft::{evaluate_poly, interpolate_poly, ..._with_offset, get_twiddles}
//                                      (math/src/fft/mod.rs:85-505) — same argument meaning, same panics as exceptions
//   wf::ColMatrix / wf::RowMatrix      prover/src/matrix/{col_matrix.rs, row_matrix.rs}
//   wf::MerkleTree                     crypto/src/merkle/mod.rs:91-458 (nodes in the reference's heap layout)
//   wf::build_trace_commitment         prover/src/trace/trace_lde/default/mod.rs:245-282
//   wf::new_trace_lde_from_host        Prover::new_trace_lde (prover/src/lib.rs:182-190) from host columns, PCIe legs overlapped with the kernels
//   wf::FriProver                      fri/src/prover/mod.rs:100-239 (commit phase; the channel is a caller interface)
//   wf::evaluate_constraints, wf::ood_frame, wf::deep_compose, wf::grind_query_seed  — SURVEY §8(f) N1–N3
//
// Field elements cross this layer exactly as they cross the C ABI: the reference's in-memory words (f64 / f62 Montgomery
// u64, f128 canonical u128 as two u64), extension elements as consecutive base elements.
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "winterfell_hip.h"

namespace wf {

enum class Field : int { F64 = WF_FIELD_F64, F128 = WF_FIELD_F128, F62 = WF_FIELD_F62 };
enum class Hash : int {
    Blake3_256 = WF_HASH_BLAKE3_256,
    Rp64_256 = WF_HASH_RP64_256,
    Sha3_256 = WF_HASH_SHA3_256,
    RpJive64_256 = WF_HASH_RPJIVE64_256,
    Rp62_248 = WF_HASH_RP62_248,
    Blake3_192 = WF_HASH_BLAKE3_192
};

// words (u64) per base-field element
inline uint32_t words(Field f) { return f == Field::F128 ? 2u : 1u; }

// a non-zero status of the C ABI; the Rust shim would panic! / return Err at the same places
class Error : public std::runtime_error {
  public:
    Error(int status, const char *where) : std::runtime_error(std::string(where) + ": " + wf_strerror(status)), status_(status) {}
    int status() const { return status_; }

  private:
    int status_;
};
inline void check(int status, const char *where) {
    if (status != WF_OK) throw Error(status, where);
}

class Context {
  public:
    explicit Context(int device = 0) { check(wf_ctx_create(device, &h_), "wf_ctx_create"); }
    ~Context() {
        if (h_) wf_ctx_destroy(h_);
    }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    wf_ctx *handle() const { return h_; }
    void sync() { check(wf_ctx_sync(h_), "wf_ctx_sync"); }
    void prof_enable(bool on) { check(wf_prof_enable(h_, on ? 1 : 0), "wf_prof_enable"); }
    std::string prof_collect() {
        std::vector<char> buf(1 << 16);
        check(wf_prof_collect(h_, buf.data(), buf.size()), "wf_prof_collect");
        return std::string(buf.data());
    }

  private:
    wf_ctx *h_ = nullptr;
};

// HBM allocation of `n` u64 words (or bytes for digests), freed on destruction
class DeviceBuffer {
  public:
    DeviceBuffer() = default;
    DeviceBuffer(Context &ctx, size_t bytes) : ctx_(&ctx), bytes_(bytes) {
        if (bytes) check(wf_malloc(ctx.handle(), bytes, &p_), "wf_malloc");
    }
    DeviceBuffer(Context &ctx, const void *host, size_t bytes) : DeviceBuffer(ctx, bytes) { upload(host, bytes); }
    template <class T>
    DeviceBuffer(Context &ctx, const std::vector<T> &host) : DeviceBuffer(ctx, host.data(), host.size() * sizeof(T)) {}
    ~DeviceBuffer() { release(); }
    DeviceBuffer(DeviceBuffer &&o) noexcept { *this = std::move(o); }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
        if (this != &o) {
            release();
            ctx_ = o.ctx_;
            p_ = o.p_;
            bytes_ = o.bytes_;
            o.p_ = nullptr;
            o.bytes_ = 0;
        }
        return *this;
    }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;

    void *data() const { return p_; }
    size_t bytes() const { return bytes_; }
    void upload(const void *host, size_t bytes) { check(wf_memcpy_h2d(ctx_->handle(), p_, host, bytes), "wf_memcpy_h2d"); }
    void download(void *host, size_t bytes, size_t offset = 0) const {
        check(wf_memcpy_d2h(ctx_->handle(), host, (const uint8_t *)p_ + offset, bytes), "wf_memcpy_d2h");
    }
    template <class T>
    std::vector<T> to_host() const {
        std::vector<T> out(bytes_ / sizeof(T));
        if (bytes_) download(out.data(), bytes_);
        return out;
    }
    DeviceBuffer clone() const {
        DeviceBuffer c(*ctx_, bytes_);
        if (bytes_) check(wf_memcpy_d2d(ctx_->handle(), c.p_, p_, bytes_), "wf_memcpy_d2d");
        return c;
    }
    Context &ctx() const { return *ctx_; }

  private:
    void release() {
        if (p_) wf_free(ctx_->handle(), p_);
        p_ = nullptr;
    }
    Context *ctx_ = nullptr;
    void *p_ = nullptr;
    size_t bytes_ = 0;
};

inline uint32_t log2_exact(uint64_t n, const char *what) {
    if (n == 0 || (n & (n - 1))) throw std::invalid_argument(std::string("number of ") + what + " must be a power of 2");   // fft/mod.rs:90-93
    uint32_t l = 0;
    while ((1ull << l) < n) l++;
    return l;
}

// ---- math::fft ---------------------------------------------------------------------------------------------------
namespace fft {

// fft::get_twiddles / get_inv_twiddles (mod.rs:455-505): n/2 elements
inline DeviceBuffer get_twiddles(Context &ctx, Field f, uint64_t domain_size, bool inverse = false) {
    const uint32_t log_n = log2_exact(domain_size, "domain size");
    DeviceBuffer out(ctx, (domain_size / 2) * 8 * words(f));
    check(wf_fft_get_twiddles(ctx.handle(), (int)f, log_n, inverse ? 1 : 0, out.data()), "wf_fft_get_twiddles");
    return out;
}
// fft::evaluate_poly (mod.rs:85-112): in place over `num_elements` elements of extension degree `ext_degree`
inline void evaluate_poly(DeviceBuffer &p, Field f, uint64_t num_elements, uint32_t ext_degree = 1, uint32_t batch = 1) {
    check(wf_fft_evaluate_poly(p.ctx().handle(), (int)f, ext_degree, p.data(), log2_exact(num_elements, "coefficients"), batch), "wf_fft_evaluate_poly");
}
// fft::interpolate_poly (mod.rs:264-295)
inline void interpolate_poly(DeviceBuffer &evals, Field f, uint64_t num_elements, uint32_t ext_degree = 1, uint32_t batch = 1) {
    check(wf_fft_interpolate_poly(evals.ctx().handle(), (int)f, ext_degree, evals.data(), log2_exact(num_elements, "values"), batch), "wf_fft_interpolate_poly");
}
// fft::evaluate_poly_with_offset (mod.rs:168-211): `offset` = one base element (W words, internal form)
inline DeviceBuffer evaluate_poly_with_offset(const DeviceBuffer &p, Field f, uint64_t num_elements, const uint64_t *offset,
                                              uint64_t blowup_factor, uint32_t ext_degree = 1) {
    DeviceBuffer out(p.ctx(), num_elements * blowup_factor * ext_degree * 8 * words(f));
    check(wf_fft_evaluate_poly_with_offset(p.ctx().handle(), (int)f, ext_degree, p.data(), log2_exact(num_elements, "coefficients"), offset,
                                           log2_exact(blowup_factor, "blowup factor"), out.data()), "wf_fft_evaluate_poly_with_offset");
    return out;
}
// fft::interpolate_poly_with_offset (mod.rs:351-386): in place
inline void interpolate_poly_with_offset(DeviceBuffer &evals, Field f, uint64_t num_elements, const uint64_t *offset, uint32_t ext_degree = 1) {
    check(wf_fft_interpolate_poly_with_offset(evals.ctx().handle(), (int)f, ext_degree, evals.data(), log2_exact(num_elements, "values"), offset),
          "wf_fft_interpolate_poly_with_offset");
}

}  // namespace fft

// ---- air::PartitionOptions (air/src/options.rs:405-451) ----------------------------------------------------------------
struct PartitionOptions {
    uint32_t num_partitions = 1, hash_rate = 1;
};

// ---- crypto::MerkleTree (crypto/src/merkle/mod.rs) -----------------------------------------------------------------------
class MerkleTree {
  public:
    // MerkleTree::new (mod.rs:116-135): `leaves` = num_leaves 32-byte digests in HBM.  TooFewLeaves /
    // NumberOfLeavesNotPowerOfTwo surface as wf::Error with WF_ERR_TOO_FEW_LEAVES / WF_ERR_NOT_POWER_OF_TWO.
    MerkleTree(Hash h, DeviceBuffer leaves, uint64_t num_leaves) : hash_(h), leaves_(std::move(leaves)), n_(num_leaves), nodes_(leaves_.ctx(), num_leaves * 32) {
        check(wf_merkle_build(leaves_.ctx().handle(), (int)h, leaves_.data(), num_leaves, nodes_.data()), "wf_merkle_build");
    }
    MerkleTree(Hash h, DeviceBuffer leaves, DeviceBuffer nodes, uint64_t num_leaves)
        : hash_(h), leaves_(std::move(leaves)), n_(num_leaves), nodes_(std::move(nodes)) {}
    uint64_t num_leaves() const { return n_; }
    uint32_t depth() const { return log2_exact(n_, "leaves"); }
    const DeviceBuffer &nodes_device() const { return nodes_; }
    const DeviceBuffer &leaves_device() const { return leaves_; }
    // heap layout: root at 1, children of i at 2i / 2i+1, nodes[0] = default digest (mod.rs:344-368)
    const std::vector<uint8_t> &nodes() const {
        if (h_nodes_.empty()) h_nodes_ = nodes_.to_host<uint8_t>();
        return h_nodes_;
    }
    const std::vector<uint8_t> &leaves() const {
        if (h_leaves_.empty()) h_leaves_ = leaves_.to_host<uint8_t>();
        return h_leaves_;
    }
    // digests at `positions` of the leaf / node array: one device gather (wf_rows_fetch, 32-byte rows) — an opening touches
    // depth + 1 of the 2n digests, so the tree stays in HBM unless nodes() / leaves() were asked for explicitly
    std::vector<uint8_t> fetch(const DeviceBuffer &from, const std::vector<uint64_t> &positions) const {
        std::vector<uint8_t> out(positions.size() * 32);
        if (!positions.empty())
            check(wf_rows_fetch(from.ctx().handle(), from.data(), 4, 4, 8, positions.data(), (uint32_t)positions.size(), out.data()), "wf_rows_fetch");
        return out;
    }
    std::vector<uint8_t> root() const { return fetch(nodes_, {1}); }
    // MerkleTree::prove (mod.rs:193-215): [leaf, sibling leaf, then the sibling node of every level up to the root]
    std::vector<std::vector<uint8_t>> prove(uint64_t index) const {
        if (index >= n_) throw std::out_of_range("LeafIndexOutOfBounds");
        auto digest = [](const std::vector<uint8_t> &v, uint64_t i) { return std::vector<uint8_t>(v.begin() + 32 * i, v.begin() + 32 * (i + 1)); };
        const std::vector<uint8_t> lv = fetch(leaves_, {index, index ^ 1});
        std::vector<uint64_t> path;
        for (uint64_t i = (index + n_) >> 1; i > 1; i >>= 1) path.push_back(i ^ 1);
        const std::vector<uint8_t> nd = fetch(nodes_, path);
        std::vector<std::vector<uint8_t>> out{digest(lv, 0), digest(lv, 1)};
        for (uint64_t k = 0; k < path.size(); k++) out.push_back(digest(nd, k));
        return out;
    }

    // MerkleTree::prove_batch (mod.rs:217-272): the opened leaves (in the order of `indexes`) and, per pair of sibling leaf
    // positions, the proof nodes a verifier cannot recompute.  The walk only computes WHICH digests are needed; they come
    // back in two device gathers.  Throws std::out_of_range / std::invalid_argument for LeafIndexOutOfBounds,
    // TooFewLeafIndexes, DuplicateLeafIndex.
    struct BatchProof {
        std::vector<std::vector<std::vector<uint8_t>>> nodes;   // nodes[pair][k] = 32 bytes
        uint32_t depth = 0;
    };
    std::pair<std::vector<std::vector<uint8_t>>, BatchProof> prove_batch(const std::vector<uint64_t> &indexes) const {
        if (indexes.empty()) throw std::invalid_argument("TooFewLeafIndexes");
        std::map<uint64_t, size_t> index_map;
        for (size_t k = 0; k < indexes.size(); k++) {
            if (indexes[k] >= n_) throw std::out_of_range("LeafIndexOutOfBounds");
            if (!index_map.emplace(indexes[k], k).second) throw std::invalid_argument("DuplicateLeafIndex");
        }
        std::vector<uint64_t> pairs;
        for (const auto &kv : index_map)
            if (pairs.empty() || pairs.back() != (kv.first & ~1ull)) pairs.push_back(kv.first & ~1ull);
        struct Ref { bool leaf; size_t slot; };
        std::vector<uint64_t> want_l, want_n, cur;
        std::vector<size_t> leaf_slot(indexes.size());
        std::vector<std::vector<Ref>> refs(pairs.size());
        for (size_t i = 0; i < pairs.size(); i++) {
            for (uint64_t j = pairs[i]; j < pairs[i] + 2; j++) {
                want_l.push_back(j);
                auto it = index_map.find(j);
                if (it != index_map.end()) leaf_slot[it->second] = want_l.size() - 1;
                else refs[i].push_back(Ref{true, want_l.size() - 1});
            }
            cur.push_back((pairs[i] + n_) >> 1);
        }
        for (uint32_t d = 1; d < depth(); d++) {
            std::vector<uint64_t> nxt;
            for (size_t i = 0; i < cur.size(); i++) {
                const uint64_t node = cur[i], sib = node ^ 1;
                if (i + 1 < cur.size() && cur[i + 1] == sib) {
                    i++;
                } else {
                    want_n.push_back(sib);
                    refs[i].push_back(Ref{false, want_n.size() - 1});   // filed under the node's position in this level's list
                }
                nxt.push_back(sib >> 1);
            }
            cur.swap(nxt);
        }
        const std::vector<uint8_t> got_l = fetch(leaves_, want_l), got_n = fetch(nodes_, want_n);
        auto digest = [](const std::vector<uint8_t> &v, size_t i) { return std::vector<uint8_t>(v.begin() + 32 * i, v.begin() + 32 * (i + 1)); };
        std::vector<std::vector<uint8_t>> opened;
        for (size_t k = 0; k < indexes.size(); k++) opened.push_back(digest(got_l, leaf_slot[k]));
        BatchProof bp;
        bp.depth = depth();
        for (const auto &lst : refs) {
            bp.nodes.emplace_back();
            for (const Ref &r : lst) bp.nodes.back().push_back(digest(r.leaf ? got_l : got_n, r.slot));
        }
        return {opened, bp};
    }

  private:
    Hash hash_;
    DeviceBuffer leaves_;
    uint64_t n_;
    DeviceBuffer nodes_;
    mutable std::vector<uint8_t> h_nodes_, h_leaves_;
};

// ---- prover::matrix --------------------------------------------------------------------------------------------------------
// ColMatrix (col_matrix.rs:28-62): num_cols columns of num_rows elements, column k at k * col_stride base elements
struct ColMatrix {
    DeviceBuffer data;
    Field field = Field::F64;
    uint32_t num_cols = 0, ext_degree = 1;
    uint64_t num_rows = 0;
    uint64_t col_stride() const { return num_rows * ext_degree; }
    // ColMatrix::interpolate_columns (col_matrix.rs:192-202): returns a new matrix of coefficients
    ColMatrix interpolate_columns() const {
        ColMatrix out{data.clone(), field, num_cols, ext_degree, num_rows};
        check(wf_interpolate_columns(data.ctx().handle(), (int)field, ext_degree, out.data.data(), num_cols, col_stride(), log2_exact(num_rows, "rows")),
              "wf_interpolate_columns");
        return out;
    }
    // ColMatrix::evaluate_columns_over (col_matrix.rs:230-243): column-major coset LDE
    ColMatrix evaluate_columns_over(uint64_t blowup, const uint64_t *domain_offset) const {
        Context &ctx = data.ctx();
        ColMatrix out{DeviceBuffer(ctx, num_cols * num_rows * blowup * ext_degree * 8 * words(field)), field, num_cols, ext_degree, num_rows * blowup};
        check(wf_evaluate_columns_over(ctx.handle(), (int)field, ext_degree, data.data(), num_cols, col_stride(), log2_exact(num_rows, "rows"),
                                       log2_exact(blowup, "blowup factor"), domain_offset, out.data.data(), out.col_stride()), "wf_evaluate_columns_over");
        return out;
    }
    // ColMatrix::commit_to_rows (col_matrix.rs:262-286)
    MerkleTree commit_to_rows(Hash h) const {
        Context &ctx = data.ctx();
        DeviceBuffer leaves(ctx, num_rows * 32);
        check(wf_hash_columns(ctx.handle(), (int)h, (int)field, ext_degree, data.data(), num_cols, col_stride(), num_rows, leaves.data()), "wf_hash_columns");
        return MerkleTree(h, std::move(leaves), num_rows);
    }
};

// RowMatrix (row_matrix.rs:28-41): data[row * row_width + col], row_width = 8 * ceil(base columns / 8)
struct RowMatrix {
    DeviceBuffer data;
    Field field = Field::F64;
    uint64_t num_rows = 0, row_width = 0;
    uint32_t elements_per_row = 0, ext_degree = 1;
    // RowMatrix::evaluate_polys_over::<8> (row_matrix.rs:84-100)
    static RowMatrix evaluate_polys_over(const ColMatrix &polys, uint64_t blowup, const uint64_t *domain_offset) {
        Context &ctx = polys.data.ctx();
        RowMatrix m;
        m.field = polys.field;
        m.ext_degree = polys.ext_degree;
        m.num_rows = polys.num_rows * blowup;
        m.row_width = wf_row_width(polys.num_cols, polys.ext_degree);
        m.elements_per_row = polys.num_cols * polys.ext_degree;
        m.data = DeviceBuffer(ctx, m.num_rows * m.row_width * 8 * words(m.field));
        check(wf_evaluate_polys_over(ctx.handle(), (int)m.field, m.ext_degree, polys.data.data(), polys.num_cols, polys.col_stride(),
                                     log2_exact(polys.num_rows, "rows"), log2_exact(blowup, "blowup factor"), domain_offset, m.data.data()),
              "wf_evaluate_polys_over");
        return m;
    }
    // RowMatrix::commit_to_rows (row_matrix.rs:184-228)
    MerkleTree commit_to_rows(Hash h, PartitionOptions po = {}) const {
        Context &ctx = data.ctx();
        DeviceBuffer leaves(ctx, num_rows * 32);
        check(wf_hash_rows(ctx.handle(), (int)h, (int)field, ext_degree, data.data(), num_rows, row_width, elements_per_row, po.num_partitions,
                           po.hash_rate, leaves.data()), "wf_hash_rows");
        return MerkleTree(h, std::move(leaves), num_rows);
    }
    // TraceLde::query row gather (trace_lde/default/mod.rs:199-215)
    std::vector<uint64_t> rows(const std::vector<uint64_t> &positions) const {
        std::vector<uint64_t> out(positions.size() * elements_per_row * words(field));
        if (!positions.empty())
            check(wf_rows_fetch(data.ctx().handle(), data.data(), row_width, elements_per_row, 8 * words(field), positions.data(), (uint32_t)positions.size(),
                                out.data()), "wf_rows_fetch");
        return out;
    }
};

// build_trace_commitment (trace_lde/default/mod.rs:245-282): interpolate -> coset LDE -> row hashes -> Merkle tree in one
// library call.  Returns (trace_lde, tree, trace_polys).
struct TraceCommitment {
    RowMatrix lde;
    MerkleTree tree;
    ColMatrix polys;
};
inline TraceCommitment build_trace_commitment(Hash h, const ColMatrix &trace, uint64_t blowup, const uint64_t *domain_offset,
                                              PartitionOptions po = {}, bool skip_interpolate = false) {
    Context &ctx = trace.data.ctx();
    ColMatrix polys{trace.data.clone(), trace.field, trace.num_cols, trace.ext_degree, trace.num_rows};
    RowMatrix lde;
    lde.field = trace.field;
    lde.ext_degree = trace.ext_degree;
    lde.num_rows = trace.num_rows * blowup;
    lde.row_width = wf_row_width(trace.num_cols, trace.ext_degree);
    lde.elements_per_row = trace.num_cols * trace.ext_degree;
    lde.data = DeviceBuffer(ctx, lde.num_rows * lde.row_width * 8 * words(trace.field));
    DeviceBuffer leaves(ctx, lde.num_rows * 32), nodes(ctx, lde.num_rows * 32);
    check(wf_build_trace_commitment(ctx.handle(), (int)h, (int)trace.field, trace.ext_degree, polys.data.data(), trace.num_cols, trace.col_stride(),
                                    log2_exact(trace.num_rows, "rows"), log2_exact(blowup, "blowup factor"), domain_offset, po.num_partitions,
                                    po.hash_rate, skip_interpolate ? 1 : 0, lde.data.data(), leaves.data(), nodes.data(), nullptr),
          "wf_build_trace_commitment");
    return TraceCommitment{std::move(lde), MerkleTree(h, std::move(leaves), std::move(nodes), trace.num_rows * blowup), std::move(polys)};
}

// Prover::new_trace_lde from HOST columns to a HOST TracePolyTable (prover/src/lib.rs:182-190), pipelined over PCIe:
//   uploader thread    column k -> HBM on its own context (= its own stream); page-locked columns go as one DMA each
//   calling thread     as soon as a group of `group` columns has landed: wf_interpolate_columns on it; after the last group
//                      wf_build_trace_commitment(skip_interpolate = 1): coset LDE + row hashes + tree
//   downloader thread  the polynomials of every interpolated group -> polys_out on a third context, WHILE later groups upload
//                      (PCIe is full duplex) and while the LDE runs (it only reads the polynomials)
// Leaves and nodes stay in HBM: openings go through MerkleTree::prove / prove_batch (device gathers) and RowMatrix::rows
// (wf_rows_fetch), as DefaultTraceLde::query needs (trace_lde/default/mod.rs:199-215).  The serial form — upload all, one call,
// download polys + leaves + nodes — moves n c s (in) + n c s + 64 b n (out) around the kernels; tools/host_pipeline_bench.cpp times both.
// `up` and `down` must be contexts of the same device as `ctx`; the three are used from three threads, one each.
inline TraceCommitment new_trace_lde_from_host(Context &ctx, Context &up, Context &down, Hash h, Field f, const std::vector<const uint64_t *> &cols,
                                               uint64_t num_rows, uint64_t blowup, const uint64_t *domain_offset, const std::vector<uint64_t *> &polys_out,
                                               PartitionOptions po = {}, uint32_t ext_degree = 1, uint32_t group = 8) {
    const uint32_t c = (uint32_t)cols.size(), log_n = log2_exact(num_rows, "rows");
    if (c == 0 || polys_out.size() != cols.size() || group == 0) throw Error(WF_ERR_INVALID_ARG, "new_trace_lde_from_host");
    const uint64_t stride = num_rows * ext_degree;                          // base elements per column
    const size_t col_bytes = (size_t)stride * 8 * words(f);
    ColMatrix polys{DeviceBuffer(ctx, (size_t)c * col_bytes), f, c, ext_degree, num_rows};
    uint8_t *d_cols = (uint8_t *)polys.data.data();
    std::atomic<uint32_t> uploaded{0}, interpolated{0};
    std::atomic<int> failed{0};
    auto wait_for = [&](std::atomic<uint32_t> &v, uint32_t want) {
        while (v.load(std::memory_order_acquire) < want && !failed.load(std::memory_order_acquire)) std::this_thread::yield();
        return !failed.load(std::memory_order_acquire);
    };
    std::thread uploader([&] {
        for (uint32_t k = 0; k < c; k++) {
            if (wf_memcpy_h2d(up.handle(), d_cols + (size_t)k * col_bytes, cols[k], col_bytes) != WF_OK) { failed.store(1); return; }
            uploaded.store(k + 1, std::memory_order_release);              // wf_memcpy_h2d returns after the copy
        }
    });
    std::thread downloader([&] {
        for (uint32_t k = 0; k < c; k++) {
            if (!wait_for(interpolated, k + 1)) return;
            if (wf_memcpy_d2h(down.handle(), polys_out[k], d_cols + (size_t)k * col_bytes, col_bytes) != WF_OK) { failed.store(2); return; }
        }
    });
    int status = WF_OK;
    RowMatrix lde;
    DeviceBuffer leaves, nodes;
    try {
        for (uint32_t g0 = 0; g0 < c && status == WF_OK; g0 += group) {
            const uint32_t cnt = c - g0 < group ? c - g0 : group;
            if (!wait_for(uploaded, g0 + cnt)) break;
            status = wf_interpolate_columns(ctx.handle(), (int)f, ext_degree, d_cols + (size_t)g0 * col_bytes, cnt, stride, log_n);
            if (status == WF_OK) status = wf_ctx_sync(ctx.handle());      // the downloader may read these columns from here on
            if (status == WF_OK) interpolated.store(g0 + cnt, std::memory_order_release);
        }
        if (status == WF_OK && !failed.load()) {
            lde.field = f;
            lde.ext_degree = ext_degree;
            lde.num_rows = num_rows * blowup;
            lde.row_width = wf_row_width(c, ext_degree);
            lde.elements_per_row = c * ext_degree;
            lde.data = DeviceBuffer(ctx, lde.num_rows * lde.row_width * 8 * words(f));
            leaves = DeviceBuffer(ctx, lde.num_rows * 32);
            nodes = DeviceBuffer(ctx, lde.num_rows * 32);
            status = wf_build_trace_commitment(ctx.handle(), (int)h, (int)f, ext_degree, d_cols, c, stride, log_n, log2_exact(blowup, "blowup factor"),
                                               domain_offset, po.num_partitions, po.hash_rate, 1, lde.data.data(), leaves.data(), nodes.data(), nullptr);
            if (status == WF_OK) status = wf_ctx_sync(ctx.handle());
        }
    } catch (...) {
        failed.store(3);
        uploader.join();
        downloader.join();
        throw;
    }
    if (status != WF_OK) failed.store(3);
    uploader.join();
    downloader.join();
    check(status, "new_trace_lde_from_host");
    if (failed.load()) throw Error(WF_ERR_HIP, failed.load() == 1 ? "new_trace_lde_from_host: upload" : "new_trace_lde_from_host: download");
    return TraceCommitment{std::move(lde), MerkleTree(h, std::move(leaves), std::move(nodes), num_rows * blowup), std::move(polys)};
}

// ---- multi-device: one rank per GPU (wf_comm_*) -------------------------------------------------------------------------------
// Comm: this rank's handle on the device interconnect.  Comm::rccl(ctx, id, rank, world) for one process (or thread) per GPU —
// rank 0 creates the 128-byte id with Comm::unique_id() and the host distributes it; Comm::loopback(contexts) for ranks that
// are threads of one process (no RCCL; also G logical ranks on one GPU).  sharded_commit is DefaultTraceLde::new with the
// columns sharded by partition (PartitionOptions::new(G, .), air/src/options.rs:391-451): the partition digests cross the
// interconnect in an all-to-all, the G sub-roots in an all-gather, everything else is local.
struct ShardedCommitment {
    RowMatrix lde;                  // this rank's columns over the whole LDE domain
    DeviceBuffer leaves, nodes;     // this rank's row range: N / G leaves, and the subtree over them (heap order)
    DeviceBuffer top;               // the top log2 G levels (G digests, heap order), identical on every rank
    ColMatrix polys;
    std::vector<uint8_t> root;
};

class Comm {
public:
    static std::vector<uint8_t> unique_id() {
        std::vector<uint8_t> id(WF_COMM_ID_BYTES);
        check(wf_comm_get_unique_id(id.data()), "wf_comm_get_unique_id");
        return id;
    }
    static Comm rccl(Context &ctx, const std::vector<uint8_t> &id, int rank, int world) {
        wf_comm *c = nullptr;
        check(wf_comm_init_rank(ctx.handle(), id.data(), rank, world, &c), "wf_comm_init_rank");
        return Comm(c, &ctx);
    }
    static std::vector<Comm> loopback(const std::vector<Context *> &ctxs) {
        std::vector<wf_ctx *> h;
        for (Context *c : ctxs) h.push_back(c->handle());
        std::vector<wf_comm *> out(ctxs.size());
        check(wf_comm_init_loopback(h.data(), (int)h.size(), out.data()), "wf_comm_init_loopback");
        std::vector<Comm> r;
        for (size_t i = 0; i < out.size(); i++) r.emplace_back(Comm(out[i], ctxs[i]));
        return r;
    }
    Comm(Comm &&o) noexcept : c_(o.c_), ctx_(o.ctx_) { o.c_ = nullptr; }
    Comm(const Comm &) = delete;
    ~Comm() {
        if (c_) wf_comm_destroy(c_);
    }
    int rank() const { return wf_comm_rank(c_); }
    int size() const { return wf_comm_size(c_); }
    void all_gather(const void *d_send, void *d_recv, uint64_t bytes) { check(wf_comm_all_gather(c_, d_send, d_recv, bytes), "wf_comm_all_gather"); }
    void all_to_all(const void *d_send, void *d_recv, uint64_t bytes) { check(wf_comm_all_to_all(c_, d_send, d_recv, bytes), "wf_comm_all_to_all"); }

    // `shard`: this rank's partition of the trace columns (evaluations over the trace domain)
    ShardedCommitment sharded_commit(Hash h, const ColMatrix &shard, uint64_t blowup, const uint64_t *domain_offset) {
        Context &ctx = *ctx_;
        const uint64_t N = shard.num_rows * blowup, per = N / (uint64_t)size();
        ShardedCommitment r{RowMatrix{}, DeviceBuffer(ctx, per * 32), DeviceBuffer(ctx, per * 32), DeviceBuffer(ctx, (size_t)size() * 32),
                            ColMatrix{shard.data.clone(), shard.field, shard.num_cols, shard.ext_degree, shard.num_rows}, std::vector<uint8_t>(32)};
        r.lde.field = shard.field;
        r.lde.ext_degree = shard.ext_degree;
        r.lde.num_rows = N;
        r.lde.elements_per_row = shard.num_cols * shard.ext_degree;
        r.lde.row_width = wf_row_width(shard.num_cols, shard.ext_degree);
        r.lde.data = DeviceBuffer(ctx, N * r.lde.row_width * 8 * words(shard.field));
        check(wf_comm_sharded_commit(c_, (int)h, (int)shard.field, shard.ext_degree, r.polys.data.data(), shard.num_cols, shard.col_stride(),
                                     log2_exact(shard.num_rows, "rows"), log2_exact(blowup, "blowup factor"), domain_offset, 0, r.lde.data.data(),
                                     r.leaves.data(), r.nodes.data(), r.top.data(), r.root.data()),
              "wf_comm_sharded_commit");
        return r;
    }

    // FriProver::build_layers with the layers sharded by contiguous row ranges (wf_comm_sharded_fri_layers): this rank's piece of
    // the evaluations in, per layer this rank's rows / leaves / subtree, the top tree, its piece of the next layer; roots and alphas
    // (device) are identical on every rank.  `coin`: this rank's copy of the device coin (64 bytes, WF_COIN_BYTES).
    struct ShardedFriLayer {
        DeviceBuffer rows, leaves, nodes, top, folded;
    };
    struct ShardedFri {
        std::vector<ShardedFriLayer> layers;
        DeviceBuffer roots, alphas;
    };
    ShardedFri sharded_fri_layers(Hash h, Field f, uint32_t ext_degree, const DeviceBuffer &piece, uint64_t length, uint32_t folding, uint32_t num_layers,
                                  const uint64_t *domain_offset, void *d_coin) {
        Context &ctx = *ctx_;
        const size_t eb = (size_t)ext_degree * words(f) * 8;
        ShardedFri out;
        std::vector<void *> p_rows, p_leaves, p_nodes, p_top, p_folded;
        uint64_t len = length;
        for (uint32_t k = 0; k < num_layers; k++) {
            const uint64_t rl = len / folding / (uint64_t)size();
            out.layers.push_back(ShardedFriLayer{DeviceBuffer(ctx, rl * folding * eb), DeviceBuffer(ctx, rl * 32), DeviceBuffer(ctx, rl * 32),
                                                 DeviceBuffer(ctx, (size_t)size() * 32), DeviceBuffer(ctx, rl * eb)});
            ShardedFriLayer &l = out.layers.back();
            p_rows.push_back(l.rows.data());
            p_leaves.push_back(l.leaves.data());
            p_nodes.push_back(l.nodes.data());
            p_top.push_back(l.top.data());
            p_folded.push_back(l.folded.data());
            len /= folding;
        }
        out.roots = DeviceBuffer(ctx, (size_t)num_layers * 32);
        out.alphas = DeviceBuffer(ctx, (size_t)num_layers * eb);
        check(wf_comm_sharded_fri_layers(c_, (int)h, (int)f, ext_degree, piece.data(), log2_exact(length, "evaluations"), folding, num_layers, domain_offset,
                                         d_coin, p_rows.data(), p_leaves.data(), p_nodes.data(), p_top.data(), p_folded.data(), out.roots.data(),
                                         out.alphas.data()),
              "wf_comm_sharded_fri_layers");
        return out;
    }

private:
    Comm(wf_comm *c, Context *ctx) : c_(c), ctx_(ctx) {}
    wf_comm *c_;
    Context *ctx_;
};

// ---- fri::FriProver (commit phase) -------------------------------------------------------------------------------------------
// Context::to_elements (air/src/proof/context.rs:106-137) as canonical integers — what the prover channel seeds its coin with,
// followed by PublicInputs::to_elements() (prover/src/channel.rs:57-75).  TraceInfo::to_elements (air/src/air/trace_info.rs:209-238):
// segment widths packed 8 bits each, then the trace length; the field modulus' little-endian bytes split into two elements;
// the number of constraints; ProofOptions::to_elements (air/src/options.rs:294-305) with FieldExtension None / Quadratic / Cubic
// = 1 / 2 / 3 = the extension degree.  (64-bit fields; no auxiliary segment, no trace metadata: the built-in AIRs have neither.)
inline std::vector<uint64_t> context_to_elements_f64(uint32_t main_width, uint64_t trace_length, uint64_t modulus, uint32_t num_constraints,
                                                     uint32_t ext_degree, uint32_t fri_folding_factor, uint32_t fri_remainder_max_degree,
                                                     uint32_t blowup_factor, uint32_t grinding_factor, uint32_t num_queries) {
    const uint64_t options = ((((((uint64_t)ext_degree << 8) | fri_folding_factor) << 8) | fri_remainder_max_degree) << 8) | blowup_factor;
    return {(uint64_t)main_width << 8, trace_length & 0xffffffffull, modulus & 0xffffffffull, modulus >> 32, num_constraints, options,
            grinding_factor, num_queries};
}

// crypto::DefaultRandomCoin with its state on the device (wf_coin_*; crypto/src/random/default.rs): the hand-over point between a
// host coin and a chain of device stages that reseed and draw (FriProver::build_layers below)
class DeviceCoin {
  public:
    DeviceCoin(Context &ctx, Hash h, Field f, const uint8_t seed[32]) : hash_(h), field_(f), state_(ctx, WF_COIN_BYTES) {
        check(wf_coin_init(ctx.handle(), state_.data(), seed), "wf_coin_init");
    }
    // RandomCoin::reseed with a digest that is on the device; the digest is also copied to d_copy when given
    void reseed(const void *d_digest, void *d_copy = nullptr) {
        check(wf_coin_reseed(state_.ctx().handle(), (int)hash_, state_.data(), d_digest, d_copy), "wf_coin_reseed");
    }
    // `count` x RandomCoin::draw::<E>: count * ext_degree * W words on the device
    DeviceBuffer draw(uint32_t ext_degree, uint32_t count = 1) {
        DeviceBuffer out(state_.ctx(), (size_t)count * ext_degree * words(field_) * 8);
        check(wf_coin_draw(state_.ctx().handle(), (int)hash_, (int)field_, ext_degree, state_.data(), count, out.data()), "wf_coin_draw");
        return out;
    }
    // the state after everything queued so far (waits for the stream); throws if a draw ran out of its 1000 tries
    std::pair<std::array<uint8_t, 32>, uint64_t> read() const {
        std::array<uint8_t, 32> seed{};
        uint64_t counter = 0;
        check(wf_coin_read(state_.ctx().handle(), state_.data(), seed.data(), &counter), "wf_coin_read");
        return {seed, counter};
    }
    void *state() const { return state_.data(); }
    Hash hash() const { return hash_; }

  private:
    Hash hash_;
    Field field_;
    DeviceBuffer state_;
};

// fri::ProverChannel (fri/src/prover/channel.rs:24-50): the host Fiat-Shamir transcript, supplied by the caller
struct ProverChannel {
    virtual ~ProverChannel() = default;
    virtual void commit_fri_layer(const uint8_t root[32]) = 0;
    virtual std::vector<uint64_t> draw_fri_alpha() = 0;   // one element of the extension field (ext_degree * W words)
};

struct FriOptions {   // fri/src/options.rs:13-93
    uint64_t blowup_factor, folding_factor, remainder_max_degree;
    uint64_t num_fri_layers(uint64_t domain_size) const {
        uint64_t result = 0;
        const uint64_t max_rem = (remainder_max_degree + 1) * blowup_factor;
        while (domain_size > max_rem) {
            domain_size /= folding_factor;
            result++;
        }
        return result;
    }
};

struct FriLayer {   // fri/src/prover/mod.rs:111-115
    MerkleTree commitment;
    DeviceBuffer evaluations;   // transposed: [len / folding][folding * ext_degree * W]
};

class FriProver {
  public:
    FriProver(FriOptions o, Hash h, Field f, uint32_t ext_degree, std::vector<uint64_t> domain_offset)
        : opts_(o), hash_(h), field_(f), D_(ext_degree), offset_(std::move(domain_offset)) {}
    // FriProver::build_layers (mod.rs:179-199); `evaluations` is consumed
    void build_layers(ProverChannel &channel, DeviceBuffer evaluations, uint64_t length) {
        if (!layers_.empty()) throw std::logic_error("a prior proof generation request has not been completed yet");
        Context &ctx = evaluations.ctx();
        const uint32_t ew = D_ * words(field_);
        const uint64_t N = opts_.folding_factor;
        for (uint64_t k = opts_.num_fri_layers(length); k > 0; k--) {
            const uint32_t log_len = log2_exact(length, "evaluations");
            const uint64_t rows = length / N;
            DeviceBuffer transposed(ctx, rows * N * ew * 8), leaves(ctx, rows * 32), nodes(ctx, rows * 32);
            uint8_t root[32];
            check(wf_fri_layer_commit(ctx.handle(), (int)hash_, (int)field_, D_, evaluations.data(), log_len, (uint32_t)N, transposed.data(), leaves.data(),
                                      nodes.data(), root), "wf_fri_layer_commit");
            channel.commit_fri_layer(root);
            const std::vector<uint64_t> alpha = channel.draw_fri_alpha();
            DeviceBuffer folded(ctx, rows * ew * 8);
            check(wf_fri_apply_drp(ctx.handle(), (int)field_, D_, transposed.data(), log_len, (uint32_t)N, offset_.data(), alpha.data(), folded.data()),
                  "wf_fri_apply_drp");
            layers_.push_back(FriLayer{MerkleTree(hash_, std::move(leaves), std::move(nodes), rows), std::move(transposed)});
            evaluations = std::move(folded);
            length = rows;
        }
        set_remainder(channel, evaluations, length);
    }
    // The same with the channel's coin on the device (wf_fri_build_layers): every layer's commit, reseed, draw and fold is queued
    // back to back; what the channel would have recorded comes back at the end — the layer commitments (then the remainder's)
    // in `roots`, the folding challenges in `alphas` — and the coin has absorbed all of them.
    struct Transcript {
        std::vector<std::array<uint8_t, 32>> roots;
        std::vector<uint64_t> alphas;      // one E element (ext_degree * W words) per layer
    };
    Transcript build_layers(DeviceCoin &coin, DeviceBuffer evaluations, uint64_t length) {
        if (!layers_.empty()) throw std::logic_error("a prior proof generation request has not been completed yet");
        Context &ctx = evaluations.ctx();
        const uint32_t ew = D_ * words(field_);
        const uint64_t N = opts_.folding_factor;
        const uint32_t nl = (uint32_t)opts_.num_fri_layers(length);
        Transcript tr;
        std::vector<DeviceBuffer> transposed, leaves, nodes, folded;
        std::vector<void *> p_tr, p_lv, p_nd, p_fo;
        uint64_t rows = length;
        for (uint32_t k = 0; k < nl; k++) {
            rows /= N;
            transposed.emplace_back(ctx, rows * N * ew * 8);
            leaves.emplace_back(ctx, rows * 32);
            nodes.emplace_back(ctx, rows * 32);
            folded.emplace_back(ctx, rows * ew * 8);
            p_tr.push_back(transposed.back().data());
            p_lv.push_back(leaves.back().data());
            p_nd.push_back(nodes.back().data());
            p_fo.push_back(folded.back().data());
        }
        if (nl) {
            DeviceBuffer d_roots(ctx, (size_t)(nl + 1) * 32), d_alphas(ctx, (size_t)nl * ew * 8);
            check(wf_fri_build_layers(ctx.handle(), (int)hash_, (int)field_, D_, evaluations.data(), log2_exact(length, "evaluations"), (uint32_t)N, nl,
                                      offset_.data(), coin.state(), p_tr.data(), p_lv.data(), p_nd.data(), p_fo.data(), d_roots.data(), d_alphas.data(),
                                      0, nullptr),     // the remainder step below goes through this class's own set_remainder
                  "wf_fri_build_layers");
            const std::vector<uint8_t> r = d_roots.to_host<uint8_t>();
            tr.roots.resize(nl);
            for (uint32_t k = 0; k < nl; k++) std::memcpy(tr.roots[k].data(), &r[(size_t)k * 32], 32);
            tr.alphas = d_alphas.to_host<uint64_t>();
            uint64_t rk = length;
            for (uint32_t k = 0; k < nl; k++) {
                rk /= N;
                layers_.push_back(FriLayer{MerkleTree(hash_, std::move(leaves[k]), std::move(nodes[k]), rk), std::move(transposed[k])});
            }
            evaluations = std::move(folded[nl - 1]);
            length = rows;
        }
        struct CoinSink : ProverChannel {      // the remainder commitment goes into the device coin as well (mod.rs:230-239)
            DeviceCoin &coin;
            Context &ctx;
            Transcript &tr;
            CoinSink(DeviceCoin &c, Context &x, Transcript &t) : coin(c), ctx(x), tr(t) {}
            void commit_fri_layer(const uint8_t root[32]) override {
                DeviceBuffer d(ctx, root, 32);
                coin.reseed(d.data());
                check(wf_ctx_sync(ctx.handle()), "wf_ctx_sync");     // `d` is released when this returns
                std::array<uint8_t, 32> a{};
                std::memcpy(a.data(), root, 32);
                tr.roots.push_back(a);
            }
            std::vector<uint64_t> draw_fri_alpha() override { throw std::logic_error("not part of set_remainder"); }
        } sink(coin, ctx, tr);
        set_remainder(sink, evaluations, length);
        return tr;
    }
    const std::vector<FriLayer> &layers() const { return layers_; }
    // fri::folding::fold_positions (fri/src/folding/mod.rs:159-176)
    static std::vector<uint64_t> fold_positions(const std::vector<uint64_t> &positions, uint64_t source_domain_size, uint64_t folding_factor) {
        const uint64_t target = source_domain_size / folding_factor;
        std::vector<uint64_t> out;
        for (uint64_t p : positions) {
            const uint64_t q = p % target;
            if (std::find(out.begin(), out.end(), q) == out.end()) out.push_back(q);
        }
        return out;
    }
    // one layer of FriProver::build_proof (mod.rs:253-317): the queried rows (N evaluations each, fold_positions order) and
    // the batch opening against the layer commitment
    struct ProofLayer {
        std::vector<uint64_t> values;
        MerkleTree::BatchProof proof;
    };
    struct Proof {
        std::vector<ProofLayer> layers;
        std::vector<uint64_t> remainder;
        uint32_t num_partitions = 1;
    };
    // FriProver::build_proof: query phase; clears the layers (mod.rs:283-289)
    Proof build_proof(std::vector<uint64_t> positions) {
        if (remainder_.empty()) throw std::logic_error("FRI layers have not been built yet");
        Proof out;
        const uint64_t N = opts_.folding_factor;
        const uint32_t row_words = (uint32_t)(N * D_ * words(field_));
        uint64_t domain_size = layers_.empty() ? 0 : layers_[0].commitment.num_leaves() * N;
        for (const FriLayer &layer : layers_) {
            positions = fold_positions(positions, domain_size, N);
            ProofLayer pl;
            pl.proof = layer.commitment.prove_batch(positions).second;
            pl.values.resize(positions.size() * row_words);
            check(wf_rows_fetch(layer.evaluations.ctx().handle(), layer.evaluations.data(), N * D_, (uint32_t)(N * D_), 8 * words(field_), positions.data(),
                                (uint32_t)positions.size(), pl.values.data()), "wf_rows_fetch");
            out.layers.push_back(std::move(pl));
            domain_size /= N;
        }
        out.remainder = remainder_;
        layers_.clear();
        remainder_.clear();
        return out;
    }
    // reversed coefficients of the remainder polynomial (mod.rs:230-239), ext_degree * W words each
    const std::vector<uint64_t> &remainder_poly() const { return remainder_; }

  private:
    void set_remainder(ProverChannel &channel, DeviceBuffer &ev, uint64_t length) {
        Context &ctx = ev.ctx();
        const uint32_t ew = D_ * words(field_);
        if (length > 1) fft::interpolate_poly_with_offset(ev, field_, length, offset_.data(), D_);
        std::vector<uint64_t> coeffs = ev.to_host<uint64_t>();
        const uint64_t size = length / opts_.blowup_factor;
        remainder_.assign(size * ew, 0);
        for (uint64_t i = 0; i < size; i++) std::memcpy(&remainder_[i * ew], &coeffs[(size - 1 - i) * ew], ew * 8);
        DeviceBuffer d_rem(ctx, remainder_), d_out(ctx, 32);
        check(wf_hash_elements_batch(ctx.handle(), (int)hash_, (int)field_, d_rem.data(), 1, size * D_, (uint32_t)(size * D_), d_out.data()),
              "wf_hash_elements_batch");
        uint8_t com[32];
        d_out.download(com, 32);
        channel.commit_fri_layer(com);
    }
    FriOptions opts_;
    Hash hash_;
    Field field_;
    uint32_t D_;
    std::vector<uint64_t> offset_;
    std::vector<FriLayer> layers_;
    std::vector<uint64_t> remainder_;
};

// ---- SURVEY §8(f) N1–N3 ------------------------------------------------------------------------------------------------------
// ProverChannel::grind_query_seed (prover/src/channel.rs:169-185): smallest nonce >= 1 with enough trailing zeros
inline uint64_t grind_query_seed(Context &ctx, Hash h, const uint8_t seed[32], uint32_t grinding_factor) {
    uint64_t nonce = 0;
    check(wf_grind(ctx.handle(), (int)h, seed, grinding_factor, 1, ~0ull - 1, &nonce), "wf_grind");
    return nonce;
}

// TracePolyTable::get_ood_frame / CompositionPoly::get_ood_frame: all columns at the given points;
// out[point][column] elements of ext_degree * W words
inline std::vector<uint64_t> evaluate_columns_at(const ColMatrix &polys, const std::vector<uint64_t> &points, uint32_t num_points, uint32_t ext_degree) {
    std::vector<uint64_t> out((size_t)num_points * polys.num_cols * ext_degree * words(polys.field));
    check(wf_polys_evaluate_at(polys.data.ctx().handle(), (int)polys.field, polys.ext_degree, ext_degree, polys.data.data(), polys.num_cols, polys.col_stride(),
                               log2_exact(polys.num_rows, "rows"), points.data(), num_points, out.data()), "wf_polys_evaluate_at");
    return out;
}

// DeepCompositionPoly::add_trace_polys (prover/src/composer/mod.rs:67-169) -> coefficients (num_rows elements over E)
inline DeviceBuffer deep_compose(const ColMatrix &main_polys, const ColMatrix *aux_polys, const ColMatrix &quotient_polys, uint32_t ext_degree,
                                 const std::vector<uint64_t> &z, const std::vector<uint64_t> &cc_trace, const std::vector<uint64_t> &cc_constraints) {
    Context &ctx = main_polys.data.ctx();
    DeviceBuffer out(ctx, main_polys.num_rows * ext_degree * 8 * words(main_polys.field));
    check(wf_deep_compose(ctx.handle(), (int)main_polys.field, ext_degree, main_polys.data.data(), main_polys.num_cols, main_polys.col_stride(),
                          aux_polys ? aux_polys->data.data() : nullptr, aux_polys ? aux_polys->num_cols : 0, aux_polys ? aux_polys->col_stride() : 0,
                          quotient_polys.data.data(), quotient_polys.num_cols, quotient_polys.col_stride(), log2_exact(main_polys.num_rows, "rows"), z.data(),
                          cc_trace.data(), cc_constraints.data(), out.data()), "wf_deep_compose");
    return out;
}

// Assertion::single + DefaultConstraintEvaluator::evaluate for the built-in AIRs (wf_evaluate_constraints)
struct Assertion {
    uint32_t column;
    uint64_t step;
    std::vector<uint64_t> value;   // one base element (W words)
};
inline DeviceBuffer evaluate_constraints(int air, const RowMatrix &trace_lde, uint64_t trace_length, uint64_t lde_blowup, uint64_t ce_blowup,
                                         const uint64_t *domain_offset, uint32_t ext_degree, const std::vector<uint64_t> &cc_transition,
                                         const std::vector<Assertion> &assertions, const std::vector<uint64_t> &cc_boundary) {
    Context &ctx = trace_lde.data.ctx();
    const uint32_t W = words(trace_lde.field);
    std::vector<uint32_t> cols;
    std::vector<uint64_t> steps, vals;
    for (const Assertion &a : assertions) {
        cols.push_back(a.column);
        steps.push_back(a.step);
        vals.insert(vals.end(), a.value.begin(), a.value.begin() + W);
    }
    DeviceBuffer out(ctx, trace_length * ce_blowup * ext_degree * 8 * W);
    check(wf_evaluate_constraints(ctx.handle(), air, (int)trace_lde.field, ext_degree, trace_lde.data.data(), trace_lde.row_width,
                                  log2_exact(trace_length, "rows"), log2_exact(lde_blowup, "blowup factor"), log2_exact(ce_blowup, "blowup factor"),
                                  domain_offset, cc_transition.data(), (uint32_t)assertions.size(), cols.data(), steps.data(), vals.data(), cc_boundary.data(),
                                  out.data()), "wf_evaluate_constraints");
    return out;
}

// the same for a trace with an auxiliary segment (wf_evaluate_constraints_aux; evaluate_fragment_full, default.rs:214-271):
// aux assertion values and the segment's random elements are elements of E (ext_degree * W words each); cc_transition = the main
// constraints' coefficients, then the auxiliary ones; cc_boundary / cc_aux_boundary per assertion.
inline DeviceBuffer evaluate_constraints_aux(int air, const RowMatrix &main_lde, const RowMatrix &aux_lde, uint64_t trace_length, uint64_t lde_blowup,
                                             uint64_t ce_blowup, const uint64_t *domain_offset, uint32_t ext_degree,
                                             const std::vector<uint64_t> &cc_transition, const std::vector<Assertion> &assertions,
                                             const std::vector<uint64_t> &cc_boundary, const std::vector<Assertion> &aux_assertions,
                                             const std::vector<uint64_t> &cc_aux_boundary, const std::vector<uint64_t> &aux_rand_elements) {
    Context &ctx = main_lde.data.ctx();
    const uint32_t W = words(main_lde.field);
    std::vector<uint32_t> cols, xcols;
    std::vector<uint64_t> steps, vals, xsteps, xvals;
    for (const Assertion &a : assertions) {
        cols.push_back(a.column);
        steps.push_back(a.step);
        vals.insert(vals.end(), a.value.begin(), a.value.begin() + W);
    }
    for (const Assertion &a : aux_assertions) {
        xcols.push_back(a.column);
        xsteps.push_back(a.step);
        xvals.insert(xvals.end(), a.value.begin(), a.value.begin() + (size_t)ext_degree * W);
    }
    DeviceBuffer out(ctx, trace_length * ce_blowup * ext_degree * 8 * W);
    check(wf_evaluate_constraints_aux(ctx.handle(), air, (int)main_lde.field, ext_degree, main_lde.data.data(), main_lde.row_width, aux_lde.data.data(),
                                      aux_lde.row_width, log2_exact(trace_length, "rows"), log2_exact(lde_blowup, "blowup factor"),
                                      log2_exact(ce_blowup, "blowup factor"), domain_offset, cc_transition.data(), (uint32_t)assertions.size(), cols.data(),
                                      steps.data(), vals.data(), cc_boundary.data(), (uint32_t)aux_assertions.size(), xcols.data(), xsteps.data(),
                                      xvals.data(), cc_aux_boundary.data(), aux_rand_elements.data(), out.data()),
          "wf_evaluate_constraints_aux");
    return out;
}

}  // namespace wf
