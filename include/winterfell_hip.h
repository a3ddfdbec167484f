/*
 * winterfell_hip.h — C ABI of libwinterfell_hip.so, the MI355X (gfx950) implementation of Winterfell's
 * STARK proving hot path: math::fft NTTs / coset LDE and the crypto::merkle commitment layer.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / C++ types.  Each entry point names
 * the reference interface it stands behind (paths relative to the reference repository root).  The
 * Rust-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions:
 *  - Every function returns an `int` status: WF_OK (0) or a WF_ERR_* code; nothing aborts.  The Rust shim maps
 *    non-zero to panic! for fft/matrix preconditions (math/src/fft/mod.rs:90-102 assert!s) and to
 *    Err(MerkleTreeError::..) for the vector commitment (crypto/src/merkle/mod.rs:117-122).
 *  - Memory representation is exactly the reference's: f64 / f62 elements are u64 Montgomery residues
 *    (math/src/field/f64/mod.rs:60), f128 elements are little-endian canonical u128; an extension element
 *    is `ext_degree` consecutive base elements (math/src/field/extensions/quadratic.rs:30-33, cubic.rs:30-33);
 *    Blake3 digests are 32 raw bytes (crypto/src/hash/mod.rs:85-114), Rp64_256 digests are 4 u64 Montgomery
 *    residues (crypto/src/hash/rescue/rp64_256/digest.rs:16); Merkle `nodes` are in the reference heap
 *    order: nodes[0] = zero digest, nodes[1] = root, children of i at 2i, 2i+1 (crypto/src/merkle/mod.rs:344-368).
 *  - Pointers named d_* are DEVICE pointers (from wf_malloc or any HIP allocation, e.g. a torch tensor);
 *    pointers named h_* are host pointers.  Work is enqueued on the context's stream; results are
 *    complete after wf_ctx_sync() (functions that return host values synchronise themselves).
 *  - Every entry point locks its context for the duration of the call: calls on ONE context from several host threads are safe and
 *    run one after the other (TraceLde: Sync — the reference reads frames from Rayon workers, prover/src/constraints/evaluator/
 *    default.rs:187).  For calls that should run concurrently use one context per thread (each has its own stream, caches, pool).
 */
#ifndef WINTERFELL_HIP_H
#define WINTERFELL_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wf_ctx wf_ctx;

/* ---- status codes ---------------------------------------------------------------------------------- */
enum {
    WF_OK = 0,
    WF_ERR_INVALID_ARG = 1,          /* null pointer, zero size, bad enum ...                             */
    WF_ERR_NOT_POWER_OF_TWO = 2,     /* fft/mod.rs:90 "number of coefficients must be a power of 2";
                                        MerkleTreeError::NumberOfLeavesNotPowerOfTwo (merkle/mod.rs:120)  */
    WF_ERR_TOO_FEW_LEAVES = 3,       /* MerkleTreeError::TooFewLeaves (merkle/mod.rs:117)                 */
    WF_ERR_DOMAIN_TOO_LARGE = 4,     /* fft/mod.rs:96-100 "multiplicative subgroup of size .. does not exist" */
    WF_ERR_UNSUPPORTED = 5,          /* field / hash / extension degree combination not available         */
    WF_ERR_HIP = 6,                  /* a HIP runtime call failed; see wf_last_hip_error()                */
    WF_ERR_NO_DEVICE = 7,
    WF_ERR_ZERO_OFFSET = 8,          /* fft/mod.rs:185 "domain offset cannot be zero"                     */
    WF_ERR_NOT_FOUND = 9,            /* prover/src/channel.rs:175 "nonce not found"                        */
    WF_ERR_COMM_ABORTED = 10,        /* a peer rank of a loopback wf_comm failed: the collective was abandoned */
    WF_ERR_DEVICE_STATUS = 11        /* a kernel flagged a protocol failure in the context's status word (a Merkle ticket word that was
                                        not in the state the launch expects: results of the call are not to be trusted); returned by the
                                        next synchronising call, bits in wf_last_device_status()                          */
};

/* ---- enums ----------------------------------------------------------------------------------------- */
enum { WF_FIELD_F64 = 0, WF_FIELD_F128 = 1, WF_FIELD_F62 = 2 };   /* math/src/field/{f64,f128,f62}       */
enum {                                   /* crypto/src/hash/... */
    WF_HASH_BLAKE3_256 = 0,              /* blake/mod.rs:24-66                                  */
    WF_HASH_RP64_256 = 1,                /* rescue/rp64_256/mod.rs (f64 only)                   */
    WF_HASH_SHA3_256 = 2,                /* sha/mod.rs:21-66                                    */
    WF_HASH_RPJIVE64_256 = 3,            /* rescue/rp64_256_jive/mod.rs (f64 only, Jive 2-to-1) */
    WF_HASH_RP62_248 = 4,                /* rescue/rp62_248/mod.rs (f62 only)                   */
    WF_HASH_BLAKE3_192 = 5               /* blake/mod.rs:68-125: 24-byte digests in 32-byte slots, bytes 24..31 zero */
};

/* ---- context / memory ------------------------------------------------------------------------------ */
int wf_version(void);
const char *wf_strerror(int status);
int wf_device_count(int *h_count);
/* Environment variables, each read ONCE when a context is created (results never depend on them; they select between kernels that
 * compute the same words and exist for A/B measurements, tools/time_two_pass.py):
 *   WF_NTT_BIG=0|1           two-pass f64 NTT plans (three-step passes of radix 2^10 .. 2^12, csrc/ntt_big.cuh): never | for every
 *                            eligible transform of 2^20 .. 2^24 points; unset = where measured faster (single 2^21 / 2^22-point
 *                            transforms, batches of >= 8 vectors of 2^20 points, batches of >= 256 vectors of 2^21 points)
 *   WF_NTT_F64_TABLES=0|1    inter-pass twiddles of the f64 passes from one-word tables: never | wherever a table fits the cache budget;
 *                            unset = where measured faster (batches of vectors of at most 2^19 points)
 *   WF_NTT_BT=0|1            block tiles for single three-pass f64 transforms (transposed first pass, tile-shared twiddles in the middle
 *                            pass): never | wherever eligible; unset = 2^23-point transforms, where measured faster
 *   WF_LDE_VT=0              the coset LDE of a wide f64 trace (a multiple of 16 / 32 columns) runs on position-major tiles with per-lane
 *                            twiddle progressions instead of vector tiles with tile-shared twiddle rows (csrc/ntt_engine.cuh, VT)
 *   WF_ROWS_HASH_WIDE=0      rows of 9 .. 32 f64 columns are hashed by the separate row-hash kernel, not by the last NTT pass
 *   WF_NTT_COSET_ORDER=0     the first pass of a coset LDE walks its tiles vector by vector (no L2 sharing of the source tile)
 *   WF_NTT_PLAN=L:r1,r2,..   pass radices for transforms of 2^L points (measurements)
 *   WF_DEBUG_GUARD=1|2       guard pages / red zones around every device block (tests/README_guard.md) */
int wf_ctx_create(int device_id, wf_ctx **out);
/* The same on a stream the caller owns from the start (a host that always runs on its framework's stream: no private stream is
 * created only to be destroyed by the first wf_ctx_set_stream). */
int wf_ctx_create_on_stream(int device_id, void *hip_stream, wf_ctx **out);
int wf_ctx_destroy(wf_ctx *ctx);
/* Use a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own stream. */
int wf_ctx_set_stream(wf_ctx *ctx, void *hip_stream);
int wf_ctx_get_stream(wf_ctx *ctx, void **hip_stream);
int wf_ctx_sync(wf_ctx *ctx);
int wf_last_hip_error(wf_ctx *ctx);
uint32_t wf_last_device_status(wf_ctx *ctx);

/* Measurement hook (no reference counterpart; plays the role of the reference's tracing spans,
 * prover/src/trace/trace_lde/default/mod.rs:258,277): when enabled every kernel launch is bracketed by HIP events
 * on the context's stream; wf_prof_collect synchronises and writes "kernel_name launches total_ms" lines.  on = 2: no brackets, one
 * event before the first launch and one after the last since the previous collect: a single line "__span__ launches ms", the device
 * time of a multi-launch call without the bracket overhead (a bracket adds ~2-4 us to its launch). */
int wf_prof_enable(wf_ctx *ctx, int on);
int wf_prof_collect(wf_ctx *ctx, char *h_buf, size_t buf_len);
/* Measurement hook: the shader clock in MHz while the work queued on the context's stream runs — one wavefront on a stream of its
 * own counts shader cycles (s_memtime) over `spin_us` microseconds of the constant-rate clock (s_memrealtime).  The part clocks to its
 * power budget; bench.py quotes its issue ceilings at this clock, not at the 2.4 GHz maximum. */
int wf_debug_shader_clock(wf_ctx *ctx, uint32_t spin_us, double *h_mhz);
/* Debug allocator (WF_DEBUG_GUARD=1 guard pages / 2 red zones, read once per process; 0 = off): every device allocation of the library
 * — and, through torch's pluggable allocator pointed at the two functions below, of a test session — becomes a block of its own with
 * unmapped pages right behind its last byte.  tests/conftest.py, tools/guard_session.sh. */
void *wf_debug_torch_malloc(ssize_t size, int device, void *stream);
void wf_debug_torch_free(void *ptr, ssize_t size, int device, void *stream);
int wf_debug_guard_mode(void);
/* Test hook: overwrite the ticket word the next one-launch Merkle tree of this context will use (the failure-detection test). */
int wf_debug_poke_tree_ticket(wf_ctx *ctx, uint32_t value);

/* Device memory for the caller's buffers.  wf_free does not synchronise: blocks go to a per-context pool and are handed out
 * again by later wf_malloc calls — safe because everything a context does is ordered on its stream (a buffer shared with
 * another stream must be synchronised by the caller before it is freed).  wf_free accepts live blocks of THIS context's pool only
 * (anything else — another context's block, a double free — is WF_ERR_INVALID_ARG; it is never passed on to hipFree).  wf_ctx_trim
 * returns the cached blocks to the driver; wf_ctx_destroy does so implicitly, blocks the caller never freed included.
 * wf_memcpy_h2d / wf_memcpy_d2h return after the copy.  Pageable host memory goes through page-locked bounce buffers inside the
 * library (the runtime is never handed a pageable range: it would pin it in place and find the pinned object again by address after
 * the buffer was freed and re-allocated — a GPU fault at a host address); ranges page-locked by wf_host_register /
 * hipHostMalloc are copied directly at PCIe rate. */
int wf_malloc(wf_ctx *ctx, size_t bytes, void **d_ptr);
int wf_free(wf_ctx *ctx, void *d_ptr);
int wf_ctx_trim(wf_ctx *ctx);
int wf_memcpy_h2d(wf_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int wf_memcpy_d2h(wf_ctx *ctx, void *h_dst, const void *d_src, size_t bytes); /* synchronises */
int wf_memcpy_d2d(wf_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* Page-lock a host buffer the caller owns (a Rust Vec's allocation: ColMatrix columns going in, RowMatrix / polys /
 * nodes coming out) so that wf_memcpy_h2d / wf_memcpy_d2h move it by DMA at PCIe rate instead of staging pageable
 * memory through the runtime's bounce buffers.  No reference counterpart (the reference never leaves the host);
 * the shim registers a buffer once, before the copies, and unregisters it before the Vec is dropped. */
int wf_host_register(wf_ctx *ctx, void *h_ptr, size_t bytes);
int wf_host_unregister(wf_ctx *ctx, void *h_ptr);

/* ---- math::fft ------------------------------------------------------------------------------------- */
/* fft::get_twiddles / get_inv_twiddles (math/src/fft/mod.rs:455-505): n/2 elements, bit-reverse permuted. */
int wf_fft_get_twiddles(wf_ctx *ctx, int field, uint32_t log_n, int inverse, void *d_out);

/* fft::evaluate_poly (math/src/fft/mod.rs:85-112): in place, natural-order coefficients -> natural-order
 * evaluations over the domain of size n = 2^log_n.  `batch` independent polynomials laid out back to back
 * (vector v at d_p + v * n * ext_degree elements). */
int wf_fft_evaluate_poly(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_p, uint32_t log_n, uint32_t batch);

/* fft::interpolate_poly (mod.rs:264-295): inverse of the above (includes the 1/n scaling). */
int wf_fft_interpolate_poly(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_evals, uint32_t log_n,
                            uint32_t batch);

/* fft::evaluate_poly_with_offset (mod.rs:168-211): d_result[k] = p(offset * g^k), k < n * 2^log_blowup,
 * g = root of unity of the extended domain.  h_offset points to ONE base-field element (internal form). */
int wf_fft_evaluate_poly_with_offset(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_p, uint32_t log_n,
                                     const void *h_offset, uint32_t log_blowup, void *d_result);

/* fft::interpolate_poly_with_offset (mod.rs:351-386): in place. */
int wf_fft_interpolate_poly_with_offset(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_evals, uint32_t log_n,
                                        const void *h_offset);

/* math::get_power_series_with_offset (math/src/utils/mod.rs:69-79; get_power_series is s = ONE): d_out[i] = s * b^i,
 * i < n, base-field elements.  h_b / h_s: one element each, internal form. */
int wf_get_power_series_with_offset(wf_ctx *ctx, int field, const void *h_b, const void *h_s, uint64_t n, void *d_out);

/* math::batch_inversion (math/src/utils/mod.rs:169-215): d_out[i] = d_values[i]^-1, zero where d_values[i] is zero
 * (serial_batch_inversion's rule).  Base-field elements; d_out may alias d_values. */
int wf_batch_inversion(wf_ctx *ctx, int field, const void *d_values, uint64_t n, void *d_out);

/* ---- prover::matrix -------------------------------------------------------------------------------- */
/* ColMatrix::interpolate_columns (prover/src/matrix/col_matrix.rs:192-202): `num_cols` columns of n elements,
 * column k at d_cols + k * col_stride elements (col_stride >= n*ext_degree, in base elements); in place. */
int wf_interpolate_columns(wf_ctx *ctx, int field, uint32_t ext_degree, void *d_cols, uint32_t num_cols,
                           uint64_t col_stride, uint32_t log_n);

/* Row width (in base elements) of the RowMatrix built by evaluate_polys_over::<8>:
 * 8 * ceil(num_cols*ext_degree / 8)  (prover/src/matrix/row_matrix.rs:112-124, 275-285). */
uint64_t wf_row_width(uint32_t num_cols, uint32_t ext_degree);

/* RowMatrix::evaluate_polys_over::<8> (row_matrix.rs:84-100): coset LDE of all columns into the row-major
 * matrix d_lde[(n << log_blowup)][row_width]; element (row r, base column j) at d_lde[r*row_width + j];
 * padding columns are written as zero (segments.rs:67-75). */
int wf_evaluate_polys_over(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_polys, uint32_t num_cols,
                           uint64_t col_stride, uint32_t log_n, uint32_t log_blowup, const void *h_offset,
                           void *d_lde);

/* Hasher::hash(&[u8]) for the byte hashers (crypto/src/hash/blake/mod.rs:29-31,80-84; sha/mod.rs:26-28): `count` byte strings
 * of len_bytes each, message i at d_msgs + i * stride_bytes (stride a multiple of 8, the bytes between len_bytes and the
 * next multiple of 8 zero).  The Rescue hashers' hash(bytes) is hash_elements over the string's 7-byte chunks (a host-side
 * conversion, rp64_256/mod.rs:123-178), so they return WF_ERR_UNSUPPORTED here. */
int wf_hash_bytes_batch(wf_ctx *ctx, int hash, const void *d_msgs, uint64_t count, uint64_t stride_bytes, uint64_t len_bytes, void *d_out);

/* ColMatrix::evaluate_columns_over (col_matrix.rs:230-243; the column-major LDE the reference's benches/row_matrix.rs
 * compares RowMatrix against): column k of d_out (at k * out_col_stride base elements) = fft::evaluate_poly_with_offset
 * of column k of d_polys; natural order, (n << log_blowup) elements each. */
int wf_evaluate_columns_over(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_polys, uint32_t num_cols,
                             uint64_t col_stride, uint32_t log_n, uint32_t log_blowup, const void *h_offset, void *d_out,
                             uint64_t out_col_stride);

/* ColMatrix::commit_to_rows, row-hash part (col_matrix.rs:262-286): leaf[r] = H::hash_elements([col_0[r], col_1[r], ...])
 * for a COLUMN-major matrix (column k at d_cols + k * col_stride base elements, num_rows elements of ext_degree words).
 * No partitions (the reference's ColMatrix has none).  Feed d_leaves to wf_merkle_build for the commitment. */
int wf_hash_columns(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_cols, uint32_t num_cols,
                    uint64_t col_stride, uint64_t num_rows, void *d_leaves);

/* RowMatrix::commit_to_rows, row-hash part (row_matrix.rs:184-228) with PartitionOptions
 * (air/src/options.rs:428-444): leaf[r] = H::hash_elements(row r) or, when partitioned,
 * H::merge_many(H::hash_elements(chunk_k)).  A row is its first elems_per_row base elements. */
int wf_hash_rows(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_rows, uint64_t num_rows,
                 uint64_t row_width, uint32_t elems_per_row, uint32_t num_partitions, uint32_t hash_rate,
                 void *d_leaves);

/* ---- crypto::merkle -------------------------------------------------------------------------------- */
/* MerkleTree::new / build_merkle_nodes (crypto/src/merkle/mod.rs:116-135, 344-368; concurrent.rs:26-75).
 * d_leaves: num_leaves digests; d_nodes: num_leaves digests in heap order (nodes[0] zeroed). */
int wf_merkle_build(wf_ctx *ctx, int hash, const void *d_leaves, uint64_t num_leaves, void *d_nodes);

/* Hasher::merge over `count` independent pairs (crypto/src/hash/mod.rs:31-50): out[i] = merge(in[2i], in[2i+1]). */
int wf_hash_merge_batch(wf_ctx *ctx, int hash, const void *d_pairs, uint64_t count, void *d_out);

/* Hasher::merge_many over `count` independent groups of k digests (crypto/src/hash/mod.rs:31-50; blake/mod.rs:37-39,
 * rescue/rp64_256/mod.rs:194-196): d_digests[count][k][32 B] -> d_out[count][32 B].  This is the second half of a
 * partitioned row commitment (row_matrix.rs:204-223) and what each GPU runs on the partition digests it received
 * from its peers when columns are sharded across devices. */
int wf_hash_merge_many_batch(wf_ctx *ctx, int hash, const void *d_digests, uint64_t count, uint32_t k, void *d_out);

/* Hasher::merge_with_int for `count` consecutive integers (blake/mod.rs:41-46, rescue/rp64_256/mod.rs:198-219):
 * d_out[i] = merge_with_int(seed, first_value + i).  h_seed: 32 digest bytes in the library's digest layout
 * (Rp64_256: four internal-form words).  This is RandomCoin::next / check_leading_zeros' hash
 * (crypto/src/random/default.rs:92-98,141-146) evaluated for a range of counters / nonces. */
int wf_hash_merge_with_int_batch(wf_ctx *ctx, int hash, const void *h_seed, uint64_t first_value, uint64_t count,
                                 void *d_out);

/* ProverChannel::grind_query_seed (prover/src/channel.rs:169-185): the smallest nonce in [first_nonce, max_nonce]
 * with RandomCoin::check_leading_zeros(nonce) >= grinding_factor (random/default.rs:141-146: trailing zero bits of
 * the first 8 bytes, read little-endian, of merge_with_int(seed, nonce)).  The reference's serial path starts at 1
 * and returns the first hit, i.e. the minimum; so does this (its `concurrent` path may return any hit).
 * WF_ERR_NOT_FOUND when no nonce in the range qualifies. */
int wf_grind(wf_ctx *ctx, int hash, const void *h_seed, uint32_t grinding_factor, uint64_t first_nonce,
             uint64_t max_nonce, uint64_t *h_nonce);

/* ---- crypto::DefaultRandomCoin with its state on the device (crypto/src/random/default.rs:60-250) ---------------
 * The coin's state (seed digest, counter) lives in WF_COIN_BYTES of device memory, so that a chain of
 * commit -> reseed -> draw -> use can be queued without a host round trip per link (wf_fri_build_layers below).  The
 * host-side coin of an integration hands its state over with wf_coin_init and takes it back with wf_coin_read.
 *   wf_coin_init    seed := h_seed (32 digest bytes, library layout), counter := 0
 *   wf_coin_reseed  RandomCoin::reseed (:150-153): seed := merge(seed, d_digest), counter := 0; the digest is also
 *                   copied to d_digest_copy when that is not NULL (the commitment the caller will put in its proof)
 *   wf_coin_draw    `count` x RandomCoin::draw::<E> (:185-199) into d_out (count * ext_degree elements, internal form):
 *                   next() = merge_with_int(seed, ++counter) until the first ELEMENT_BYTES of Digest::as_bytes decode to
 *                   canonical base elements, at most 1000 times per element
 *   wf_coin_reseed_draw  reseed with d_digest, then one draw, in one launch (commit_fri_layer + draw_fri_alpha)
 *   wf_coin_read    waits for the stream; h_seed / h_counter := the state; WF_ERR_NOT_FOUND if a draw used up its 1000
 *                   tries (the reference's RandomCoinError::FailedToDrawFieldElement) */
/* state layout: bytes [0, 32) the seed digest, [32, 40) the counter (little-endian u64), [40, 44) a little-endian u32 of failure
 * bits — bit 0: a draw ran out of its 1000 tries, bit 1: wf_coin_grind found no nonce in its range ("nonce not found") —, the rest
 * reserved — a caller may also write / read the 64 bytes itself instead of wf_coin_init / wf_coin_read */
#define WF_COIN_BYTES 64
int wf_coin_init(wf_ctx *ctx, void *d_coin, const void *h_seed);
int wf_coin_reseed(wf_ctx *ctx, int hash, void *d_coin, const void *d_digest, void *d_digest_copy);
int wf_coin_draw(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, void *d_coin, uint32_t count, void *d_out);
int wf_coin_reseed_draw(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, void *d_coin, const void *d_digest, void *d_digest_copy,
                        void *d_out);
int wf_coin_read(wf_ctx *ctx, const void *d_coin, void *h_seed, uint64_t *h_counter);
/* The query phase's two coin steps against the device coin (round 4: with them Prover::generate_proof, prover/src/lib.rs:282-492, has
 * no host round trip between the first commitment and the query positions):
 *   wf_coin_grind          ProverChannel::grind_query_seed (prover/src/channel.rs:169-185): *d_nonce := the smallest nonce in
 *                          [1, 2^log_max_tries] with check_leading_zeros(nonce) >= grinding_factor for the coin's CURRENT seed (the coin
 *                          is not changed); a fixed queue of batches in increasing order, each returning at once when an earlier one
 *                          has found a nonce, so nothing is read back in between.  No nonce in the range: bit 1 of the coin's
 *                          failure word (wf_coin_read: WF_ERR_NOT_FOUND; the coin is still reseeded by a later wf_coin_draw_integers).  log_max_tries <= grinding_factor + 12 keeps the queue short; the search
 *                          fails with probability exp(-2^(log_max_tries - grinding_factor)).
 *   wf_coin_draw_integers  RandomCoin::draw_integers (crypto/src/random/default.rs:209-248): seed := merge_with_int(seed, *d_nonce),
 *                          counter := 0, d_out[i] := the first 8 bytes of next(), little-endian, masked to 2^log_domain_size, for
 *                          i < num_values <= 256 (duplicates are removed by the caller, prover/src/channel.rs:156-165). */
int wf_coin_grind(wf_ctx *ctx, int hash, const void *d_coin, uint32_t grinding_factor, uint32_t log_max_tries, void *d_nonce /* uint64 */);
int wf_coin_draw_integers(wf_ctx *ctx, int hash, void *d_coin, const void *d_nonce, uint32_t num_values, uint32_t log_domain_size,
                          void *d_out /* num_values x uint64 */);

/* ElementHasher::hash_elements over `count` independent rows (hash/mod.rs:56-64); same layout as wf_hash_rows
 * without partitions. */
int wf_hash_elements_batch(wf_ctx *ctx, int hash, int field, const void *d_elems, uint64_t count,
                           uint64_t row_width, uint32_t elems_per_row, void *d_out);

/* ---- prover::trace::trace_lde ---------------------------------------------------------------------- */
/* build_trace_commitment (prover/src/trace/trace_lde/default/mod.rs:245-282), i.e. DefaultTraceLde::new /
 * set_aux_trace and, with polynomial columns as input, build_constraint_commitment's LDE+commit
 * (prover/src/constraints/commitment/default.rs:136-147):
 *   d_trace  IN  trace columns (evaluations), OUT trace polynomials (TracePolyTable contents)
 *   d_lde    OUT row-major LDE matrix, (n << log_blowup) x wf_row_width() base elements
 *   d_leaves OUT row digests, d_nodes OUT Merkle nodes (heap order), h_root OUT 32-byte commitment
 * If `skip_interpolate` is non-zero the columns are taken to be polynomials already (constraint
 * composition columns). */
int wf_build_trace_commitment(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, void *d_trace, uint32_t num_cols,
                              uint64_t col_stride, uint32_t log_n, uint32_t log_blowup, const void *h_offset,
                              uint32_t num_partitions, uint32_t hash_rate, int skip_interpolate, void *d_lde,
                              void *d_leaves, void *d_nodes, void *h_root);

/* TraceLde::query / read_*_frame_into row access (trace_lde/default/mod.rs:169-215): gather `count` rows
 * (elems_per_row base elements each) of a row-major device matrix into a host buffer. */
int wf_rows_fetch(wf_ctx *ctx, const void *d_rows, uint64_t row_width, uint32_t elems_per_row, uint32_t elem_bytes,
                  const uint64_t *h_positions, uint32_t count, void *h_out);

/* ---- prover::constraints (constraint evaluation for the example AIRs) -------------------------------- */
/* AIR transition functions are user code in the reference (Air::evaluate_transition closures), so only the AIRs the
 * reference ships as examples are built in: */
enum {
    WF_AIR_FIB_SMALL = 0,   /* examples/src/fibonacci/fib_small/air.rs (2 columns, any field)                         */
    WF_AIR_RESCUE = 1,      /* examples/src/rescue/air.rs + rescue.rs (4 columns, 9 periodic columns, f128 only)      */
    WF_AIR_FIB8 = 2,        /* examples/src/fibonacci/fib8/air.rs (2 columns, 8 terms per step, ce_blowup 2)          */
    WF_AIR_MULFIB2 = 3,     /* examples/src/fibonacci/mulfib2/air.rs (2 columns, degree-2 constraints, ce_blowup 2)   */
    WF_AIR_MULFIB8 = 4,     /* examples/src/fibonacci/mulfib8/air.rs (8 columns, degree 2, ce_blowup 2)               */
    WF_AIR_VDF = 5,         /* examples/src/vdf/regular/air.rs (1 column, degree 3, ce_blowup 2)                      */
    WF_AIR_VDF_EXEMPT = 6,  /* examples/src/vdf/exempt/air.rs (the same with 2 transition exemptions)                 */
    WF_AIR_RESCUE_RAPS = 7  /* examples/src/rescue_raps/air.rs (8 + 3 auxiliary columns, 10 periodic columns, f128 only;
                               wf_evaluate_constraints_aux)                                                            */
};

/* DefaultConstraintEvaluator::evaluate for a single-segment trace (prover/src/constraints/evaluator/default.rs:52-106,
 * 165-210) followed by ConstraintEvaluationTable::combine (evaluation_table.rs:163-176,317-407): for every step of the
 * constraint-evaluation domain (2^(log_n + log_ce_blowup) points) read the frame (rows step and step + 1 trace step)
 * from the device-resident row-major trace LDE (RowMatrix layout, row_width words per row), evaluate the AIR's
 * transition constraints, merge them with h_cc_transition, divide by the transition divisor
 * (x^n - 1) / (x - g^(n-1)) (air/src/air/divisor.rs:43-51), add every boundary group
 * sum_a cc_a (state[col_a] - value_a) / (x - g^step_a) (evaluator/boundary.rs:213-232,318-327; divisor.rs:55-71).
 * Assertions here are single-value (Assertion::single; periodic and sequence assertions: wf_evaluate_constraints_assertions below):
 * columns / steps / values (base field, internal form) / coefficients
 * (ext_degree words each) in any order — the result does not depend on it.  log_ce_blowup must be the AIR's
 * ce_blowup_factor (FibSmall 2, Rescue 4; air/src/air/context.rs:104-117), log_lde_blowup >= log_ce_blowup.
 * d_out: 2^(log_n + log_ce_blowup) elements of ext_degree words = CompositionPolyTrace, the input of
 * CompositionPoly::new (wf_fft_interpolate_poly_with_offset). */
int wf_evaluate_constraints(wf_ctx *ctx, int air, int field, uint32_t ext_degree, const void *d_trace_lde, uint64_t row_width,
                            uint32_t log_n, uint32_t log_lde_blowup, uint32_t log_ce_blowup, const void *h_domain_offset,
                            const void *h_cc_transition, uint32_t num_assertions, const uint32_t *h_assert_columns,
                            const uint64_t *h_assert_steps, const void *h_assert_values, const void *h_cc_boundary,
                            void *d_out);

/* The same with assertions of every kind (air/src/air/assertions/mod.rs:57-120) against the main segment: assertion k is
 * (column, first_step, stride, num_values) with its values back to back in h_assert_values —
 *   stride 0, one value:                       Assertion::single   (the entry point above)
 *   stride 2^j >= 2, first_step < stride, one value:            Assertion::periodic (the value at first_step + i stride for every i)
 *   the same with trace_length / stride values:                 Assertion::sequence (values[i] at first_step + i stride).
 * The boundary constraint is f(x) - b(x) over the divisor x^k - g^(first_step k), k = the number of asserted steps
 * (air/src/air/divisor.rs:64-97); b is the value, or for a sequence the polynomial interpolated from the values and evaluated at
 * x g^(-first_step) (air/src/air/boundary/constraint.rs:60-147) — an inverse transform of the values and one coset evaluation over the
 * constraint-evaluation domain per sequence, on the device.  Groups are keyed by (stride, first_step) (boundary/mod.rs:168-186); at most 8
 * groups and 64 assertions.  WF_ERR_INVALID_ARG for a stride / first step / value count the reference's constructors reject. */
int wf_evaluate_constraints_assertions(wf_ctx *ctx, int air, int field, uint32_t ext_degree, const void *d_trace_lde, uint64_t row_width,
                                       uint32_t log_n, uint32_t log_lde_blowup, uint32_t log_ce_blowup, const void *h_domain_offset,
                                       const void *h_cc_transition, uint32_t num_assertions, const uint32_t *h_assert_columns,
                                       const uint64_t *h_assert_first_steps, const uint64_t *h_assert_strides,
                                       const uint64_t *h_assert_num_values, const void *h_assert_values, const void *h_cc_boundary,
                                       void *d_out);

/* wf_evaluate_constraints_assertions with the composition coefficients in DEVICE memory, in the order the coin draws them
 * (ProverChannel::get_constraint_composition_coeffs, prover/src/channel.rs:126-138; air/src/air/mod.rs:529-560): d_cc_transition =
 * num_transition_constraints elements, d_cc_boundary = one element per assertion, assertion k of the h_assert_* arrays taking
 * coefficient k.  h_assert_strides / h_assert_num_values may both be NULL (every assertion single-valued).  The small tables of the
 * call are uploaded through page-locked slots of the context: nothing waits for the stream.  What can still wait: the first call of a
 * shape (table construction, scratch growth, a pool block of a size the context has not cached yet — hipMalloc), and a sequence
 * assertion whose values exceed one 64 KiB stage slot (bounce-buffer copy). */
int wf_evaluate_constraints_dev(wf_ctx *ctx, int air, int field, uint32_t ext_degree, const void *d_trace_lde, uint64_t row_width,
                                uint32_t log_n, uint32_t log_lde_blowup, uint32_t log_ce_blowup, const void *h_domain_offset,
                                const void *d_cc_transition, uint32_t num_assertions, const uint32_t *h_assert_columns,
                                const uint64_t *h_assert_first_steps, const uint64_t *h_assert_strides,
                                const uint64_t *h_assert_num_values, const void *h_assert_values, const void *d_cc_boundary,
                                void *d_out);

/* The same for a trace with an auxiliary segment (TraceInfo::is_multi_segment): evaluate_fragment_full
 * (evaluator/default.rs:214-271) = the main transition constraints as above plus Air::evaluate_aux_transition over the
 * main frame, the auxiliary frame (rows of the device-resident row-major LDE of the aux segment: aux_row_width base
 * elements per row, the segment's columns first, ext_degree words each) and the segment's random elements
 * (AuxRandElements: h_aux_rand_elements, elements of E), merged under the one transition divisor with the coefficients
 * h_cc_transition = [main constraints..., aux constraints...] (coefficient times evaluation over E, default.rs:325-331);
 * BoundaryConstraints::evaluate_all (boundary.rs:103-115): an assertion against the aux segment (column, step, value in
 * E, coefficient) joins the boundary group of its step (boundary.rs:58-73).  Built in: WF_AIR_RESCUE_RAPS. */
int wf_evaluate_constraints_aux(wf_ctx *ctx, int air, int field, uint32_t ext_degree, const void *d_main_lde, uint64_t main_row_width,
                                const void *d_aux_lde, uint64_t aux_row_width, uint32_t log_n, uint32_t log_lde_blowup,
                                uint32_t log_ce_blowup, const void *h_domain_offset, const void *h_cc_transition,
                                uint32_t num_assertions, const uint32_t *h_assert_columns, const uint64_t *h_assert_steps,
                                const void *h_assert_values, const void *h_cc_boundary, uint32_t num_aux_assertions,
                                const uint32_t *h_aux_assert_columns, const uint64_t *h_aux_assert_steps,
                                const void *h_aux_assert_values, const void *h_cc_aux_boundary, const void *h_aux_rand_elements,
                                void *d_out);

/* ---- prover::composer (DEEP composition) and out-of-domain frames ------------------------------------- */
/* ColMatrix::evaluate_columns_at for one or more points (prover/src/matrix/col_matrix.rs; polynom::eval,
 * math/src/polynom/mod.rs:55-61).  TracePolyTable::get_ood_frame (prover/src/trace/poly_table.rs:68-76) and
 * CompositionPoly::get_ood_frame (prover/src/constraints/composition_poly.rs:101-108) are this call with the points
 * {z, z*g}.  d_polys: num_cols columns of 2^log_n coefficients, col_stride words apart; a coefficient is
 * poly_ext_degree (1 = base-field column, or ext_degree) words.  h_points: num_points elements of ext_degree words.
 * h_out[num_points][num_cols] elements of ext_degree words (host memory; the call synchronises the stream). */
int wf_polys_evaluate_at(wf_ctx *ctx, int field, uint32_t poly_ext_degree, uint32_t ext_degree, const void *d_polys,
                         uint32_t num_cols, uint64_t col_stride, uint32_t log_n, const void *h_points,
                         uint32_t num_points, void *h_out);

/* get_ood_frame with the point in DEVICE memory (d_point: one element of ext_degree words, where wf_coin_draw put it) and the values
 * left on the device: d_out[p][col] for p = 0 (at the point) and, with_next != 0, p = 1 (at point * g, g = the generator of the
 * 2^log_n domain).  Nothing waits for the stream. */
int wf_polys_evaluate_at_dev(wf_ctx *ctx, int field, uint32_t poly_ext_degree, uint32_t ext_degree, const void *d_polys,
                             uint32_t num_cols, uint64_t col_stride, uint32_t log_n, const void *d_point, int with_next, void *d_out);

/* DeepCompositionPoly::add_trace_polys (prover/src/composer/mod.rs:67-169): the coefficients of
 *   sum_i cc_i * [ (T_i(x) - T_i(z)) / (x - z) + (T_i(x) - T_i(z*g)) / (x - z*g) ]
 * over the main-segment polys (base field), the aux-segment polys and the composition-poly columns (both over the
 * extension), g = generator of the trace domain.  The reference subtracts cc_i * T_i(z) from coefficient 0 and then
 * runs syn_div_in_place (math/src/polynom/mod.rs:499-506), which drops the remainder: the out-of-domain values only
 * affect that dropped remainder, so they are not inputs here (composer/mod.rs:200-210).
 * h_cc_trace: num_main + num_aux coefficients, h_cc_constraints: num_quotient coefficients, h_z: one element; all of
 * ext_degree words in internal form.  d_out: 2^log_n elements (degree 2^log_n - 2: the last one is zero).
 * Feed d_out to wf_fft_evaluate_poly_with_offset for DeepCompositionPoly::evaluate (composer/mod.rs:174-181). */
int wf_deep_compose(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_main_polys, uint32_t num_main,
                    uint64_t main_stride, const void *d_aux_polys, uint32_t num_aux, uint64_t aux_stride,
                    const void *d_quotient_polys, uint32_t num_quotient, uint64_t quotient_stride, uint32_t log_n,
                    const void *h_z, const void *h_cc_trace, const void *h_cc_constraints, void *d_out);

/* wf_deep_compose with z and the coefficients in DEVICE memory: d_cc = num_main + num_aux + num_quotient elements, trace columns first
 * (DeepCompositionCoefficients {trace, constraints} in draw order, air/src/air/coefficients.rs:201-206). */
int wf_deep_compose_dev(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_main_polys, uint32_t num_main,
                        uint64_t main_stride, const void *d_aux_polys, uint32_t num_aux, uint64_t aux_stride,
                        const void *d_quotient_polys, uint32_t num_quotient, uint64_t quotient_stride, uint32_t log_n,
                        const void *d_z, const void *d_cc, void *d_out);

/* ---- fri::FriProver (commit phase) ------------------------------------------------------------------- */
/* FriProver::build_layer, first half (fri/src/prover/mod.rs:202-211): transpose_slice::<E, N> (utils/core/src/lib.rs:
 * 166-183) into d_transposed (kept: it is the layer's `evaluations` used by query_layer, mod.rs:297-319), then
 * build_layer_commitment (mod.rs:321-336): leaf_i = H::hash_elements(row i), MerkleTree over the len/N leaves.
 * folding N in {2, 4, 8, 16}.  h_root receives the layer commitment that the caller writes into its channel. */
int wf_fri_layer_commit(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_evals, uint32_t log_len,
                        uint32_t folding, void *d_transposed, void *d_leaves, void *d_nodes, void *h_root);

/* FriProver::build_layer, second half = folding::apply_drp (fri/src/folding/mod.rs:86-118) once the caller has drawn
 * alpha from its channel: d_folded[i] = DRP of row i.  h_domain_offset: one base element, h_alpha: one E element
 * (ext_degree words), both in internal form. */
int wf_fri_apply_drp(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_transposed, uint32_t log_len,
                     uint32_t folding, const void *h_domain_offset, const void *h_alpha, void *d_folded);

/* wf_fri_apply_drp_rows with alpha in device memory (where wf_coin_draw / wf_coin_reseed_draw put it). */
int wf_fri_apply_drp_rows_dev(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_transposed_rows, uint32_t log_len,
                              uint32_t folding, uint64_t row_start, uint64_t num_rows, const void *h_domain_offset,
                              const void *d_alpha, void *d_folded);

/* FriProver::build_layers (fri/src/prover/mod.rs:179-239) against a device-resident coin: for each of the num_layers layers
 * build_layer = commit (as wf_fri_layer_commit), channel.commit_fri_layer(root) = coin.reseed(root),
 * alpha = channel.draw_fri_alpha() = coin.draw::<E>(), apply_drp (as wf_fri_apply_drp) — queued back to back on the
 * context's stream, nothing waits on the host (a layer's fold and the next layer's transpose + leaf hashes are one launch
 * where a fused kernel exists).  Layer k (k = 0 .. num_layers-1) works on 2^log_len / folding^k points:
 *   d_transposed[k], d_leaves[k], d_nodes[k]   OUT  the layer's evaluations / row digests / Merkle nodes (kept for the query phase)
 *   d_folded[k]                                OUT  the next layer's evaluations
 *   d_roots   OUT (num_layers + 1) x 32 bytes, d_alphas OUT num_layers x ext_degree elements (what the channel would have seen)
 * The four pointer arrays are host arrays of device pointers.  With d_remainder != NULL the remainder step (set_remainder,
 * mod.rs:230-239) follows on the stream: the last evaluations (d_folded[num_layers-1], whose contents are unspecified afterwards,
 * or a private copy of d_evals when there is no layer — d_evals itself is never written) are interpolated over the coset, the
 * first len / blowup coefficients go to d_remainder in reverse order, their hash_elements digest to d_roots[num_layers] and into
 * the coin.  d_remainder == NULL: the caller does that step (blowup unused).  From the first layer of at most 1024 rows on, the
 * layers and the remainder are ONE launch where a fused kernel exists (f64, BLAKE3 family).
 * The caller reads roots / alphas / remainder / the coin back once, after the call. */
int wf_fri_build_layers(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_evals, uint32_t log_len,
                        uint32_t folding, uint32_t num_layers, const void *h_domain_offset, void *d_coin,
                        void *const *d_transposed, void *const *d_leaves, void *const *d_nodes, void *const *d_folded,
                        void *d_roots, void *d_alphas, uint32_t blowup, void *d_remainder);

/* The same fold for `num_rows` consecutive rows, starting at row_start, of a layer whose full domain has 2^log_len
 * points: d_transposed_rows holds only those rows, d_folded receives num_rows elements.  This is what one GPU runs on
 * its contiguous row range when a FRI layer is sharded across devices (row i uses x_i = offset * g^i with the GLOBAL
 * i, fri/src/folding/mod.rs:181-188); row_start = 0, num_rows = 2^log_len / folding is wf_fri_apply_drp. */
int wf_fri_apply_drp_rows(wf_ctx *ctx, int field, uint32_t ext_degree, const void *d_transposed_rows, uint32_t log_len,
                          uint32_t folding, uint64_t row_start, uint64_t num_rows, const void *h_domain_offset,
                          const void *h_alpha, void *d_folded);

/* ---- multi-device: one rank per GPU (SURVEY 8b `device_ids[]`, 8e) ------------------------------------------- */
/* The reference's multi-device hook is PartitionOptions (air/src/options.rs:391-451): with num_partitions = G every device
 * owns one partition's columns, hashes its part of every row (row_matrix.rs:204-223), and a row's leaf is merge_many over
 * the G partition digests.  A wf_comm ties one context (= one GPU) to its peers; the two exchange steps of the commitment
 * run on the device interconnect (RCCL over xGMI: all-to-all of the 32-byte partition digests, all-gather of the G
 * sub-roots).  One process per GPU is the intended deployment: rank 0 calls wf_comm_get_unique_id and hands the 128 bytes
 * to the other ranks (the Rust host already has a channel for that), every rank calls wf_comm_init_rank.  Ranks that are
 * threads of one process can use wf_comm_init_loopback instead (peer copies between the ranks' buffers, no RCCL): the
 * single-process multi-GPU case and the way G logical ranks are tested on one GPU.  Collectives must be entered by every
 * rank, in the same order. */
#define WF_COMM_ID_BYTES 128
typedef struct wf_comm wf_comm;
int wf_comm_get_unique_id(uint8_t *id /* WF_COMM_ID_BYTES */);
int wf_comm_init_rank(wf_ctx *ctx, const uint8_t *id, int rank, int world, wf_comm **out);
int wf_comm_init_loopback(wf_ctx *const *ctxs, int world, wf_comm **out /* world handles, one per context */);
int wf_comm_destroy(wf_comm *comm);
int wf_comm_rank(const wf_comm *comm);
int wf_comm_size(const wf_comm *comm);
/* every rank contributes `bytes` bytes; d_recv receives world * bytes in rank order (sub-roots, FRI tails, remainders) */
int wf_comm_all_gather(wf_comm *comm, const void *d_send, void *d_recv, uint64_t bytes);
/* block k of d_send (`bytes` bytes each) goes to rank k, block k of d_recv comes from rank k; distinct buffers (partition
 * digests, the FRI re-stride of fri/src/prover/mod.rs:202-211 over row-range shards) */
int wf_comm_all_to_all(wf_comm *comm, const void *d_send, void *d_recv, uint64_t bytes);
/* DefaultTraceLde::new / build_trace_commitment (prover/src/trace/trace_lde/default/mod.rs:63-86,245-282) with the columns
 * sharded by partition: this rank holds partition `rank` (shard_cols columns of 2^log_n evaluations, col_stride base-field
 * ELEMENTS apart, as in wf_build_trace_commitment).  It interpolates and extends its columns, hashes its part of every LDE row, receives the other partitions'
 * digests of ITS row range [rank N/G, (rank+1) N/G), N = 2^(log_n + log_blowup), merges them into leaves, builds the
 * subtree over them, and after the sub-root all-gather the top log2 G levels.  On return (all device memory of this rank):
 * d_trace_shard = polynomials, d_lde_shard = the shard's RowMatrix [N][wf_row_width(shard_cols, ext_degree)],
 * d_leaves / d_nodes = N/G digests each (nodes in heap order), d_top = G digests, heap order, d_top[1] = the root (G = 1:
 * d_top[0] = the root).  Node for node this is the single-device commitment under PartitionOptions::new(G, hash_rate) when
 * shard_cols is that option's partition size; G must be a power of two. */
int wf_comm_sharded_commit(wf_comm *comm, int hash, int field, uint32_t ext_degree, void *d_trace_shard, uint32_t shard_cols,
                           uint64_t col_stride, uint32_t log_n, uint32_t log_blowup, const void *h_offset, int skip_interpolate,
                           void *d_lde_shard, void *d_leaves, void *d_nodes, void *d_top, void *h_root);

/* FriProver::build_layers with the layers sharded by contiguous row ranges, this rank's part (call it on every rank; DESIGN.md
 * section 6): rank r holds piece r — the contiguous 2^log_len / G evaluations [r len/G, (r+1) len/G) — of the first layer.
 * For each of the num_layers layers (the caller picks how many stay sharded: rows per rank >= 2 and a multiple of G):
 *   1. re-stride: the rank needs rows [r rc/G, (r+1) rc/G) of the transposed layer, i.e. chunk (j, r) of every strided run j —
 *      one all-to-all of equal blocks when G divides the folding factor, one all-gather of the layer otherwise;
 *   2. the rank's rows, their leaves and the subtree over them (as wf_fri_layer_commit);  3. all-gather of the G sub-roots,
 *      top tree on every rank;  4. coin.reseed(root), alpha = coin.draw() on every rank's identical device coin;
 *   5. fold the local rows with their GLOBAL row index (as wf_fri_apply_drp_rows): the rank's piece of the next layer.
 * Outputs per layer k (host arrays of device pointers): d_rows[k] rows_local x folding elements, d_leaves[k] / d_nodes[k]
 * rows_local x 32 bytes (subtree, heap order), d_top[k] G x 32 bytes (heap order, [1] = the layer root; G = 1: the root at
 * [0]), d_folded[k] rows_local elements; d_roots num_layers x 32, d_alphas num_layers elements — identical on every rank.
 * Node for node FriProver::build_layers on one device (tests/test_gpu_comm.py).  The caller gathers the last pieces
 * (wf_comm_all_gather) and finishes the small layers and the remainder unsharded (wf_fri_build_layers on every rank). */
int wf_comm_sharded_fri_layers(wf_comm *comm, int hash, int field, uint32_t ext_degree, const void *d_piece, uint32_t log_len,
                               uint32_t folding, uint32_t num_layers, const void *h_domain_offset, void *d_coin,
                               void *const *d_rows, void *const *d_leaves, void *const *d_nodes, void *const *d_top,
                               void *const *d_folded, void *d_roots, void *d_alphas);

#ifdef __cplusplus
}
#endif
#endif /* WINTERFELL_HIP_H */
