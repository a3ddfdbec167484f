// Row hashing (ElementHasher::hash_elements / merge_many over matrix rows) and Merkle tree construction.
//
// Reference behaviour reproduced here:
//   RowMatrix::commit_to_rows          prover/src/matrix/row_matrix.rs:184-228 (+ PartitionOptions, air/src/options.rs:428-444)
//   Blake3_256::{hash_elements,merge}  crypto/src/hash/blake/mod.rs:33-65  (f64: canonical LE bytes of as_int())
//   Rp64_256::{hash_elements,merge}    crypto/src/hash/rescue/rp64_256/mod.rs:181-257
//   build_merkle_nodes                 crypto/src/merkle/mod.rs:344-368, concurrent.rs:26-75
//
// Kernels: one row (or one row partition) per lane for leaf hashing; one workgroup per 1024-input subtree for the
// tree (10 levels per launch, intermediate levels staged in LDS, every node written to the reference's heap layout).
#include <string.h>

#include "blake3.cuh"
#include "keccak.cuh"
#include "fields.cuh"
#include "rp62.cuh"
#include "rp64.cuh"
#include "rpjive64.cuh"
#include "rescue_coop.cuh"
#include "wf_internal.h"

namespace {

// MODE_DIGESTS: the words are 32-byte digest slots (merge_many); identical to MODE_RAW except for 24-byte digests
enum { MODE_F64_CANON = 0, MODE_RAW = 1, MODE_F62_CANON = 2, MODE_DIGESTS = 3 };

struct Digest {
    uint32_t w[8];
};

#ifndef WF_B3_STAGE_LEVELS
#define WF_B3_STAGE_LEVELS 10
#endif

// ---- per-hasher primitives on 32-byte digests ----------------------------------------------------------------
struct HBlake3 {
    static constexpr bool WIDE = true;               // rows of >= 64 bytes: wave-cooperative block loads (hash_rows_wide_kernel)
    static constexpr int WIDE_BW = 8;                // 64-bit words per message block
    static constexpr bool BYTES = true;              // Hasher::hash(&[u8]) supported on the device
    // Hasher::hash (blake/mod.rs:29-31): p = the message as zero-padded 64-bit words
    static __device__ __forceinline__ void hash_bytes(const uint64_t *p, uint64_t nbytes, uint32_t (&out)[8]) {
        const uint32_t nwords = (uint32_t)((nbytes + 3) / 4);
        auto fetch = [&](uint32_t blk, uint32_t nvalid, uint32_t (&m)[16]) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint64_t v = (uint32_t)(2 * i) < nvalid ? p[blk * 8 + i] : 0;
                m[2 * i] = (uint32_t)v;
                m[2 * i + 1] = (uint32_t)(2 * i + 1) < nvalid ? (uint32_t)(v >> 32) : 0u;
            }
        };
        b3::hash_blocks(fetch, nwords, out, (uint32_t)(nwords * 4 - nbytes));
    }
    template <class FB>
    static __device__ __forceinline__ void hash_wide(const FB &fetch64, uint32_t nelem, uint32_t (&out)[8]) {
        auto fetch = [&](uint32_t blk, uint32_t, uint32_t (&m)[16]) {
            uint64_t v[8];
            fetch64(blk, v);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                m[2 * i] = (uint32_t)v[i];
                m[2 * i + 1] = (uint32_t)(v[i] >> 32);
            }
        };
        b3::hash_blocks(fetch, nelem * 2, out);
    }
    static constexpr bool COOP = false;
    // levels reduced per Merkle launch: BLAKE3 merges are cheap, so a workgroup walks 10 levels through LDS
    static constexpr uint32_t STAGE_LEVELS = WF_B3_STAGE_LEVELS;
    static constexpr bool WAVE_TREE = true;    // barrier-free wavefront Merkle stage (merkle_wave_kernel)
    static const char *row_name() { return "hash_rows_blake3"; }
    static const char *merkle_name() { return "merkle_stage_blake3"; }
    static const char *grind_name() { return "grind_blake3"; }
    static __device__ __forceinline__ void merge(const uint32_t (&in)[16], uint32_t (&out)[8]) { b3::merge(in, out); }
    // merge_with_int (blake/mod.rs:41-46): hash of the 40 bytes seed || value.to_le_bytes() — one block
    static __device__ __forceinline__ void merge_with_int(const uint32_t (&seed)[8], uint64_t value, uint32_t (&out)[8]) {
        uint32_t cv[8], m[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            cv[i] = b3::iv(i);
            m[i] = seed[i];
            m[8 + i] = 0;
        }
        m[8] = (uint32_t)value;
        m[9] = (uint32_t)(value >> 32);
        b3::compress(cv, m, 0, 40, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT, out);
    }
    // first 8 digest bytes as a little-endian integer (random/default.rs:141-146)
    static __device__ __forceinline__ uint64_t head(const uint32_t (&d)[8]) { return (uint64_t)d[0] | ((uint64_t)d[1] << 32); }
    // Digest::as_bytes as eight little-endian words (zero beyond the digest's own length, crypto/src/hash/mod.rs ByteDigest)
    static __device__ __forceinline__ void as_bytes(const uint32_t (&d)[8], uint32_t (&b)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) b[i] = d[i];
    }
    // hash `nelem` 64-bit words starting at p; MODE selects how a word is turned into message bytes
    template <int MODE, bool MULTI>
    static __device__ __forceinline__ void hash_elems(const uint64_t *p, uint32_t nelem, uint32_t (&out)[8]) {
        auto w = [&](uint32_t i) -> uint32_t {
            uint64_t v = p[i >> 1];
            if (MODE == MODE_F64_CANON) v = gl::to_int(v);                       // as_int(): canonical integer
            else if (MODE == MODE_F62_CANON) v = f62::mul(f62::norm(v), 1);      // as_int(): Montgomery multiply by 1
            return (i & 1) ? (uint32_t)(v >> 32) : (uint32_t)v;
        };
        if (MULTI) b3::hash_words(w, nelem * 2, out);
        else b3::chunk(w, 0, nelem * 2, 0, true, out);                           // <= 1024 bytes: a single chunk, no CV stack
    }
};

// Blake3_192<B> (crypto/src/hash/blake/mod.rs:68-125): BLAKE3 truncated to 24 bytes.  Digests live in the library's
// 32-byte slots with bytes 24..31 zero; what is hashed is the reference's byte string (48 bytes for a merge, 24 k bytes
// for merge_many, seed[..24] || value for merge_with_int).
struct HBlake3_192 {
    static constexpr bool WIDE = true;
    static constexpr int WIDE_BW = 8;
    static constexpr bool BYTES = true;
    static __device__ __forceinline__ void hash_bytes(const uint64_t *p, uint64_t nbytes, uint32_t (&out)[8]) {
        HBlake3::hash_bytes(p, nbytes, out);
        out[6] = out[7] = 0;
    }
    template <class FB>
    static __device__ __forceinline__ void hash_wide(const FB &fetch64, uint32_t nelem, uint32_t (&out)[8]) {
        HBlake3::hash_wide(fetch64, nelem, out);
        out[6] = out[7] = 0;
    }
    static constexpr bool COOP = false;
    static constexpr uint32_t STAGE_LEVELS = WF_B3_STAGE_LEVELS;
    static constexpr bool WAVE_TREE = true;    // barrier-free wavefront Merkle stage (merkle_wave_kernel)
    static const char *row_name() { return "hash_rows_blake3_192"; }
    static const char *merkle_name() { return "merkle_stage_blake3_192"; }
    static const char *grind_name() { return "grind_blake3_192"; }
    static __device__ __forceinline__ void trunc(uint32_t (&out)[8]) { out[6] = out[7] = 0; }
    static __device__ __forceinline__ void merge(const uint32_t (&in)[16], uint32_t (&out)[8]) {
        uint32_t cv[8], m[16];
#pragma unroll
        for (int i = 0; i < 8; i++) cv[i] = b3::iv(i);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            m[i] = in[i];
            m[6 + i] = in[8 + i];
        }
        m[12] = m[13] = m[14] = m[15] = 0;
        b3::compress(cv, m, 0, 48, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT, out);
        trunc(out);
    }
    static __device__ __forceinline__ void merge_with_int(const uint32_t (&seed)[8], uint64_t value, uint32_t (&out)[8]) {
        uint32_t cv[8], m[16];
#pragma unroll
        for (int i = 0; i < 8; i++) cv[i] = b3::iv(i);
#pragma unroll
        for (int i = 0; i < 16; i++) m[i] = i < 6 ? seed[i] : 0;
        m[6] = (uint32_t)value;
        m[7] = (uint32_t)(value >> 32);
        b3::compress(cv, m, 0, 32, b3::CHUNK_START | b3::CHUNK_END | b3::ROOT, out);
        trunc(out);
    }
    static __device__ __forceinline__ uint64_t head(const uint32_t (&d)[8]) { return (uint64_t)d[0] | ((uint64_t)d[1] << 32); }
    // Digest::as_bytes as eight little-endian words (zero beyond the digest's own length, crypto/src/hash/mod.rs ByteDigest)
    static __device__ __forceinline__ void as_bytes(const uint32_t (&d)[8], uint32_t (&b)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) b[i] = d[i];
    }
    template <int MODE, bool MULTI>
    static __device__ __forceinline__ void hash_elems(const uint64_t *p, uint32_t nelem, uint32_t (&out)[8]) {
        if (MODE == MODE_DIGESTS) {
            // nelem / 4 digest slots, 6 message words each
            const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
            auto w = [&](uint32_t i) -> uint32_t { return q[(i / 6) * 8 + (i % 6)]; };
            b3::hash_words(w, (nelem / 4) * 6, out);
        } else {
            HBlake3::hash_elems<MODE, MULTI>(p, nelem, out);
        }
        trunc(out);
    }
};

// RpJive64_256 (crypto/src/hash/rescue/rp64_256_jive/mod.rs): ElementDigest like Rp64_256, width-8 permutation
struct HRpJive {
    static constexpr bool WIDE = false;
    static constexpr bool BYTES = false;             // hash(bytes) = hash_elements over the 7-byte chunks (host conversion)
    static constexpr bool COOP = true;              // small batches: one state word per lane (rescue_coop.cuh)
    typedef rcoop::CoopRpJive Coop;
    static constexpr uint32_t STAGE_LEVELS = 1;      // as for Rp64_256: one full-width level per launch
    static constexpr bool WAVE_TREE = false;    // barrier-free wavefront Merkle stage (merkle_wave_kernel)
    static const char *row_name() { return "hash_rows_rpjive"; }
    static const char *merkle_name() { return "merkle_stage_rpjive"; }
    static const char *grind_name() { return "grind_rpjive"; }
    static __device__ __forceinline__ void put(const uint64_t (&d)[4], uint32_t (&out)[8]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            out[2 * i] = (uint32_t)d[i];
            out[2 * i + 1] = (uint32_t)(d[i] >> 32);
        }
    }
    static __device__ __forceinline__ void merge(const uint32_t (&in)[16], uint32_t (&out)[8]) {
        uint64_t two[8], d[4];
#pragma unroll
        for (int i = 0; i < 8; i++) two[i] = (uint64_t)in[2 * i] | ((uint64_t)in[2 * i + 1] << 32);
        rpj::merge(two, d);
        put(d, out);
    }
    // merge_with_int (mod.rs:223-263): seed in state[0..4], value in state[4] (and [5] when it exceeds the modulus),
    // element count in state[7]; Jive summation with the initial state
    static __device__ __forceinline__ void merge_with_int(const uint32_t (&seed)[8], uint64_t value, uint32_t (&out)[8]) {
        uint64_t st[8], init[8], d[4];
#pragma unroll
        for (int i = 0; i < 8; i++) st[i] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) st[i] = (uint64_t)seed[2 * i] | ((uint64_t)seed[2 * i + 1] << 32);
        constexpr uint64_t R2 = 0xfffffffe00000001ull;
        st[4] = gl::mul(value >= gl::P ? value - gl::P : value, R2);
        if (value < gl::P) st[7] = rp64::mont_small(5);
        else {
            st[5] = rp64::mont_small(1);
            st[7] = rp64::mont_small(6);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) init[i] = st[i];
        rpj::permute(st);
        rpj::jive_sum(init, st, d);
        put(d, out);
    }
    static __device__ __forceinline__ uint64_t head(const uint32_t (&d)[8]) {
        return gl::to_int((uint64_t)d[0] | ((uint64_t)d[1] << 32));
    }
    // ElementDigest::as_bytes: the canonical little-endian bytes of the four elements
    static __device__ __forceinline__ void as_bytes(const uint32_t (&d)[8], uint32_t (&b)[8]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t v = gl::to_int((uint64_t)d[2 * i] | ((uint64_t)d[2 * i + 1] << 32));
            b[2 * i] = (uint32_t)v;
            b[2 * i + 1] = (uint32_t)(v >> 32);
        }
    }
    template <int MODE, bool MULTI>
    static __device__ __forceinline__ void hash_elems(const uint64_t *p, uint32_t nelem, uint32_t (&out)[8]) {
        uint64_t d[4];
        auto e = [&](uint32_t i) -> uint64_t { return p[i]; };
        rpj::hash_elements(e, nelem, d);
        put(d, out);
    }
};

// Rp62_248 (crypto/src/hash/rescue/rp62_248/mod.rs): four f62 words per digest, defined over f62 only
struct HRp62 {
    static constexpr bool WIDE = false;
    static constexpr bool BYTES = false;             // hash(bytes) = hash_elements over the 7-byte chunks (host conversion)
    static constexpr bool COOP = true;              // small batches: one state word per lane (rescue_coop.cuh)
    typedef rcoop::CoopRp62 Coop;
    static constexpr uint32_t STAGE_LEVELS = 1;
    static constexpr bool WAVE_TREE = false;    // barrier-free wavefront Merkle stage (merkle_wave_kernel)
    static const char *row_name() { return "hash_rows_rp62"; }
    static const char *merkle_name() { return "merkle_stage_rp62"; }
    static const char *grind_name() { return "grind_rp62"; }
    static __device__ __forceinline__ void put(const uint64_t (&d)[4], uint32_t (&out)[8]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            out[2 * i] = (uint32_t)d[i];
            out[2 * i + 1] = (uint32_t)(d[i] >> 32);
        }
    }
    static __device__ __forceinline__ void merge(const uint32_t (&in)[16], uint32_t (&out)[8]) {
        uint64_t two[8], d[4];
#pragma unroll
        for (int i = 0; i < 8; i++) two[i] = (uint64_t)in[2 * i] | ((uint64_t)in[2 * i + 1] << 32);
        rp62::merge(two, d);
        put(d, out);
    }
    // merge_with_int (mod.rs:172-201)
    static __device__ __forceinline__ void merge_with_int(const uint32_t (&seed)[8], uint64_t value, uint32_t (&out)[8]) {
        uint64_t st[12], d[4];
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) st[i] = f62::norm((uint64_t)seed[2 * i] | ((uint64_t)seed[2 * i + 1] << 32));
        st[4] = rp62::to_mont(value % f62::M);
        if (value < f62::M) st[11] = rp62::to_mont(5);
        else {
            st[5] = rp62::to_mont(value / f62::M);
            st[11] = rp62::to_mont(6);
        }
        rp62::permute(st);
#pragma unroll
        for (int i = 0; i < 4; i++) d[i] = st[i];
        put(d, out);
    }
    // ElementDigest::as_bytes packs 4 x 62 bits (digest.rs:37-51): the first 8 bytes are v1 | (v2 << 62)
    static __device__ __forceinline__ uint64_t head(const uint32_t (&d)[8]) {
        const uint64_t v1 = f62::mul(f62::norm((uint64_t)d[0] | ((uint64_t)d[1] << 32)), 1);
        const uint64_t v2 = f62::mul(f62::norm((uint64_t)d[2] | ((uint64_t)d[3] << 32)), 1);
        return v1 | (v2 << 62);
    }
    static __device__ __forceinline__ void as_bytes(const uint32_t (&d)[8], uint32_t (&b)[8]) {   // digest.rs:37-51, all four words
        uint64_t v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = f62::mul(f62::norm((uint64_t)d[2 * i] | ((uint64_t)d[2 * i + 1] << 32)), 1);
        const uint64_t w[4] = {v[0] | (v[1] << 62), (v[1] >> 2) | (v[2] << 60), (v[2] >> 4) | (v[3] << 58), v[3] >> 6};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            b[2 * i] = (uint32_t)w[i];
            b[2 * i + 1] = (uint32_t)(w[i] >> 32);
        }
    }
    template <int MODE, bool MULTI>
    static __device__ __forceinline__ void hash_elems(const uint64_t *p, uint32_t nelem, uint32_t (&out)[8]) {
        uint64_t d[4];
        auto e = [&](uint32_t i) -> uint64_t { return p[i]; };
        rp62::hash_elements(e, nelem, d);
        put(d, out);
    }
};

// Sha3_256<B> (crypto/src/hash/sha/mod.rs:21-66): same byte-level structure as Blake3_256 with SHA3-256 as the byte hash
struct HSha3 {
    static constexpr bool WIDE = true;
    static constexpr int WIDE_BW = 17;               // the 136-byte rate
    static constexpr bool BYTES = true;
    // Hasher::hash (sha/mod.rs:26-28)
    static __device__ __forceinline__ void hash_bytes(const uint64_t *p, uint64_t nbytes, uint32_t (&out)[8]) {
        const uint32_t nwords = (uint32_t)((nbytes + 7) / 8);
        auto fetch = [&](uint32_t blk, uint64_t (&m)[17]) {
#pragma unroll
            for (int i = 0; i < 17; i++) m[i] = blk * 17 + i < nwords ? p[blk * 17 + i] : 0;
        };
        uint64_t d[4];
        k3::sha3_256_bytes(fetch, nbytes, d);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            out[2 * i] = (uint32_t)d[i];
            out[2 * i + 1] = (uint32_t)(d[i] >> 32);
        }
    }
    template <class FB>
    static __device__ __forceinline__ void hash_wide(const FB &fetch64, uint32_t nelem, uint32_t (&out)[8]) {
        uint64_t d[4];
        k3::sha3_256_blocks(fetch64, nelem, d);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            out[2 * i] = (uint32_t)d[i];
            out[2 * i + 1] = (uint32_t)(d[i] >> 32);
        }
    }
    static constexpr bool COOP = false;
    static constexpr uint32_t STAGE_LEVELS = 8;
    static constexpr bool WAVE_TREE = false;    // barrier-free wavefront Merkle stage (merkle_wave_kernel)
    static const char *row_name() { return "hash_rows_sha3"; }
    static const char *merkle_name() { return "merkle_stage_sha3"; }
    static const char *grind_name() { return "grind_sha3"; }
    static __device__ __forceinline__ void put(const uint64_t (&d)[4], uint32_t (&out)[8]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            out[2 * i] = (uint32_t)d[i];
            out[2 * i + 1] = (uint32_t)(d[i] >> 32);
        }
    }
    static __device__ __forceinline__ void merge(const uint32_t (&in)[16], uint32_t (&out)[8]) {
        uint64_t d[4];
        auto w = [&](uint32_t i) -> uint64_t { return (uint64_t)in[2 * i] | ((uint64_t)in[2 * i + 1] << 32); };
        k3::sha3_256_words(w, 8, d);
        put(d, out);
    }
    static __device__ __forceinline__ void merge_with_int(const uint32_t (&seed)[8], uint64_t value, uint32_t (&out)[8]) {
        uint64_t d[4];
        auto w = [&](uint32_t i) -> uint64_t { return i < 4 ? ((uint64_t)seed[2 * i] | ((uint64_t)seed[2 * i + 1] << 32)) : value; };
        k3::sha3_256_words(w, 5, d);
        put(d, out);
    }
    static __device__ __forceinline__ uint64_t head(const uint32_t (&d)[8]) { return (uint64_t)d[0] | ((uint64_t)d[1] << 32); }
    // Digest::as_bytes as eight little-endian words (zero beyond the digest's own length, crypto/src/hash/mod.rs ByteDigest)
    static __device__ __forceinline__ void as_bytes(const uint32_t (&d)[8], uint32_t (&b)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) b[i] = d[i];
    }
    template <int MODE, bool MULTI>
    static __device__ __forceinline__ void hash_elems(const uint64_t *p, uint32_t nelem, uint32_t (&out)[8]) {
        uint64_t d[4];
        auto w = [&](uint32_t i) -> uint64_t {
            uint64_t v = p[i];
            if (MODE == MODE_F64_CANON) v = gl::to_int(v);
            else if (MODE == MODE_F62_CANON) v = f62::mul(f62::norm(v), 1);
            return v;
        };
        k3::sha3_256_words(w, nelem, d);
        put(d, out);
    }
};

struct HRp64 {
    static constexpr bool WIDE = false;
    static constexpr bool BYTES = false;             // hash(bytes) = hash_elements over the 7-byte chunks (host conversion)
    static constexpr bool COOP = true;              // small batches: one state word per lane (rescue_coop.cuh)
    typedef rcoop::CoopRp64 Coop;
    // a Rescue merge is ~6400 modmuls (0.2 ms of one wave): the nearly empty upper levels of a multi-level workgroup
    // would serialise ten such latencies per workgroup, so the tree is built one full-width level per launch
    static constexpr uint32_t STAGE_LEVELS = 1;
    static constexpr bool WAVE_TREE = false;    // barrier-free wavefront Merkle stage (merkle_wave_kernel)
    static const char *row_name() { return "hash_rows_rp64"; }
    static const char *merkle_name() { return "merkle_stage_rp64"; }
    static const char *grind_name() { return "grind_rp64"; }
    static __device__ __forceinline__ void merge(const uint32_t (&in)[16], uint32_t (&out)[8]) {
        uint64_t two[8], d[4];
#pragma unroll
        for (int i = 0; i < 8; i++) two[i] = (uint64_t)in[2 * i] | ((uint64_t)in[2 * i + 1] << 32);
        rp64::merge(two, d);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            out[2 * i] = (uint32_t)d[i];
            out[2 * i + 1] = (uint32_t)(d[i] >> 32);
        }
    }
    // merge_with_int (rp64_256/mod.rs:198-219): seed in rate[0..4], value (split at the modulus) in rate[4..6],
    // capacity[0] = number of elements absorbed
    static __device__ __forceinline__ void merge_with_int(const uint32_t (&seed)[8], uint64_t value, uint32_t (&out)[8]) {
        uint64_t st[12];
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) st[4 + i] = (uint64_t)seed[2 * i] | ((uint64_t)seed[2 * i + 1] << 32);
        constexpr uint64_t R2 = 0xfffffffe00000001ull;                  // 2^128 mod p: BaseElement::new(v) = mont(v * R2)
        st[8] = gl::mul(value >= gl::P ? value - gl::P : value, R2);
        if (value < gl::P) st[0] = rp64::mont_small(5);
        else {
            st[9] = rp64::mont_small(1);                                // value / M = 1 for any u64 >= M
            st[0] = rp64::mont_small(6);
        }
        rp64::permute(st);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            out[2 * i] = (uint32_t)st[4 + i];
            out[2 * i + 1] = (uint32_t)(st[4 + i] >> 32);
        }
    }
    // ElementDigest::as_bytes starts with the canonical LE bytes of the first element (rp64_256/digest.rs)
    static __device__ __forceinline__ uint64_t head(const uint32_t (&d)[8]) {
        return gl::to_int((uint64_t)d[0] | ((uint64_t)d[1] << 32));
    }
    // ElementDigest::as_bytes: the canonical little-endian bytes of the four elements
    static __device__ __forceinline__ void as_bytes(const uint32_t (&d)[8], uint32_t (&b)[8]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t v = gl::to_int((uint64_t)d[2 * i] | ((uint64_t)d[2 * i + 1] << 32));
            b[2 * i] = (uint32_t)v;
            b[2 * i + 1] = (uint32_t)(v >> 32);
        }
    }
    template <int MODE, bool MULTI>
    static __device__ __forceinline__ void hash_elems(const uint64_t *p, uint32_t nelem, uint32_t (&out)[8]) {
        uint64_t d[4];
        auto e = [&](uint32_t i) -> uint64_t { return p[i]; };
        rp64::hash_elements(e, nelem, d);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            out[2 * i] = (uint32_t)d[i];
            out[2 * i + 1] = (uint32_t)(d[i] >> 32);
        }
    }
};

__device__ __forceinline__ void store_digest(void *dst, uint64_t idx, const uint32_t (&d)[8]) {
    uint4 *q = reinterpret_cast<uint4 *>(dst) + idx * 2;
    q[0] = make_uint4(d[0], d[1], d[2], d[3]);
    q[1] = make_uint4(d[4], d[5], d[6], d[7]);
}

__device__ __forceinline__ void load_pair(const void *src, uint64_t pair_idx, uint32_t (&m)[16]) {
    const uint4 *q = reinterpret_cast<const uint4 *>(src) + pair_idx * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint4 v = q[i];
        m[4 * i] = v.x;
        m[4 * i + 1] = v.y;
        m[4 * i + 2] = v.z;
        m[4 * i + 3] = v.w;
    }
}

// leaf[r * parts + k] = H(words [k*part_words, min((k+1)*part_words, words_per_row)) of row r).  The partition index is
// blockIdx.y, so the message length is uniform across a workgroup (scalar branches in the block loop).
template <class H, int MODE, bool MULTI>
__global__ __launch_bounds__(256) void hash_rows_kernel(const uint64_t *rows, uint64_t num_rows, uint64_t row_width,
                                                        uint32_t elems_per_row, uint32_t part_elems, uint32_t parts, void *out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= num_rows) return;
    const uint32_t k = blockIdx.y;
    const uint32_t e0 = k * part_elems;
    const uint32_t e1 = (e0 + part_elems < elems_per_row) ? e0 + part_elems : elems_per_row;
    uint32_t d[8];
    H::template hash_elems<MODE, MULTI>(rows + r * row_width + e0, e1 - e0, d);
    store_digest(out, r * parts + k, d);
}

template <class H>
__global__ __launch_bounds__(256) void merge_batch_kernel(const void *pairs, uint64_t count, void *out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    uint32_t m[16], d[8];
    load_pair(pairs, gid, m);
    H::merge(m, d);
    store_digest(out, gid, d);
}

struct Seed {
    uint32_t w[8];
};

// digest[i] = merge_with_int(seed, first + i)
template <class H>
__global__ __launch_bounds__(256) void merge_with_int_kernel(Seed seed, uint64_t first, uint64_t count, void *out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    uint32_t d[8];
    H::merge_with_int(seed.w, first + gid, d);
    store_digest(out, gid, d);
}

// Proof-of-work search (prover/src/channel.rs:169-185): lanes test nonces first + gid; every nonce whose new seed has
// >= `factor` trailing zero bits in its 8-byte head competes for the minimum, which is what the reference's serial
// `find` returns.
template <class H>
__global__ __launch_bounds__(256) void grind_kernel(Seed seed, uint64_t first, uint64_t count, uint32_t factor,
                                                    unsigned long long *best) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    uint32_t d[8];
    H::merge_with_int(seed.w, first + gid, d);
    const uint64_t h = H::head(d);
    const uint32_t tz = h ? (uint32_t)__builtin_ctzll(h) : 64u;
    if (tz >= factor) atomicMin(best, (unsigned long long)(first + gid));
}

// ---- crypto::DefaultRandomCoin with its state in device memory (crypto/src/random/default.rs) -------------------------------
// One lane: every step of the coin is one small hash that depends on the previous one.  What this buys is that a chain of
// commit -> reseed -> draw -> use (the FRI layers) is queued on the stream without a host round trip per link.
struct CoinState {
    uint32_t seed[8];
    uint64_t counter;
    uint32_t failed;     // a draw ran out of its 1000 tries (default.rs:185-199: FailedToDrawFieldElement)
    uint32_t pad;
};
static_assert(sizeof(CoinState) <= WF_COIN_BYTES, "WF_COIN_BYTES");

// reseed (default.rs:150-153): seed = merge(seed, data), counter = 0; the digest is also copied to root_out when given
template <class H>
__global__ void coin_reseed_kernel(CoinState *c, const uint32_t *digest, uint32_t *root_out) {
    uint32_t m[16], d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        m[i] = c->seed[i];
        m[8 + i] = digest[i];
    }
    H::merge(m, d);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c->seed[i] = d[i];
        if (root_out) root_out[i] = digest[i];
    }
    c->counter = 0;
}

// E::from_random_bytes over the first ELEMENT_BYTES of as_bytes: every base element must already be canonical
template <int FIELD, int D>
__device__ __forceinline__ bool coin_element(const uint32_t (&b)[8], uint64_t *out) {
    if constexpr (FIELD == WF_FIELD_F128) {
        uint64_t w[2 * D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const f128::u128 v = f128::join(b[4 * d], b[4 * d + 1], b[4 * d + 2], b[4 * d + 3]);
            if (v >= f128::modulus()) return false;
            w[2 * d] = (uint64_t)v;
            w[2 * d + 1] = (uint64_t)(v >> 64);
        }
#pragma unroll
        for (int i = 0; i < 2 * D; i++) out[i] = w[i];
    } else {
        uint64_t w[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint64_t v = (uint64_t)b[2 * d] | ((uint64_t)b[2 * d + 1] << 32);
            if constexpr (FIELD == WF_FIELD_F64) {
                if (v >= gl::P) return false;
                w[d] = gl::mul(v, 0xfffffffe00000001ull);   // BaseElement::new: times R^2 = 2^128 mod p
            } else {
                if (v >= f62::M) return false;
                w[d] = rp62::to_mont(v);
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) out[d] = w[d];
    }
    return true;
}

// draw::<E>() `count` times (default.rs:185-199): next() = merge_with_int(seed, ++counter) until the bytes decode
template <class H, int FIELD, int D>
__global__ void coin_draw_kernel(CoinState *c, uint32_t count, uint64_t *out) {
    constexpr int WORDS = (FIELD == WF_FIELD_F128 ? 2 : 1) * D;
    uint32_t seed[8];
#pragma unroll
    for (int i = 0; i < 8; i++) seed[i] = c->seed[i];
    uint64_t counter = c->counter;
    for (uint32_t k = 0; k < count; k++) {
        bool ok = false;
        for (int tries = 0; tries < 1000 && !ok; tries++) {
            uint32_t d[8], b[8];
            counter++;
            H::merge_with_int(seed, counter, d);
            H::as_bytes(d, b);
            ok = coin_element<FIELD, D>(b, out + (uint64_t)k * WORDS);
        }
        if (!ok) {
            c->failed = 1;
            break;
        }
    }
    c->counter = counter;
}

// commit_fri_layer + draw_fri_alpha in one launch: reseed with `digest`, then one draw
template <class H, int FIELD, int D>
__global__ void coin_reseed_draw_kernel(CoinState *c, const uint32_t *digest, uint32_t *root_out, uint64_t *out) {
    uint32_t m[16], seed[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        m[i] = c->seed[i];
        m[8 + i] = digest[i];
    }
    H::merge(m, seed);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c->seed[i] = seed[i];
        if (root_out) root_out[i] = digest[i];
    }
    uint64_t counter = 0;
    bool ok = false;
    for (int tries = 0; tries < 1000 && !ok; tries++) {
        uint32_t d[8], b[8];
        counter++;
        H::merge_with_int(seed, counter, d);
        H::as_bytes(d, b);
        ok = coin_element<FIELD, D>(b, out);
    }
    if (!ok) c->failed = 1;
    c->counter = counter;
}

template <class H>
int launch_coin_reseed_draw(wf_ctx *ctx, int field, uint32_t D, CoinState *c, const uint32_t *dg, uint32_t *cp, uint64_t *o) {
#define WF_RD(FIELD, DEG) hipLaunchKernelGGL((coin_reseed_draw_kernel<H, FIELD, DEG>), dim3(1), dim3(1), 0, ctx->stream, c, dg, cp, o)
    if (field == WF_FIELD_F128) {
        if (D == 1) WF_RD(WF_FIELD_F128, 1);
        else WF_RD(WF_FIELD_F128, 2);
    } else if (field == WF_FIELD_F64) {
        if (D == 1) WF_RD(WF_FIELD_F64, 1);
        else if (D == 2) WF_RD(WF_FIELD_F64, 2);
        else WF_RD(WF_FIELD_F64, 3);
    } else {
        if (D == 1) WF_RD(WF_FIELD_F62, 1);
        else if (D == 2) WF_RD(WF_FIELD_F62, 2);
        else WF_RD(WF_FIELD_F62, 3);
    }
#undef WF_RD
    return WF_OK;
}

template <class H>
int launch_coin_draw(wf_ctx *ctx, int field, uint32_t D, CoinState *c, uint32_t count, uint64_t *o) {
#define WF_DRAW(FIELD, DEG) hipLaunchKernelGGL((coin_draw_kernel<H, FIELD, DEG>), dim3(1), dim3(1), 0, ctx->stream, c, count, o)
    if (field == WF_FIELD_F128) {
        if (D == 1) WF_DRAW(WF_FIELD_F128, 1);
        else WF_DRAW(WF_FIELD_F128, 2);
    } else if (field == WF_FIELD_F64) {
        if (D == 1) WF_DRAW(WF_FIELD_F64, 1);
        else if (D == 2) WF_DRAW(WF_FIELD_F64, 2);
        else WF_DRAW(WF_FIELD_F64, 3);
    } else {
        if (D == 1) WF_DRAW(WF_FIELD_F62, 1);
        else if (D == 2) WF_DRAW(WF_FIELD_F62, 2);
        else WF_DRAW(WF_FIELD_F62, 3);
    }
#undef WF_DRAW
    return WF_OK;
}


// One stage of the tree: `count` input digests (a power of two), each workgroup reduces a chunk of
// CH = min(count, 1024) of them through log2(CH) levels.  Level d of the stage has count >> (d+1) nodes that
// live at heap indices [count >> (d+1), count >> d) of `nodes`.
template <class H>
__global__ __launch_bounds__(256) void merkle_stage_kernel(const void *in, void *nodes, uint64_t count, uint32_t log_ch) {
    __shared__ uint4 bufA[512 * 2];
    __shared__ uint4 bufB[256 * 2];
    const uint32_t ch = 1u << log_ch;
    const uint64_t wg = blockIdx.x;
    const int tid = threadIdx.x;
    // level 0: from global
    {
        const uint32_t cnt = ch >> 1;
        for (uint32_t i = tid; i < cnt; i += 256) {
            uint32_t m[16], d[8];
            load_pair(in, wg * cnt + i, m);
            H::merge(m, d);
            store_digest(nodes, (count >> 1) + wg * cnt + i, d);
            bufA[2 * i] = make_uint4(d[0], d[1], d[2], d[3]);
            bufA[2 * i + 1] = make_uint4(d[4], d[5], d[6], d[7]);
        }
    }
    uint4 *src = bufA, *dst = bufB;
    for (uint32_t lvl = 1; lvl < log_ch; lvl++) {
        __syncthreads();
        const uint32_t cnt = ch >> (lvl + 1);
        for (uint32_t i = tid; i < cnt; i += 256) {
            uint32_t m[16], d[8];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint4 v = src[4 * i + q];
                m[4 * q] = v.x;
                m[4 * q + 1] = v.y;
                m[4 * q + 2] = v.z;
                m[4 * q + 3] = v.w;
            }
            H::merge(m, d);
            store_digest(nodes, (count >> (lvl + 1)) + wg * cnt + i, d);
            dst[2 * i] = make_uint4(d[0], d[1], d[2], d[3]);
            dst[2 * i + 1] = make_uint4(d[4], d[5], d[6], d[7]);
        }
        uint4 *t = src;
        src = dst;
        dst = t;
    }
}

// The same for 4096 inputs per workgroup, 12 levels per launch.  In merkle_stage_kernel every level below 64 merges still costs
// one wavefront step (levels 4..9: six steps for 63 merges out of 21 per 1024 inputs); here a workgroup takes four 1024-input
// chunks through levels 0..3 one after the other, parks their 4 x 64 digests in LDS and runs the thin levels ONCE for all four:
// 69 wavefront steps for 4095 merges (93 % of lanes busy instead of 76 %), and a 2^23-leaf tree is two launches.
template <class H>
__global__ __launch_bounds__(256) void merkle_stage4k_kernel(const void *in, void *nodes, uint64_t count) {
    __shared__ uint4 bufA[512 * 2];
    __shared__ uint4 bufB[256 * 2];
    __shared__ uint4 top[256 * 2];
    const uint64_t wg = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    auto from_lds = [&](const uint4 *src, uint32_t i, uint32_t (&m)[16]) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = src[4 * i + q];
            m[4 * q] = v.x;
            m[4 * q + 1] = v.y;
            m[4 * q + 2] = v.z;
            m[4 * q + 3] = v.w;
        }
    };
    auto to_lds = [&](uint4 *dst, uint32_t i, const uint32_t (&d)[8]) {
        dst[2 * i] = make_uint4(d[0], d[1], d[2], d[3]);
        dst[2 * i + 1] = make_uint4(d[4], d[5], d[6], d[7]);
    };
    // level d (0-based) of this launch: count >> (d + 1) nodes at heap index (count >> (d + 1)) + position
    for (uint32_t q = 0; q < 4; q++) {
        const uint64_t base = wg * 4096 + q * 1024;             // first input of the chunk
        for (uint32_t i = tid; i < 512; i += 256) {             // level 0: from global
            uint32_t m[16], d[8];
            load_pair(in, (base >> 1) + i, m);
            H::merge(m, d);
            store_digest(nodes, (count >> 1) + (base >> 1) + i, d);
            to_lds(bufA, i, d);
        }
        __syncthreads();
        {                                                       // level 1: 256 merges
            uint32_t m[16], d[8];
            from_lds(bufA, tid, m);
            H::merge(m, d);
            store_digest(nodes, (count >> 2) + (base >> 2) + tid, d);
            to_lds(bufB, tid, d);
        }
        __syncthreads();
        if (tid < 128) {                                        // level 2
            uint32_t m[16], d[8];
            from_lds(bufB, tid, m);
            H::merge(m, d);
            store_digest(nodes, (count >> 3) + (base >> 3) + tid, d);
            to_lds(bufA, tid, d);
        }
        __syncthreads();
        if (tid < 64) {                                         // level 3 -> the chunk's 64 digests
            uint32_t m[16], d[8];
            from_lds(bufA, tid, m);
            H::merge(m, d);
            store_digest(nodes, (count >> 4) + (base >> 4) + tid, d);
            to_lds(top, q * 64 + tid, d);
        }
        __syncthreads();
    }
    uint4 *src = top, *dst = bufA;
    for (uint32_t lvl = 4; lvl < 12; lvl++) {                   // 256 -> 1
        const uint32_t cnt = 4096u >> (lvl + 1);
        if (tid < cnt) {
            uint32_t m[16], d[8];
            from_lds(src, tid, m);
            H::merge(m, d);
            store_digest(nodes, (count >> (lvl + 1)) + ((wg * 4096) >> (lvl + 1)) + tid, d);
            to_lds(dst, tid, d);
        }
        __syncthreads();
        uint4 *t = src;
        src = dst;
        dst = (t == top) ? bufB : t;
    }
}

// Barrier-free Merkle stage for the byte hashers: every WAVEFRONT owns a contiguous run of 128 * 2^T input digests and builds
// the T + 1 levels above them with all 64 lanes busy at every level, no LDS buffer and no workgroup barrier.
//   level 0: lane L merges input pair P0 + 64 b + L of batch b (coalesced 64-byte loads), b = 0 .. 2^T - 1;
//   level l: two level-(l-1) sets X, Y of 64 sibling-adjacent digests (Y follows X in the tree) are re-dealt so that lanes
//            0..31 hold the 32 sibling pairs of X and lanes 32..63 those of Y — ds_bpermute through the LDS crossbar (no LDS
//            memory), one gather per word and side after X / Y have been interleaved by lane parity (a quad-permute DPP move
//            and a select) — and merged: 64 merges,
//            64 consecutive nodes of level l, stored coalesced.
// The sets are produced depth first (a two-iteration loop per level, so the code holds T + 1 compressions, not 2^T), which keeps
// T digests live.  The stage kernels above serialise the thin upper levels on one wavefront behind barriers (a 4096-input
// workgroup's critical path is 28 compressions for 16 per wavefront of work: the BLAKE3 tree ran at half the 55e9
// compressions/s the arithmetic sustains, tools/microbench_blake3.hip); here a launch of 2^23 leaves is 31/32 of the tree at
// full lane utilisation.
// the two level-0 sets under one level-1 set: both loads are issued before the first compression, so that a wavefront has 128
// bytes per lane in flight while it hashes (level 0 is where the input stream enters)
template <class H>
struct WaveTreeLeaves {
    static __device__ __forceinline__ void build2(const void *in, void *nodes, uint64_t count, uint64_t pair0, uint32_t set, uint32_t lane,
                                                  uint32_t (&x)[8], uint32_t (&y)[8]) {
        uint32_t m0[16], m1[16];
        const uint64_t p0 = pair0 + (uint64_t)(2 * set) * 64 + lane, p1 = p0 + 64;
        load_pair(in, p0, m0);
        load_pair(in, p1, m1);
        H::merge(m0, x);
        store_digest(nodes, (count >> 1) + p0, x);
        H::merge(m1, y);
        store_digest(nodes, (count >> 1) + p1, y);
    }
};

template <class H, int L>
struct WaveTree {
    // returns, in d, lane `lane`'s node of the 64-node set number `set` (counted within the wave's run) of level L
    static __device__ __forceinline__ void build(const void *in, void *nodes, uint64_t count, uint64_t pair0, uint32_t set, uint32_t lane,
                                                 uint32_t (&d)[8]) {
        uint32_t x[8], y[8];
        if constexpr (L == 1) {
            WaveTreeLeaves<H>::build2(in, nodes, count, pair0, set, lane, x, y);
        } else {
#pragma unroll 1
            for (uint32_t h = 0; h < 2; h++) {
                uint32_t c[8];
                WaveTree<H, L - 1>::build(in, nodes, count, pair0, 2 * set + h, lane, c);
                if (h == 0) {
#pragma unroll
                    for (int w = 0; w < 8; w++) x[w] = c[w];
                } else {
#pragma unroll
                    for (int w = 0; w < 8; w++) y[w] = c[w];
                }
            }
        }
        // lanes < 32: (X[2 lane], X[2 lane + 1]); lanes >= 32: (Y[2 (lane - 32)], Y[2 (lane - 32) + 1]).
        // u = X on even lanes, Y[s - 1] on odd lanes s;  v = X on odd lanes, Y[s + 1] on even lanes s
        uint32_t m[16];
        const uint32_t j = lane & 31u, hi = lane >> 5;
        const int src_l = (int)((2 * j + hi) << 2), src_r = (int)((2 * j + 1 - hi) << 2);
        const bool odd = lane & 1u;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            // quad_perm [0,0,2,2]: odd lanes read their left neighbour; [1,1,3,3]: even lanes read their right neighbour
            const uint32_t yprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y[w], 0xA0 /* quad_perm:[0,0,2,2] */, 0xf, 0xf, false);
            const uint32_t ynext = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y[w], 0xF5 /* quad_perm:[1,1,3,3] */, 0xf, 0xf, false);
            const uint32_t u = odd ? yprev : x[w];
            const uint32_t v = odd ? x[w] : ynext;
            m[w] = (uint32_t)__builtin_amdgcn_ds_bpermute(src_l, (int)u);
            m[8 + w] = (uint32_t)__builtin_amdgcn_ds_bpermute(src_r, (int)v);
        }
        H::merge(m, d);
        store_digest(nodes, (count >> (L + 1)) + (pair0 >> L) + (uint64_t)set * 64 + lane, d);
    }
};
template <class H>
struct WaveTree<H, 0> {
    static __device__ __forceinline__ void build(const void *in, void *nodes, uint64_t count, uint64_t pair0, uint32_t set, uint32_t lane,
                                                 uint32_t (&d)[8]) {
        uint32_t m[16];
        const uint64_t pr = pair0 + (uint64_t)set * 64 + lane;
        load_pair(in, pr, m);
        H::merge(m, d);
        store_digest(nodes, (count >> 1) + pr, d);
    }
};

template <class H, int T>
__global__ __launch_bounds__(256) void merkle_wave_kernel(const void *in, void *nodes, uint64_t count) {
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t d[8];
    WaveTree<H, T>::build(in, nodes, count, wave << (6 + T), 0, lane, d);
}

__global__ void gather_rows_kernel(const uint8_t *rows, uint64_t row_bytes, uint32_t take_bytes, const uint64_t *pos,
                                   uint32_t count, uint8_t *out) {
    const uint32_t words = take_bytes / 8;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)count * words) return;
    const uint32_t r = (uint32_t)(gid / words), w = (uint32_t)(gid % words);
    reinterpret_cast<uint64_t *>(out)[gid] = reinterpret_cast<const uint64_t *>(rows + pos[r] * row_bytes)[w];
}

// Row hashes for rows of >= 64 bytes, BLAKE3 family.  One row per lane as in hash_rows_kernel, but a lane walking its own row
// reads 8 bytes per load at a row-sized stride (measured 0.26 TB/s on 256-byte rows: 32 columns x 2^23 rows took 8.4 ms for
// 1.3 ms of compressions).  Here a wavefront owns 64 consecutive rows and brings the message in one 64-byte block at a time:
// eight lanes load one row's block as 8 x u64 (64-byte contiguous segments, eight rows per load instruction), the values are
// canonicalised once by the loading lane, staged through 4.5 KiB of LDS per wavefront (wave-synchronous: DS operations of
// one wavefront execute in order), and every lane reads back its own row's 16 message words.
template <class H, int MODE>
__global__ __launch_bounds__(256) void hash_rows_wide_kernel(const uint64_t *rows, uint64_t num_rows, uint64_t row_width, uint32_t elems_per_row,
                                                             uint32_t part_elems, uint32_t parts, void *out) {
    constexpr int BW = H::WIDE_BW;                                    // 64-bit words per message block (BLAKE3 8, SHA3 17)
    constexpr int PITCH = BW | 1;                                     // odd: lanes reading their own rows hit distinct banks
    __shared__ uint64_t stage_all[4][64 * PITCH];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    volatile uint64_t *st = stage_all[wave];
    const uint64_t r_base = ((uint64_t)blockIdx.x * 4 + wave) * 64;
    if (r_base >= num_rows) return;                                   // the whole wavefront leaves together
    const uint32_t k = blockIdx.y;
    const uint32_t e0 = k * part_elems;
    const uint32_t e1 = (e0 + part_elems < elems_per_row) ? e0 + part_elems : elems_per_row;
    const uint32_t nelem = e1 - e0;
    auto fetch64 = [&](uint32_t blk, uint64_t (&m)[BW]) {
#pragma unroll
        for (uint32_t it = 0; it < BW; it++) {                        // 64 * BW words, 64 per step: runs of BW words per row
            const uint32_t idx = it * 64 + lane;
            const uint32_t rl = idx / BW, wq = idx - rl * BW;
            const uint32_t wi = blk * BW + wq;                        // word of the row part
            uint64_t row = r_base + rl;
            if (row >= num_rows) row = num_rows - 1;
            uint64_t v = 0;
            if (wi < nelem) {
                v = rows[row * row_width + e0 + wi];
                if (MODE == MODE_F64_CANON) v = gl::to_int(v);
                else if (MODE == MODE_F62_CANON) v = f62::mul(f62::norm(v), 1);
            }
            st[rl * PITCH + wq] = v;
        }
#pragma unroll
        for (int i = 0; i < BW; i++) m[i] = st[lane * PITCH + i];
    };
    uint32_t d[8];
    H::hash_wide(fetch64, nelem, d);
    if (r_base + lane < num_rows) store_digest(out, (r_base + lane) * parts + k, d);
}

// rows[r][c * W + w] = cols[c * col_words + r * W + w]  (W = 64-bit words per matrix element): column-major -> row-major
// through an LDS tile of R rows so that both the column reads (R * W consecutive words) and the row writes are coalesced
__global__ __launch_bounds__(256) void cols_to_rows_kernel(const uint64_t *cols, uint64_t col_words, uint32_t num_cols, uint32_t W,
                                                           uint64_t num_rows, uint32_t R, uint64_t *rows) {
    extern __shared__ uint64_t tile[];                 // [R][num_cols * W + 1]
    const uint32_t rw = num_cols * W, pitch = rw + 1;
    const uint64_t r0 = (uint64_t)blockIdx.x * R;
    const uint32_t nr = num_rows - r0 < R ? (uint32_t)(num_rows - r0) : R;
    const uint32_t run = nr * W;                       // consecutive words of one column in this tile
    for (uint32_t idx = threadIdx.x; idx < num_cols * run; idx += 256) {
        const uint32_t c = idx / run, k = idx % run;
        tile[(k / W) * pitch + c * W + (k % W)] = cols[(uint64_t)c * col_words + r0 * W + k];
    }
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < nr * rw; idx += 256) {
        const uint32_t r = idx / rw, j = idx % rw;
        rows[(r0 + r) * rw + j] = tile[r * pitch + j];
    }
}

template <class H, int MODE, bool MULTI>
int launch_hash_rows_t(wf_ctx *ctx, const uint64_t *rows, uint64_t num_rows, uint64_t row_width, uint32_t elems_per_row,
                       uint32_t part_elems, uint32_t parts, void *out) {
    const uint64_t blocks = (num_rows + 255) / 256;
    if (blocks > 0x7fffffffull || parts > 65535) return WF_ERR_DOMAIN_TOO_LARGE;
    if constexpr (H::WIDE) {
        if (MODE != MODE_DIGESTS && part_elems >= (uint32_t)H::WIDE_BW && elems_per_row >= (uint32_t)H::WIDE_BW) {
            wf_prof_begin(ctx, H::row_name());
            hipLaunchKernelGGL((hash_rows_wide_kernel<H, MODE>), dim3((uint32_t)blocks, parts), dim3(256), 0, ctx->stream, rows, num_rows, row_width,
                               elems_per_row, part_elems, parts, out);
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            return WF_OK;
        }
    }
    if constexpr (H::COOP) {
        if (num_rows * parts <= rcoop::COOP_MAX) {      // few rows: latency-bound, spread each state over 16 lanes
            wf_prof_begin(ctx, H::row_name());
            hipLaunchKernelGGL((rcoop::hash_rows_kernel<typename H::Coop>), dim3((uint32_t)((num_rows + 15) / 16), parts), dim3(256), 0,
                               ctx->stream, rows, num_rows, row_width, elems_per_row, part_elems, parts, (uint64_t *)out);
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            return WF_OK;
        }
    }
    wf_prof_begin(ctx, H::row_name());
    hipLaunchKernelGGL((hash_rows_kernel<H, MODE, MULTI>), dim3((uint32_t)blocks, parts), dim3(256), 0, ctx->stream, rows,
                       num_rows, row_width, elems_per_row, part_elems, parts, out);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

// all sizes in 64-bit words
template <class H>
int launch_hash_rows(wf_ctx *ctx, const uint64_t *rows, uint64_t num_rows, uint64_t row_width, uint32_t elems_per_row,
                     uint32_t part_elems, uint32_t parts, int mode, void *out) {
    const bool multi = part_elems > 128;   // more than one 1024-byte BLAKE3 chunk (ignored by Rescue)
#define WF_HR(MODE)                                                                                                              \
    return multi ? launch_hash_rows_t<H, MODE, true>(ctx, rows, num_rows, row_width, elems_per_row, part_elems, parts, out)       \
                 : launch_hash_rows_t<H, MODE, false>(ctx, rows, num_rows, row_width, elems_per_row, part_elems, parts, out)
    switch (mode) {
        case MODE_F64_CANON: WF_HR(MODE_F64_CANON);
        case MODE_F62_CANON: WF_HR(MODE_F62_CANON);
        case MODE_DIGESTS: WF_HR(MODE_DIGESTS);
        default: WF_HR(MODE_RAW);
    }
#undef WF_HR
}

// Narrow traces (one 8-column group per row): the LDE transpose of fft_api.hip (coset-major tmp[bc][u][m] -> row-major
// lde[(u + b*m)][8]) with the leaf hash folded in.  A tile holds b * TM <= 256 COMPLETE rows in LDS, so after the coalesced
// row-major store every lane hashes one row straight from the tile: the separate row-hash kernel, and its read of the
// whole matrix, disappear.  W = 64-bit words per base element.
template <class H, int MODE, class T>
__global__ __launch_bounds__(256) void lde_transpose_hash_kernel(const T *tmp, T *lde, uint32_t base_cols, uint32_t log_n, uint32_t log_b,
                                                                uint32_t log_tm, void *leaves) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lde_smem[];
    T *tile = reinterpret_cast<T *>(lde_smem);
    constexpr uint32_t W = sizeof(T) / 8;
    const uint64_t n = 1ull << log_n;
    const uint32_t b = 1u << log_b, TM = 1u << log_tm;
    const uint32_t row = b * 8 + 1;
    const uint64_t mt = blockIdx.x;
    const uint32_t total = b * 8 * TM;
    for (uint32_t idx = threadIdx.x; idx < total; idx += 256) {
        const uint32_t ml = idx & (TM - 1), uc = idx >> log_tm;
        const uint32_t u = uc & (b - 1), cl = uc >> log_b;
        const uint64_t m = (mt << log_tm) + ml;
        T v = 0;
        if (cl < base_cols && m < n) v = tmp[(((uint64_t)cl << log_b) + u) * n + m];
        tile[ml * row + u * 8 + cl] = v;
    }
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < total; idx += 256) {
        const uint32_t cl = idx & 7, u = (idx >> 3) & (b - 1), ml = idx >> (3 + log_b);
        const uint64_t m = (mt << log_tm) + ml;
        if (m < n) lde[(u + ((uint64_t)m << log_b)) * 8 + cl] = tile[ml * row + u * 8 + cl];
    }
    for (uint32_t r = threadIdx.x; r < b * TM; r += 256) {
        const uint32_t u = r & (b - 1), ml = r >> log_b;
        const uint64_t m = (mt << log_tm) + ml;
        if (m >= n) continue;
        uint32_t d[8];
        H::template hash_elems<MODE, false>(reinterpret_cast<const uint64_t *>(tile + ml * row + u * 8), base_cols * W, d);
        store_digest(leaves, u + (m << log_b), d);
    }
}

template <class H, class T>
int launch_lde_transpose_hash(wf_ctx *ctx, int mode, const void *tmp, void *lde, uint32_t base_cols, uint32_t log_n, uint32_t log_b,
                              uint32_t log_tm, void *leaves) {
    const uint64_t n = 1ull << log_n;
    const uint64_t blocks = (n + (1ull << log_tm) - 1) >> log_tm;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    const size_t lds = ((size_t)1 << log_tm) * ((8u << log_b) + 1) * sizeof(T);
    wf_prof_begin(ctx, "lde_transpose_hash");
#define WF_LT(MODE) hipLaunchKernelGGL((lde_transpose_hash_kernel<H, MODE, T>), dim3((uint32_t)blocks), dim3(256), lds, ctx->stream, (const T *)tmp, (T *)lde, base_cols, log_n, log_b, log_tm, leaves)
    switch (mode) {
        case MODE_F64_CANON: WF_LT(MODE_F64_CANON); break;
        case MODE_F62_CANON: WF_LT(MODE_F62_CANON); break;
        default: WF_LT(MODE_RAW); break;
    }
#undef WF_LT
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

// FRI layer commit, first half, in one pass (fri/src/prover/mod.rs:321-336 = transpose_slice + hash each row):
//   tr[i][j] = ev[i + j * rc]   and   leaf_i = H::hash_elements(tr[i]).
// A workgroup takes R consecutive rows: the N strided runs of R elements are read coalesced into an LDS tile
// [R][row_words + 1], every lane hashes its row straight out of the tile, and the tile is written to the transposed matrix
// as ONE contiguous block.  (The separate transpose wrote 8 or 16 bytes per lane at a row-sized stride — 1 TB/s — and the
// row hash then read the matrix back.)  EW = 64-bit words per element (ext_degree * words per base element).
template <class H, int MODE>
__global__ __launch_bounds__(256) void fri_rows_kernel(const uint64_t *ev, uint64_t rc, uint32_t N, uint32_t EW, uint32_t R, uint64_t *tr,
                                                       void *leaves) {
    extern __shared__ uint64_t fri_tile[];
    const uint32_t row_words = N * EW, pitch = row_words + 1;
    const uint64_t r0 = (uint64_t)blockIdx.x * R;
    const uint32_t nr = rc - r0 < R ? (uint32_t)(rc - r0) : R;
    const uint32_t run = nr * EW;                        // consecutive words of one strided run
    for (uint32_t idx = threadIdx.x; idx < N * run; idx += 256) {
        const uint32_t j = idx / run, k = idx - j * run;
        fri_tile[(k / EW) * pitch + j * EW + (k % EW)] = ev[(r0 + (uint64_t)j * rc) * EW + k];
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < nr; t += 256) {
        uint32_t d[8];
        H::template hash_elems<MODE, false>(fri_tile + t * pitch, row_words, d);
        store_digest(leaves, r0 + t, d);
    }
    for (uint32_t idx = threadIdx.x; idx < nr * row_words; idx += 256) {
        const uint32_t r = idx / row_words, w = idx - r * row_words;
        tr[r0 * row_words + idx] = fri_tile[r * pitch + w];
    }
}

template <class H>
int launch_fri_rows(wf_ctx *ctx, int mode, const uint64_t *ev, uint64_t rc, uint32_t N, uint32_t EW, uint64_t *tr, void *leaves) {
    const uint32_t pitch = N * EW + 1;
    uint32_t R = 256;
    while (R > 32 && (size_t)R * pitch * 8 > 40960) R >>= 1;
    const uint64_t blocks = (rc + R - 1) / R;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    const size_t lds = (size_t)R * pitch * 8;
    wf_prof_begin(ctx, "fri_transpose_hash");
#define WF_FR(MODE) hipLaunchKernelGGL((fri_rows_kernel<H, MODE>), dim3((uint32_t)blocks), dim3(256), lds, ctx->stream, ev, rc, N, EW, R, tr, leaves)
    switch (mode) {
        case MODE_F64_CANON: WF_FR(MODE_F64_CANON); break;
        case MODE_F62_CANON: WF_FR(MODE_F62_CANON); break;
        default: WF_FR(MODE_RAW); break;
    }
#undef WF_FR
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

template <class H>
int launch_merkle(wf_ctx *ctx, const void *leaves, uint64_t num_leaves, void *nodes) {
    WF_HIP(hipMemsetAsync(nodes, 0, 32, ctx->stream));  // nodes[0] = Digest::default()
    const uint8_t *in = (const uint8_t *)leaves;
    uint64_t count = num_leaves;
    while (count > 1) {
        if (H::STAGE_LEVELS == 1) {
            // one level: nodes[count/2 + i] = merge(in[2i], in[2i+1]), one merge per lane
            const uint64_t half = count >> 1;
            const uint64_t blocks = (half + 255) / 256;
            if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
            wf_prof_begin(ctx, H::merkle_name());
            bool coop = false;
            if constexpr (H::COOP) {
                if (half <= rcoop::COOP_MAX) {           // the upper levels: one 0.2 ms wave per 64 merges otherwise
                    coop = true;
                    hipLaunchKernelGGL((rcoop::merge_kernel<typename H::Coop>), dim3((uint32_t)((half + 15) / 16)), dim3(256), 0, ctx->stream,
                                       (const uint64_t *)in, half, (uint64_t *)((uint8_t *)nodes + half * 32));
                }
            }
            if (!coop)
                hipLaunchKernelGGL(merge_batch_kernel<H>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, (const void *)in, half,
                                   (void *)((uint8_t *)nodes + half * 32));
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            count = half;
            in = (const uint8_t *)nodes + count * 32;
            continue;
        }
        if (H::WAVE_TREE && count >= (1u << 20)) {
            // T + 1 levels per launch, 128 * 2^T inputs per wavefront, all lanes busy at every level; T as large as still leaves
            // four wavefronts per SIMD (4096 on the chip): 2^23 inputs and up take five levels per launch
            uint32_t lg = 0;
            while ((2ull << lg) <= count) lg++;
            const uint32_t T = lg >= 23 ? 4 : lg - 19;
            const uint64_t waves = count >> (7 + T);
            if ((waves + 3) / 4 > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
            wf_prof_begin(ctx, H::merkle_name());
            if constexpr (H::WAVE_TREE) {
                const dim3 grid((uint32_t)((waves + 3) / 4));
                switch (T) {
                    case 1: hipLaunchKernelGGL((merkle_wave_kernel<H, 1>), grid, dim3(256), 0, ctx->stream, (const void *)in, nodes, count); break;
                    case 2: hipLaunchKernelGGL((merkle_wave_kernel<H, 2>), grid, dim3(256), 0, ctx->stream, (const void *)in, nodes, count); break;
                    case 3: hipLaunchKernelGGL((merkle_wave_kernel<H, 3>), grid, dim3(256), 0, ctx->stream, (const void *)in, nodes, count); break;
                    default: hipLaunchKernelGGL((merkle_wave_kernel<H, 4>), grid, dim3(256), 0, ctx->stream, (const void *)in, nodes, count); break;
                }
            }
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            count >>= T + 1;
            in = (const uint8_t *)nodes + count * 32;
            continue;
        }
        if (H::STAGE_LEVELS >= 10 && count >= (1u << 20)) {      // 12 levels per launch, 4096 inputs per workgroup: only when that still fills the chip (>= 256 workgroups)
            const uint64_t wgs4 = count >> 12;
            if (wgs4 > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
            wf_prof_begin(ctx, H::merkle_name());
            hipLaunchKernelGGL(merkle_stage4k_kernel<H>, dim3((uint32_t)wgs4), dim3(256), 0, ctx->stream, (const void *)in, nodes, count);
            wf_prof_end(ctx);
            WF_HIP(hipGetLastError());
            count = wgs4;
            in = (const uint8_t *)nodes + count * 32;
            continue;
        }
        uint32_t log_ch = 0;
        while ((1ull << log_ch) < count && log_ch < H::STAGE_LEVELS) log_ch++;
        const uint64_t wgs = count >> log_ch;
        if (wgs > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
        wf_prof_begin(ctx, H::merkle_name());
        hipLaunchKernelGGL(merkle_stage_kernel<H>, dim3((uint32_t)wgs), dim3(256), 0, ctx->stream, (const void *)in, nodes,
                           count, log_ch);
        wf_prof_end(ctx);
        WF_HIP(hipGetLastError());
        count = wgs;
        in = (const uint8_t *)nodes + count * 32;  // this stage's top level = next stage's inputs
    }
    return WF_OK;
}

int check_hash(int hash) {
    return (hash >= WF_HASH_BLAKE3_256 && hash <= WF_HASH_BLAKE3_192) ? WF_OK : WF_ERR_UNSUPPORTED;
}

// run fn(H{}) with the hasher policy selected by `hash`
template <class FN>
int with_hasher(int hash, FN &&fn) {
    switch (hash) {
        case WF_HASH_BLAKE3_256: return fn(HBlake3{});
        case WF_HASH_RP64_256: return fn(HRp64{});
        case WF_HASH_SHA3_256: return fn(HSha3{});
        case WF_HASH_RPJIVE64_256: return fn(HRpJive{});
        case WF_HASH_RP62_248: return fn(HRp62{});
        case WF_HASH_BLAKE3_192: return fn(HBlake3_192{});
        default: return WF_ERR_UNSUPPORTED;
    }
}

}  // namespace

// used by wf_build_trace_commitment through wf_evaluate_polys_over_fused (fft_api.hip).  Only the byte hashers take the fused
// path: a Rescue row hash is three orders of magnitude more arithmetic than the transpose it would be fused with.
int wf_lde_transpose_hash(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_tmp, void *d_lde, uint32_t base_cols,
                          uint32_t log_n, uint32_t log_b, uint32_t log_tm, void *d_leaves, int *done) {
    *done = 0;
    if (hash != WF_HASH_BLAKE3_256 && hash != WF_HASH_BLAKE3_192 && hash != WF_HASH_SHA3_256) return WF_OK;
    if (!d_leaves) return WF_ERR_INVALID_ARG;
    (void)ext_degree;
    const int mode = field == WF_FIELD_F64 ? MODE_F64_CANON : (field == WF_FIELD_F62 ? MODE_F62_CANON : MODE_RAW);
    *done = 1;
    return with_hasher(hash, [&](auto h) {
        typedef decltype(h) H;
        if (field == WF_FIELD_F128) return launch_lde_transpose_hash<H, f128::u128>(ctx, mode, d_tmp, d_lde, base_cols, log_n, log_b, log_tm, d_leaves);
        return launch_lde_transpose_hash<H, uint64_t>(ctx, mode, d_tmp, d_lde, base_cols, log_n, log_b, log_tm, d_leaves);
    });
}

// used by wf_fri_layer_commit (fri.hip): *done = 0 when the caller should take the unfused path (small Rescue layers, where the
// lane-cooperative row hash wins)
int wf_fri_transpose_hash(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_evals, uint32_t log_rc, uint32_t log_nf,
                          void *d_transposed, void *d_leaves, int *done) {
    *done = 0;
    WF_TRY(check_hash(hash));
    if (field != WF_FIELD_F64 && (hash == WF_HASH_RP64_256 || hash == WF_HASH_RPJIVE64_256)) return WF_ERR_UNSUPPORTED;
    if (field != WF_FIELD_F62 && hash == WF_HASH_RP62_248) return WF_ERR_UNSUPPORTED;
    const uint64_t rc = 1ull << log_rc;
    const bool rescue = hash == WF_HASH_RP64_256 || hash == WF_HASH_RPJIVE64_256 || hash == WF_HASH_RP62_248;
    if (rescue && rc <= rcoop::COOP_MAX) return WF_OK;
    const int mode = field == WF_FIELD_F64 ? MODE_F64_CANON : (field == WF_FIELD_F62 ? MODE_F62_CANON : MODE_RAW);
    const uint32_t EW = ext_degree * (field == WF_FIELD_F128 ? 2 : 1);
    *done = 1;
    return with_hasher(hash, [&](auto h) {
        return launch_fri_rows<decltype(h)>(ctx, mode, (const uint64_t *)d_evals, rc, 1u << log_nf, EW, (uint64_t *)d_transposed, d_leaves);
    });
}

extern "C" int wf_merkle_build(wf_ctx *ctx, int hash, const void *d_leaves, uint64_t num_leaves, void *d_nodes) {
    if (!ctx || !d_leaves || !d_nodes) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (num_leaves < 2) return WF_ERR_TOO_FEW_LEAVES;
    if (num_leaves & (num_leaves - 1)) return WF_ERR_NOT_POWER_OF_TWO;
    return with_hasher(hash, [&](auto h) { return launch_merkle<decltype(h)>(ctx, d_leaves, num_leaves, d_nodes); });
}

extern "C" int wf_hash_merge_batch(wf_ctx *ctx, int hash, const void *d_pairs, uint64_t count, void *d_out) {
    if (!ctx || !d_pairs || !d_out) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (count == 0) return WF_OK;
    const uint64_t blocks = (count + 255) / 256;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    WF_TRY(with_hasher(hash, [&](auto h) {
        typedef decltype(h) H;
        if constexpr (H::COOP) {
            if (count <= rcoop::COOP_MAX) {
                hipLaunchKernelGGL((rcoop::merge_kernel<typename H::Coop>), dim3((uint32_t)((count + 15) / 16)), dim3(256), 0, ctx->stream,
                                   (const uint64_t *)d_pairs, count, (uint64_t *)d_out);
                return (int)WF_OK;
            }
        }
        hipLaunchKernelGGL(merge_batch_kernel<H>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, d_pairs, count, d_out);
        return (int)WF_OK;
    }));
    WF_HIP(hipGetLastError());
    return WF_OK;
}

static int hash_rows_impl(wf_ctx *ctx, int hash, int field, uint32_t D, const void *d_rows, uint64_t num_rows,
                          uint64_t row_width, uint32_t elems_per_row, uint32_t num_partitions, uint32_t hash_rate,
                          void *d_leaves) {
    if (!ctx || !d_rows || !d_leaves || num_rows == 0 || D == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128 && field != WF_FIELD_F62) return WF_ERR_UNSUPPORTED;
    if (field != WF_FIELD_F64 && (hash == WF_HASH_RP64_256 || hash == WF_HASH_RPJIVE64_256)) return WF_ERR_UNSUPPORTED;   // Rescue over f64 only
    if (field != WF_FIELD_F62 && hash == WF_HASH_RP62_248) return WF_ERR_UNSUPPORTED;                                       // Rescue over f62 only
    if (elems_per_row > row_width || elems_per_row % D) return WF_ERR_INVALID_ARG;
    if (num_partitions < 1 || num_partitions > 16 || hash_rate < 1 || hash_rate > 256) return WF_ERR_INVALID_ARG;
    hash_rate &= 0xffu;   // PartitionOptions::new stores `hash_rate as u8` (air/src/options.rs:414-418): the permitted 256 wraps to 0
    // f64 is not IS_CANONICAL: hash the canonical LE bytes; f128 is: hash the raw element bytes (blake/mod.rs:52-65).
    // Rows are addressed in 64-bit words: an f128 element is two words.
    const int mode = field == WF_FIELD_F64 ? MODE_F64_CANON : (field == WF_FIELD_F62 ? MODE_F62_CANON : MODE_RAW);
    const uint32_t W = field == WF_FIELD_F128 ? 2 : 1;
    row_width *= W;
    elems_per_row *= W;
    D *= W;
    const uint64_t *rows = (const uint64_t *)d_rows;
    // PartitionOptions::partition_size / num_partitions — air/src/options.rs:428-444 (in columns of E)
    const uint32_t num_cols = elems_per_row / D;
    uint32_t ps = num_cols;
    if (num_partitions > 1) {
        const uint32_t min_ps = hash_rate / (D / W);
        ps = (num_cols + num_partitions - 1) / num_partitions;
        if (ps < min_ps) ps = min_ps;
    }
    // row_matrix.rs:193: the single-hash path is taken only when partition_size == num_cols; a partition size ABOVE the
    // column count (the hash_rate floor with num_partitions > 1) still goes through merge_many over its one digest
    if (ps == num_cols) {
        return with_hasher(hash, [&](auto h) {
            return launch_hash_rows<decltype(h)>(ctx, rows, num_rows, row_width, elems_per_row, elems_per_row, 1, mode, d_leaves);
        });
    }
    const uint32_t parts = (num_cols + ps - 1) / ps;
    void *tmp;
    WF_TRY(wf_scratch(ctx, 2, (size_t)num_rows * parts * 32, &tmp));
    // partition digests, then leaf = merge_many(partition digests): Blake3 hashes the raw digest bytes
    // (blake/mod.rs:37-39), Rp64_256 hashes the digests' 4*parts elements (rp64_256/mod.rs:194-196).
    return with_hasher(hash, [&](auto h) {
        typedef decltype(h) H;
        WF_TRY(launch_hash_rows<H>(ctx, rows, num_rows, row_width, elems_per_row, ps * D, parts, mode, tmp));
        return launch_hash_rows<H>(ctx, (const uint64_t *)tmp, num_rows, 4ull * parts, 4 * parts, 4 * parts, 1, MODE_DIGESTS, d_leaves);
    });
}

extern "C" int wf_hash_rows(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_rows, uint64_t num_rows,
                            uint64_t row_width, uint32_t elems_per_row, uint32_t num_partitions, uint32_t hash_rate,
                            void *d_leaves) {
    return hash_rows_impl(ctx, hash, field, ext_degree, d_rows, num_rows, row_width, elems_per_row, num_partitions,
                          hash_rate, d_leaves);
}

namespace {
template <class H>
__global__ __launch_bounds__(256) void hash_bytes_kernel(const uint64_t *msgs, uint64_t count, uint64_t stride_words, uint64_t nbytes, void *out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    uint32_t d[8];
    if constexpr (H::BYTES) H::hash_bytes(msgs + gid * stride_words, nbytes, d);
    store_digest(out, gid, d);
}
}  // namespace

extern "C" int wf_hash_bytes_batch(wf_ctx *ctx, int hash, const void *d_msgs, uint64_t count, uint64_t stride_bytes, uint64_t len_bytes,
                                   void *d_out) {
    if (!ctx || !d_out || (count && len_bytes && !d_msgs)) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (hash != WF_HASH_BLAKE3_256 && hash != WF_HASH_BLAKE3_192 && hash != WF_HASH_SHA3_256) return WF_ERR_UNSUPPORTED;
    if (stride_bytes % 8 || stride_bytes < ((len_bytes + 7) / 8) * 8 || len_bytes >= (1ull << 32)) return WF_ERR_INVALID_ARG;
    if (count == 0) return WF_OK;
    const uint64_t blocks = (count + 255) / 256;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    WF_TRY(with_hasher(hash, [&](auto h) {
        hipLaunchKernelGGL(hash_bytes_kernel<decltype(h)>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, (const uint64_t *)d_msgs, count,
                           stride_bytes / 8, len_bytes, d_out);
        return (int)WF_OK;
    }));
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_hash_columns(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, const void *d_cols, uint32_t num_cols,
                               uint64_t col_stride, uint64_t num_rows, void *d_leaves) {
    if (!ctx || !d_cols || !d_leaves || num_cols == 0 || num_rows == 0 || ext_degree == 0) return WF_ERR_INVALID_ARG;
    if (col_stride < num_rows * ext_degree) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    const uint32_t fw = field == WF_FIELD_F128 ? 2 : 1;          // 64-bit words per base element
    const uint32_t W = ext_degree * fw;
    const uint64_t rw = (uint64_t)num_cols * W;
    if (rw > 4095) return WF_ERR_UNSUPPORTED;                    // one row must fit the LDS tile
    uint32_t R = (uint32_t)(4096 / (rw + 1));                   // <= 32 KiB of LDS
    if (R > 256) R = 256;
    if (R == 0) R = 1;
    void *tmp;
    WF_TRY(wf_scratch(ctx, 1, (size_t)num_rows * rw * 8, &tmp));
    const uint64_t blocks = (num_rows + R - 1) / R;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    wf_prof_begin(ctx, "cols_to_rows");
    hipLaunchKernelGGL(cols_to_rows_kernel, dim3((uint32_t)blocks), dim3(256), (size_t)R * (rw + 1) * 8, ctx->stream, (const uint64_t *)d_cols,
                       col_stride * fw, num_cols, W, num_rows, R, (uint64_t *)tmp);
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    const uint32_t bc = num_cols * ext_degree;                   // base elements per row
    return hash_rows_impl(ctx, hash, field, ext_degree, tmp, num_rows, bc, bc, 1, 1, d_leaves);
}

extern "C" int wf_hash_elements_batch(wf_ctx *ctx, int hash, int field, const void *d_elems, uint64_t count,
                                      uint64_t row_width, uint32_t elems_per_row, void *d_out) {
    if (count == 0) return WF_OK;
    return hash_rows_impl(ctx, hash, field, 1, d_elems, count, row_width, elems_per_row, 1, 1, d_out);
}

extern "C" int wf_hash_merge_many_batch(wf_ctx *ctx, int hash, const void *d_digests, uint64_t count, uint32_t k, void *d_out) {
    if (!ctx || !d_digests || !d_out || k == 0) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (count == 0) return WF_OK;
    // Blake3: hash of the concatenated digest bytes (blake/mod.rs:37-39); Rp64_256: hash_elements over the 4*k digest
    // elements (rp64_256/mod.rs:194-196).  Both are "raw words" for the row kernel.
    return with_hasher(hash, [&](auto h) {
        return launch_hash_rows<decltype(h)>(ctx, (const uint64_t *)d_digests, count, 4ull * k, 4 * k, 4 * k, 1, MODE_DIGESTS, d_out);
    });
}

extern "C" int wf_hash_merge_with_int_batch(wf_ctx *ctx, int hash, const void *h_seed, uint64_t first_value, uint64_t count,
                                           void *d_out) {
    if (!ctx || !h_seed || !d_out) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (count == 0) return WF_OK;
    if (count - 1 > ~0ull - first_value) return WF_ERR_INVALID_ARG;   // the range must not wrap past u64::MAX
    const uint64_t blocks = (count + 255) / 256;
    if (blocks > 0x7fffffffull) return WF_ERR_DOMAIN_TOO_LARGE;
    Seed seed;
    memcpy(seed.w, h_seed, 32);
    WF_TRY(with_hasher(hash, [&](auto h) {
        typedef decltype(h) H;
        if constexpr (H::COOP) {
            if (count <= rcoop::COOP_MAX) {
                rcoop::SeedWords sw;
                memcpy(sw.w, seed.w, 32);
                hipLaunchKernelGGL((rcoop::merge_with_int_kernel<typename H::Coop>), dim3((uint32_t)((count + 15) / 16)), dim3(256), 0,
                                   ctx->stream, sw, first_value, count, (uint64_t *)d_out);
                return (int)WF_OK;
            }
        }
        hipLaunchKernelGGL(merge_with_int_kernel<H>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, seed, first_value, count, d_out);
        return (int)WF_OK;
    }));
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_coin_init(wf_ctx *ctx, void *d_coin, const void *h_seed) {
    if (!ctx || !d_coin || !h_seed) return WF_ERR_INVALID_ARG;
    CoinState st;
    memset(&st, 0, sizeof(st));
    memcpy(st.seed, h_seed, 32);
    uint8_t image[WF_COIN_BYTES] = {0};
    memcpy(image, &st, sizeof(st));
    WF_HIP(hipMemcpyAsync(d_coin, image, WF_COIN_BYTES, hipMemcpyHostToDevice, ctx->stream));
    WF_HIP(hipStreamSynchronize(ctx->stream));    // `image` is on this frame
    return WF_OK;
}

extern "C" int wf_coin_reseed(wf_ctx *ctx, int hash, void *d_coin, const void *d_digest, void *d_digest_copy) {
    if (!ctx || !d_coin || !d_digest) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    wf_prof_begin(ctx, "coin");
    WF_TRY(with_hasher(hash, [&](auto h) {
        hipLaunchKernelGGL(coin_reseed_kernel<decltype(h)>, dim3(1), dim3(1), 0, ctx->stream, (CoinState *)d_coin, (const uint32_t *)d_digest,
                           (uint32_t *)d_digest_copy);
        return (int)WF_OK;
    }));
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_coin_draw(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, void *d_coin, uint32_t count, void *d_out) {
    if (!ctx || !d_coin || !d_out) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    const uint32_t max_ext = field == WF_FIELD_F128 ? 2 : 3;       // 32 digest bytes hold two f128 or three 64-bit elements
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128 && field != WF_FIELD_F62) return WF_ERR_UNSUPPORTED;
    if (ext_degree < 1 || ext_degree > max_ext) return WF_ERR_UNSUPPORTED;
    if (count == 0) return WF_OK;
    wf_prof_begin(ctx, "coin");
    WF_TRY(with_hasher(hash, [&](auto h) { return launch_coin_draw<decltype(h)>(ctx, field, ext_degree, (CoinState *)d_coin, count, (uint64_t *)d_out); }));
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_coin_reseed_draw(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, void *d_coin, const void *d_digest, void *d_digest_copy,
                                  void *d_out) {
    if (!ctx || !d_coin || !d_digest || !d_out) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (field != WF_FIELD_F64 && field != WF_FIELD_F128 && field != WF_FIELD_F62) return WF_ERR_UNSUPPORTED;
    if (ext_degree < 1 || ext_degree > (field == WF_FIELD_F128 ? 2u : 3u)) return WF_ERR_UNSUPPORTED;
    wf_prof_begin(ctx, "coin");
    WF_TRY(with_hasher(hash, [&](auto h) {
        return launch_coin_reseed_draw<decltype(h)>(ctx, field, ext_degree, (CoinState *)d_coin, (const uint32_t *)d_digest, (uint32_t *)d_digest_copy,
                                                    (uint64_t *)d_out);
    }));
    wf_prof_end(ctx);
    WF_HIP(hipGetLastError());
    return WF_OK;
}

extern "C" int wf_coin_read(wf_ctx *ctx, const void *d_coin, void *h_seed, uint64_t *h_counter) {
    if (!ctx || !d_coin || !h_seed || !h_counter) return WF_ERR_INVALID_ARG;
    CoinState st;
    WF_HIP(hipMemcpyAsync(&st, d_coin, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    WF_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(h_seed, st.seed, 32);
    *h_counter = st.counter;
    return st.failed ? WF_ERR_NOT_FOUND : WF_OK;
}

extern "C" int wf_grind(wf_ctx *ctx, int hash, const void *h_seed, uint32_t grinding_factor, uint64_t first_nonce,
                        uint64_t max_nonce, uint64_t *h_nonce) {
    if (!ctx || !h_seed || !h_nonce || grinding_factor > 64 || first_nonce > max_nonce) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    Seed seed;
    memcpy(seed.w, h_seed, 32);
    void *tmp;
    WF_TRY(wf_scratch(ctx, 2, 8, &tmp));
    unsigned long long *d_best = (unsigned long long *)tmp;
    // batch ~ twice the expected number of trials, within [2^16, 2^24] lanes per launch
    uint32_t lg = grinding_factor + 1;
    if (lg < 16) lg = 16;
    if (lg > 24) lg = 24;
    if ((hash == WF_HASH_RP64_256 || hash == WF_HASH_RPJIVE64_256 || hash == WF_HASH_RP62_248) && lg > 21) lg = 21;   // a Rescue permutation is ~200x a BLAKE3 block
    if (hash == WF_HASH_SHA3_256 && lg > 23) lg = 23;
    const uint64_t batch = 1ull << lg;
    uint64_t first = first_nonce;
    for (;;) {
        const uint64_t left = max_nonce - first;               // nonces first .. max_nonce inclusive = left + 1
        const uint64_t count = left < batch - 1 ? left + 1 : batch;
        WF_HIP(hipMemsetAsync(d_best, 0xff, 8, ctx->stream));
        WF_TRY(with_hasher(hash, [&](auto h) {
            typedef decltype(h) H;
            wf_prof_begin(ctx, H::grind_name());
            hipLaunchKernelGGL(grind_kernel<H>, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, seed, first, count,
                               grinding_factor, d_best);
            wf_prof_end(ctx);
            return (int)WF_OK;
        }));
        WF_HIP(hipGetLastError());
        unsigned long long best;
        WF_HIP(hipMemcpyAsync(&best, d_best, 8, hipMemcpyDeviceToHost, ctx->stream));
        WF_HIP(hipStreamSynchronize(ctx->stream));
        if (best != ~0ull) {
            *h_nonce = best;
            return WF_OK;
        }
        if (count == left + 1) return WF_ERR_NOT_FOUND;
        first += count;
    }
}

extern "C" int wf_rows_fetch(wf_ctx *ctx, const void *d_rows, uint64_t row_width, uint32_t elems_per_row,
                             uint32_t elem_bytes, const uint64_t *h_positions, uint32_t count, void *h_out) {
    if (!ctx || !d_rows || !h_positions || !h_out || (elem_bytes != 8 && elem_bytes != 16)) return WF_ERR_INVALID_ARG;
    if (count == 0) return WF_OK;
    const uint64_t row_bytes = row_width * elem_bytes;
    const uint32_t take = elems_per_row * elem_bytes;
    void *tmp;
    WF_TRY(wf_scratch(ctx, 2, (size_t)count * (take + 8), &tmp));
    uint64_t *d_pos = (uint64_t *)tmp;
    uint8_t *d_out = (uint8_t *)tmp + (size_t)count * 8;
    WF_HIP(hipMemcpyAsync(d_pos, h_positions, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t total = (uint64_t)count * (take / 8);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint8_t *)d_rows, row_bytes, take, d_pos, count, d_out);
    WF_HIP(hipGetLastError());
    WF_HIP(hipMemcpyAsync(h_out, d_out, (size_t)count * take, hipMemcpyDeviceToHost, ctx->stream));
    WF_HIP(hipStreamSynchronize(ctx->stream));
    return WF_OK;
}

extern "C" int wf_build_trace_commitment(wf_ctx *ctx, int hash, int field, uint32_t ext_degree, void *d_trace,
                                         uint32_t num_cols, uint64_t col_stride, uint32_t log_n, uint32_t log_blowup,
                                         const void *h_offset, uint32_t num_partitions, uint32_t hash_rate,
                                         int skip_interpolate, void *d_lde, void *d_leaves, void *d_nodes, void *h_root) {
    if (!ctx || !d_trace || !d_lde || !d_leaves || !d_nodes) return WF_ERR_INVALID_ARG;
    WF_TRY(check_hash(hash));
    if (ext_degree == 0 || num_cols == 0 || num_partitions < 1 || num_partitions > 16 || hash_rate < 1 || hash_rate > 256) return WF_ERR_INVALID_ARG;
    // extend_execution_trace (in place: every argument is validated before the trace is touched)
    if (!skip_interpolate) WF_TRY(wf_interpolate_columns(ctx, field, ext_degree, d_trace, num_cols, col_stride, log_n));
    // (for narrow traces committed with a byte hasher and no partitions the LDE transpose also produces the row hashes)
    const uint64_t N = 1ull << (log_n + log_blowup);
    const uint64_t rw = wf_row_width(num_cols, ext_degree);
    bool one_partition = num_partitions == 1;
    if (!one_partition) {   // PartitionOptions::partition_size (air/src/options.rs:428-437): one partition if it covers all columns
        const uint32_t min_ps = (hash_rate & 0xffu) / ext_degree;
        uint32_t ps = (num_cols + num_partitions - 1) / num_partitions;
        if (ps < min_ps) ps = min_ps;
        one_partition = ps == num_cols;   // row_matrix.rs:193 (ps > num_cols keeps the merge_many step)
    }
    int fused = 0;
    WF_TRY(wf_evaluate_polys_over_fused(ctx, field, ext_degree, d_trace, num_cols, col_stride, log_n, log_blowup, h_offset, d_lde,
                                        one_partition ? hash : -1, d_leaves, &fused));
    // compute_execution_trace_commitment
    if (!fused) WF_TRY(wf_hash_rows(ctx, hash, field, ext_degree, d_lde, N, rw, num_cols * ext_degree, num_partitions, hash_rate, d_leaves));
    WF_TRY(wf_merkle_build(ctx, hash, d_leaves, N, d_nodes));
    if (h_root) WF_TRY(wf_memcpy_d2h(ctx, h_root, (const uint8_t *)d_nodes + 32, 32));
    return WF_OK;
}
