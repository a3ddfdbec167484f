// The hasher policies (one struct per Hasher of the reference's crypto crate) that the hashing, Merkle, FRI-row and coin
// kernels are written against, with the small digest load/store helpers and the run-time dispatch over them.  Included by
// every translation unit that instantiates kernels per hasher (hash_kernels.hip, merkle.hip, fri_rows.hip, coin.hip): they
// are split so that the library builds in parallel.
#pragma once
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
    static constexpr bool QUAD_MERGE = true;   // thin Merkle levels: one merge per four lanes (b3::quad_hash_block)
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
    static constexpr bool QUAD_MERGE = false;  // 48-byte blocks: not worth a second message layout
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
    static constexpr bool QUAD_MERGE = false;
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
    static constexpr bool QUAD_MERGE = false;
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
    static constexpr bool QUAD_MERGE = false;
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
    static constexpr bool QUAD_MERGE = false;
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

// one Hasher::merge per lane (the single-level Merkle launch of the Rescue hashers, wf_hash_merge_batch)
template <class H>
__global__ __launch_bounds__(256) void merge_batch_kernel(const void *pairs, uint64_t count, void *out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    uint32_t m[16], d[8];
    load_pair(pairs, gid, m);
    H::merge(m, d);
    store_digest(out, gid, d);
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
