/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see f64.h header note).
 *
 * BLAKE3 (default hash mode, 32-byte output), restated from the public BLAKE3 specification
 * (https://github.com/BLAKE3-team/BLAKE3-specs, section 2).  The reference obtains this arithmetic from the
 * un-vendored third-party crate `blake3` (crypto/Cargo.toml: `blake3 = { version = "1.8",
 * default-features = false }`, no lockfile => not pinned); its call sites are
 * crypto/src/hash/blake/mod.rs:29-65,131-151.  The reference's own tests hold NO known-answer vector for
 * it (crypto/src/hash/blake/tests.rs has only consistency checks), so this restatement is pinned instead
 * against (a) the published BLAKE3 test vectors and (b) the upstream C implementation that LLVM bundles
 * (llvm_blake3_hasher_*), see tests/test_oracle_blake3.py.
 */
#include <stdint.h>
#include <string.h>

static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                               0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t MSG_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};

enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };
#define CHUNK_LEN 1024
#define BLOCK_LEN 64

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static inline void g(uint32_t *s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx;
    s[d] = rotr(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];
    s[b] = rotr(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my;
    s[d] = rotr(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];
    s[b] = rotr(s[b] ^ s[c], 7);
}

static void round_fn(uint32_t *s, const uint32_t *m) {
    g(s, 0, 4, 8, 12, m[0], m[1]);
    g(s, 1, 5, 9, 13, m[2], m[3]);
    g(s, 2, 6, 10, 14, m[4], m[5]);
    g(s, 3, 7, 11, 15, m[6], m[7]);
    g(s, 0, 5, 10, 15, m[8], m[9]);
    g(s, 1, 6, 11, 12, m[10], m[11]);
    g(s, 2, 7, 8, 13, m[12], m[13]);
    g(s, 3, 4, 9, 14, m[14], m[15]);
}

/* full 16-word compression output */
static void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len,
                     uint32_t flags, uint32_t out[16]) {
    uint32_t s[16], m[16], t[16];
    for (int i = 0; i < 8; i++) s[i] = cv[i];
    for (int i = 0; i < 4; i++) s[8 + i] = IV[i];
    s[12] = (uint32_t)counter;
    s[13] = (uint32_t)(counter >> 32);
    s[14] = block_len;
    s[15] = flags;
    memcpy(m, block, sizeof m);
    for (int r = 0; r < 7; r++) {
        round_fn(s, m);
        if (r < 6) {
            for (int i = 0; i < 16; i++) t[i] = m[MSG_PERM[i]];
            memcpy(m, t, sizeof m);
        }
    }
    for (int i = 0; i < 8; i++) {
        out[i] = s[i] ^ s[i + 8];
        out[i + 8] = s[i + 8] ^ cv[i];
    }
}

static void load_block(const uint8_t *p, uint32_t len, uint32_t w[16]) {
    uint8_t buf[64];
    memset(buf, 0, 64);
    memcpy(buf, p, len);
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) | ((uint32_t)buf[4 * i + 2] << 16) |
               ((uint32_t)buf[4 * i + 3] << 24);
}

/* Pending final compression of a node (spec: "Output"). */
typedef struct {
    uint32_t cv[8];
    uint32_t block[16];
    uint64_t counter;
    uint32_t block_len;
    uint32_t flags;
} output_t;

/* Process one chunk (1..1024 bytes, or 0 bytes only for the empty message); leaves the last block pending. */
static void chunk_output(const uint8_t *p, uint32_t len, uint64_t chunk_counter, output_t *o) {
    uint32_t cv[8], out[16];
    memcpy(cv, IV, sizeof cv);
    uint32_t nblocks = len == 0 ? 1 : (len + BLOCK_LEN - 1) / BLOCK_LEN;
    for (uint32_t b = 0; b < nblocks; b++) {
        uint32_t off = b * BLOCK_LEN;
        uint32_t bl = (len - off) < BLOCK_LEN ? (len - off) : BLOCK_LEN;
        uint32_t flags = (b == 0 ? CHUNK_START : 0) | (b == nblocks - 1 ? CHUNK_END : 0);
        uint32_t w[16];
        load_block(p + off, bl, w);
        if (b == nblocks - 1) {
            memcpy(o->cv, cv, sizeof cv);
            memcpy(o->block, w, sizeof w);
            o->counter = chunk_counter;
            o->block_len = bl;
            o->flags = flags;
        } else {
            compress(cv, w, chunk_counter, bl, flags, out);
            memcpy(cv, out, sizeof cv);
        }
    }
}

static void output_cv(const output_t *o, uint32_t cv[8]) {
    uint32_t out[16];
    compress(o->cv, o->block, o->counter, o->block_len, o->flags, out);
    memcpy(cv, out, 32);
}

static void parent_output(const uint32_t l[8], const uint32_t r[8], output_t *o) {
    memcpy(o->cv, IV, 32);
    memcpy(o->block, l, 32);
    memcpy(o->block + 8, r, 32);
    o->counter = 0;
    o->block_len = BLOCK_LEN;
    o->flags = PARENT;
}

void or_blake3_hash(const uint8_t *in, uint64_t len, uint8_t out[32]) {
    uint32_t stack[54][8];
    int sp = 0;
    uint64_t nchunks = len == 0 ? 1 : (len + CHUNK_LEN - 1) / CHUNK_LEN;
    output_t o;
    for (uint64_t c = 0; c + 1 < nchunks; c++) {
        uint32_t cv[8];
        chunk_output(in + c * CHUNK_LEN, CHUNK_LEN, c, &o);
        output_cv(&o, cv);
        uint64_t total = c + 1;
        while ((total & 1) == 0) {
            output_t po;
            sp--;
            parent_output(stack[sp], cv, &po);
            output_cv(&po, cv);
            total >>= 1;
        }
        memcpy(stack[sp++], cv, 32);
    }
    uint64_t last = nchunks - 1;
    chunk_output(in + last * CHUNK_LEN, (uint32_t)(len - last * CHUNK_LEN), last, &o);
    while (sp > 0) {
        uint32_t cv[8];
        output_t po;
        output_cv(&o, cv);
        sp--;
        parent_output(stack[sp], cv, &po);
        o = po;
    }
    uint32_t w[16];
    compress(o.cv, o.block, 0 /* root output block counter */, o.block_len, o.flags | ROOT, w);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)w[i];
        out[4 * i + 1] = (uint8_t)(w[i] >> 8);
        out[4 * i + 2] = (uint8_t)(w[i] >> 16);
        out[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
}
