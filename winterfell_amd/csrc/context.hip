// Context, device memory helpers, scratch buffers and cached constant tables.
#include <stdlib.h>

#include "wf_internal.h"

#include <stdio.h>
#include <string.h>
#include <sys/types.h>

#include <mutex>
#include <vector>

extern "C" int wf_version(void) { return 100; }

extern "C" const char *wf_strerror(int status) {
    switch (status) {
        case WF_OK: return "ok";
        case WF_ERR_INVALID_ARG: return "invalid argument";
        case WF_ERR_NOT_POWER_OF_TWO: return "size must be a power of two";
        case WF_ERR_TOO_FEW_LEAVES: return "a Merkle tree needs at least two leaves";
        case WF_ERR_DOMAIN_TOO_LARGE: return "multiplicative subgroup of the requested size does not exist in the field";
        case WF_ERR_UNSUPPORTED: return "unsupported field / hash / extension combination";
        case WF_ERR_HIP: return "HIP runtime error";
        case WF_ERR_NO_DEVICE: return "no such HIP device";
        case WF_ERR_ZERO_OFFSET: return "domain offset cannot be zero";
        case WF_ERR_NOT_FOUND: return "nonce not found";
        case WF_ERR_COMM_ABORTED: return "a peer rank of the communicator failed; the collective was abandoned";
        case WF_ERR_DEVICE_STATUS: return "a kernel reported a protocol failure (device status word); results of the last calls are invalid";
        default: return "unknown status";
    }
}

extern "C" int wf_device_count(int *h_count) {
    if (!h_count) return WF_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *h_count = n;
    return WF_OK;
}

// ---- device allocations: plain, guarded (WF_DEBUG_GUARD=1) or red-zoned (WF_DEBUG_GUARD=2) ----------------------------------
// Guarded: every block is its own virtual-address reservation [unmapped granule][mapped pages][unmapped granule] with the block's
// LAST byte on the last mapped byte (WF_DEBUG_GUARD_ALIGN=left: its first byte on the first), so an access past the end (before the
// start) of ANY buffer the library or — through wf_debug_torch_malloc, which tests/conftest.py plugs into torch — the caller owns
// is a GPU page fault at the instruction that makes it, not a silent read of a neighbour.  The electric-fence run of the whole GPU
// suite is the memory-safety evidence DESIGN.md quotes.  Red-zoned: hipMalloc blocks with 4 KiB of 0xA5 on both sides, compared when
// the block is freed (writes only; works without the virtual-memory API).
namespace {
int guard_mode() {
    static const int m = [] {
        const char *e = getenv("WF_DEBUG_GUARD");
        return e ? atoi(e) : 0;
    }();
    return m;
}
bool guard_left() {
    static const bool l = [] {
        const char *e = getenv("WF_DEBUG_GUARD_ALIGN");
        return e && e[0] == 'l';
    }();
    return l;
}
bool guard_env_flag(const char *name) {
    const char *e = getenv(name);
    return e && e[0] && e[0] != '0';
}
struct GuardRec {
    void *va = nullptr;       // reservation (mode 1) or raw hipMalloc pointer (mode 2)
    size_t reserve = 0, mapped = 0, gran = 0, bytes = 0;
    hipMemGenericAllocationHandle_t handle{};
    int mode = 0;
};
std::mutex g_guard_mu;
std::map<void *, GuardRec> g_guard;
constexpr size_t RED = 4096;

size_t round_up(size_t v, size_t g) { return (v + g - 1) / g * g; }

hipError_t guard_alloc(int device, size_t bytes, void **out) {
    if (bytes == 0) bytes = 1;
    GuardRec r;
    r.bytes = bytes;
    if (guard_mode() == 1) {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
        if (e != hipSuccess) return e;
        if (gran < 4096) gran = 4096;
        r.gran = gran;
        r.mapped = round_up(bytes, gran);
        r.reserve = r.mapped + 2 * gran;
        if ((e = hipMemAddressReserve(&r.va, r.reserve, gran, nullptr, 0)) != hipSuccess) return e;
        if ((e = hipMemCreate(&r.handle, r.mapped, &prop, 0)) != hipSuccess) {
            (void)hipMemAddressFree(r.va, r.reserve);
            return e;
        }
        char *base = (char *)r.va + gran;
        if ((e = hipMemMap(base, r.mapped, 0, r.handle, 0)) != hipSuccess) {
            (void)hipMemRelease(r.handle);
            (void)hipMemAddressFree(r.va, r.reserve);
            return e;
        }
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if ((e = hipMemSetAccess(base, r.mapped, &acc, 1)) != hipSuccess) {
            (void)hipMemUnmap(base, r.mapped);
            (void)hipMemRelease(r.handle);
            (void)hipMemAddressFree(r.va, r.reserve);
            return e;
        }
        r.mode = 1;
        if (guard_env_flag("WF_DEBUG_GUARD_PREFILL")) {
            // experiment (tools/debug_guard192.py): fill the fresh mapping, wait, and read both ends back before handing it out
            unsigned char probe[2] = {0, 0};
            for (int attempt = 0; attempt < 8; attempt++) {
                if ((e = hipMemset(base, 0x5a, r.mapped)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) break;
                (void)hipMemcpy(&probe[0], base, 1, hipMemcpyDeviceToHost);
                (void)hipMemcpy(&probe[1], base + r.mapped - 1, 1, hipMemcpyDeviceToHost);
                if (probe[0] == 0x5a && probe[1] == 0x5a) break;
                fprintf(stderr, "wf guard: fresh mapping at %p lost its fill (attempt %d: %02x %02x)\n", (void *)base, attempt, probe[0], probe[1]);
            }
        }
        // 16-byte alignment is what the kernels assume of a caller's buffer (uint4 accesses); hipMalloc would give 256
        *out = guard_left() ? (void *)base : (void *)(base + r.mapped - round_up(bytes, 16));
    } else {
        char *raw = nullptr;
        hipError_t e = hipMalloc((void **)&raw, round_up(bytes, 256) + 2 * RED);
        if (e != hipSuccess) return e;
        if ((e = hipMemset(raw, 0xA5, RED)) != hipSuccess || (e = hipMemset(raw + RED + round_up(bytes, 256), 0xA5, RED)) != hipSuccess ||
            (e = hipDeviceSynchronize()) != hipSuccess) {
            (void)hipFree(raw);
            return e;
        }
        r.va = raw;
        r.mode = 2;
        *out = raw + RED;
    }
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard[*out] = r;
    return hipSuccess;
}

hipError_t guard_free(void *p) {
    GuardRec r;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guard.find(p);
        if (it == g_guard.end()) return hipFree(p);       // not ours: a block from before the mode was switched on
        r = it->second;
        g_guard.erase(it);
    }
    hipError_t e = hipDeviceSynchronize();                // what hipFree does implicitly
    if (r.mode == 1) {
        char *base = (char *)r.va + r.gran;
        hipError_t e2 = hipMemUnmap(base, r.mapped);
        hipError_t e3 = hipMemRelease(r.handle);
        // The reservation is KEPT: an address range is never mapped to a second allocation.  With ROCm 7.0 on gfx950 a range that is
        // unmapped, released, freed, reserved again (the runtime hands out the same address) and mapped to a NEW allocation keeps
        // serving the GPU stale translations: a fill + hipDeviceSynchronize + read-back of the fresh mapping returns the old
        // contents, kernels' stores to its first pages are lost (tools/repro_vmem_remap.hip reproduces it without this library;
        // the first electric-fence session of round 4 failed 7 tests that way and none with the reservations kept).
        // WF_DEBUG_GUARD_VA_REUSE=1 restores the free for that experiment.  A session leaks address space only (no memory).
        hipError_t e4 = guard_env_flag("WF_DEBUG_GUARD_VA_REUSE") ? hipMemAddressFree(r.va, r.reserve) : hipSuccess;
        return e != hipSuccess ? e : (e2 != hipSuccess ? e2 : (e3 != hipSuccess ? e3 : e4));
    }
    std::vector<unsigned char> z(2 * RED);
    char *raw = (char *)r.va;
    (void)hipMemcpy(z.data(), raw, RED, hipMemcpyDeviceToHost);
    (void)hipMemcpy(z.data() + RED, raw + RED + round_up(r.bytes, 256), RED, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < 2 * RED; i++) {
        if (z[i] != 0xA5) {
            fprintf(stderr, "wf guard: red zone of the %zu-byte block at %p overwritten (%s the block, zone offset %zu)\n", r.bytes, p,
                    i < RED ? "before" : "after", i % RED);
            fflush(stderr);
            abort();
        }
    }
    hipError_t e5 = hipFree(raw);
    return e != hipSuccess ? e : e5;
}
}  // namespace

int wf_dev_malloc(wf_ctx *ctx, void **d_ptr, size_t bytes) {
    if (guard_mode()) WF_HIP(guard_alloc(ctx->device, bytes, d_ptr));
    else WF_HIP(hipMalloc(d_ptr, bytes));
    return WF_OK;
}

int wf_dev_free(wf_ctx *ctx, void *d_ptr) {
    if (!d_ptr) return WF_OK;
    if (guard_mode()) WF_HIP(guard_free(d_ptr));
    else WF_HIP(hipFree(d_ptr));
    return WF_OK;
}

// torch.cuda.memory.CUDAPluggableAllocator entry points (tests/conftest.py, WF_DEBUG_GUARD set): every torch tensor of the test
// session becomes a guarded block too.  The pluggable allocator frees as soon as the tensor dies — no stream-ordered cache — so the
// free waits for the device first.
extern "C" void *wf_debug_torch_malloc(ssize_t size, int device, void *stream) {
    (void)stream;
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    if (guard_mode() == 0) return hipMalloc(&p, size > 0 ? (size_t)size : 1) == hipSuccess ? p : nullptr;
    return guard_alloc(device, (size_t)(size > 0 ? size : 1), &p) == hipSuccess ? p : nullptr;
}
extern "C" void wf_debug_torch_free(void *ptr, ssize_t size, int device, void *stream) {
    (void)size;
    (void)stream;
    if (!ptr) return;
    (void)hipSetDevice(device);
    if (guard_mode() == 0) (void)hipFree(ptr);
    else (void)guard_free(ptr);
}
extern "C" int wf_debug_guard_mode(void) { return guard_mode(); }

// ---- host <-> device copies ----------------------------------------------------------------------------------------------------------
// PAGEABLE caller memory is never handed to the runtime.  ROCclr pins a pageable range of 128 KiB .. 32 MiB in place for the copy
// and keeps the pinned object in a per-stream cache of eight, found again by (address, size) alone.  A host buffer that is freed
// (munmap / heap trim) and allocated again at the same address within those eight copies is found in the cache, but the kernel
// driver invalidated the registration when the range was unmapped and leaves it invalid ("It will fail later with a VM fault if the
// GPU tries to access it", amdgpu_amdkfd_gpuvm.c): the next copy is a GPU page fault at a HOST address and HSA aborts the process.
// tools/repro_pinned_cache.py shows it with nothing but torch; DESIGN.md section 9.  So: a range the caller page-locked
// (wf_host_register, hipHostMalloc) is copied directly, everything else goes through two page-locked bounce buffers the context owns.
namespace {
constexpr size_t BOUNCE_BYTES = 4u << 20;

bool host_byte_is_pinned(const void *p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}
// the WHOLE range [p, p + bytes) must be page-locked.  Walk the registrations that cover it: hipMemGetAddressRange on the device view of a
// page-locked host byte gives the base and size of ITS registration (hipHostRegister / hipHostMalloc), so the next byte to look at
// is the first one behind it — two adjacent registrations are both valid for the copy engine, a hole between them is found whatever
// its size (registrations are 4 KiB-granular; round 5 sampled every 2 MiB and could step over a hole).  When the runtime cannot name
// the registration's extent the range is treated as pageable (bounce buffers): slower, never wrong.
bool host_range_is_pinned(const void *p, size_t bytes) {
    const char *c = (const char *)p, *end = c + (bytes ? bytes : 1);
    for (int guard = 0; c < end && guard < 4096; guard++) {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, c) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (a.type != hipMemoryTypeHost || !a.devicePointer) return false;
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)a.devicePointer) != hipSuccess || !size) {
            (void)hipGetLastError();
            return false;
        }
        // offset of `c` inside its registration, measured in the device view (host and device views of one registration are both linear)
        const size_t off = (size_t)((const char *)a.devicePointer - (const char *)base);
        if (off >= size) return false;
        c += size - off;
    }
    return c >= end;
}

int ensure_bounce(wf_ctx *ctx) {
    if (ctx->h_bounce[0]) return WF_OK;
    for (int i = 0; i < 2; i++) {
        WF_HIP(hipHostMalloc(&ctx->h_bounce[i], BOUNCE_BYTES, hipHostMallocDefault));
        WF_HIP(hipEventCreateWithFlags(&ctx->bounce_ev[i], hipEventDisableTiming));
    }
    return WF_OK;
}
}  // namespace

int wf_copy_h2d(wf_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (bytes == 0) return WF_OK;
    if (host_range_is_pinned(h_src, bytes)) {
        WF_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        WF_HIP(hipStreamSynchronize(ctx->stream));
        return WF_OK;
    }
    WF_TRY(ensure_bounce(ctx));
    size_t off = 0;
    for (int c = 0; off < bytes; c++) {
        const int i = c & 1;
        const size_t len = bytes - off < BOUNCE_BYTES ? bytes - off : BOUNCE_BYTES;
        if (c >= 2) WF_HIP(hipEventSynchronize(ctx->bounce_ev[i]));
        memcpy(ctx->h_bounce[i], (const char *)h_src + off, len);
        WF_HIP(hipMemcpyAsync((char *)d_dst + off, ctx->h_bounce[i], len, hipMemcpyHostToDevice, ctx->stream));
        WF_HIP(hipEventRecord(ctx->bounce_ev[i], ctx->stream));
        off += len;
    }
    WF_HIP(hipStreamSynchronize(ctx->stream));
    return WF_OK;
}

int wf_copy_h2d_small_async(wf_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (bytes == 0) return WF_OK;
    if (bytes > wf_ctx::STAGE_BYTES) return wf_copy_h2d(ctx, d_dst, h_src, bytes);
    const uint32_t i = ctx->stage_next;
    ctx->stage_next = (i + 1) % wf_ctx::STAGE_SLOTS;
    if (!ctx->h_stage[i]) {
        // the slot is published only when both the buffer and its event exist
        void *buf = nullptr;
        hipEvent_t ev = nullptr;
        WF_HIP(hipHostMalloc(&buf, wf_ctx::STAGE_BYTES, hipHostMallocDefault));
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostFree(buf);
            return WF_ERR_HIP;
        }
        ctx->h_stage[i] = buf;
        ctx->stage_ev[i] = ev;
    }
    if (ctx->stage_busy[i]) WF_HIP(hipEventSynchronize(ctx->stage_ev[i]));   // the copy that last read this slot (eight uploads ago)
    memcpy(ctx->h_stage[i], h_src, bytes);
    WF_HIP(hipMemcpyAsync(d_dst, ctx->h_stage[i], bytes, hipMemcpyHostToDevice, ctx->stream));
    WF_HIP(hipEventRecord(ctx->stage_ev[i], ctx->stream));
    ctx->stage_busy[i] = true;
    return WF_OK;
}

int wf_copy_d2h(wf_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (bytes == 0) {
        WF_HIP(hipStreamSynchronize(ctx->stream));
        return WF_OK;
    }
    if (host_range_is_pinned(h_dst, bytes)) {
        WF_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        WF_HIP(hipStreamSynchronize(ctx->stream));
        return WF_OK;
    }
    WF_TRY(ensure_bounce(ctx));
    size_t off = 0, prev_off = 0, prev_len = 0;
    for (int c = 0; off < bytes; c++) {
        const int i = c & 1;
        const size_t len = bytes - off < BOUNCE_BYTES ? bytes - off : BOUNCE_BYTES;
        WF_HIP(hipMemcpyAsync(ctx->h_bounce[i], (const char *)d_src + off, len, hipMemcpyDeviceToHost, ctx->stream));
        WF_HIP(hipEventRecord(ctx->bounce_ev[i], ctx->stream));
        if (c >= 1) {                                   // drain the previous chunk while this one is in flight
            WF_HIP(hipEventSynchronize(ctx->bounce_ev[i ^ 1]));
            memcpy((char *)h_dst + prev_off, ctx->h_bounce[i ^ 1], prev_len);
        }
        prev_off = off;
        prev_len = len;
        off += len;
    }
    WF_HIP(hipStreamSynchronize(ctx->stream));
    const int last = (int)(((bytes + BOUNCE_BYTES - 1) / BOUNCE_BYTES - 1) & 1);
    memcpy((char *)h_dst + prev_off, ctx->h_bounce[last], prev_len);
    return WF_OK;
}

int wf_check_status(wf_ctx *ctx) {
    if (!ctx->h_status) return WF_OK;
    const uint32_t st = __atomic_exchange_n(ctx->h_status, 0u, __ATOMIC_ACQ_REL);
    if (st == 0) return WF_OK;
    ctx->last_device_status = st;
    if (st & WF_STATUS_MERKLE_TICKET) {            // put the ticket ring back into its initial state: the next tree starts clean
        memset(ctx->tree_ticket_epoch, 0, sizeof(ctx->tree_ticket_epoch));
        (void)hipMemsetAsync(ctx->d_tree_ticket, 0, WF_TREE_TICKETS * sizeof(uint32_t), ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
    }
    return WF_ERR_DEVICE_STATUS;
}

static int ctx_create(int device_id, bool own_stream, void *hip_stream, wf_ctx **out) {
    if (!out) return WF_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return WF_ERR_NO_DEVICE;
    wf_ctx *ctx = new wf_ctx();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) {
        delete ctx;
        return WF_ERR_HIP;
    }
    if (own_stream) {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return WF_ERR_HIP;
        }
    } else {
        ctx->stream = (hipStream_t)hip_stream;
    }
    ctx->own_stream = own_stream;
    if (const char *e = getenv("WF_NTT_PREFETCH")) ctx->ntt_prefetch = e[0] == '1';
    if (const char *e = getenv("WF_NTT_COSET_ORDER")) ctx->coset_order = e[0] != '0';
    if (const char *e = getenv("WF_ROWS_HASH_WIDE")) ctx->rows_hash_wide = e[0] != '0';
    if (const char *e = getenv("WF_LDE_VT")) ctx->lde_vt = e[0] != '0';
    if (const char *e = getenv("WF_VT_PREFETCH")) ctx->vt_prefetch = e[0] == '1';
    if (const char *e = getenv("WF_NTT_BT")) ctx->ntt_bt = e[0] == '0' ? 0 : (e[0] == '1' ? 1 : -1);
    if (const char *e = getenv("WF_NTT_F64_TABLES")) ctx->f64_tw_tables = e[0] == '0' ? 0 : (e[0] == '1' ? 1 : -1);
    if (const char *e = getenv("WF_NTT_BIG")) ctx->ntt_big = e[0] == '0' ? 0 : (e[0] == '1' ? 1 : (e[0] == '2' ? 2 : -1));
    // WF_NTT_PLAN="L:r0,r1,...": a pass plan for transforms of 2^L points (tools/time_batch_ntt.py measures alternatives with it).
    // Read ONCE, here: a stray variable cannot change the pass shapes of a running host process from one call to the next.
    if (const char *e = getenv("WF_NTT_PLAN")) {
        unsigned l = 0, r[6] = {0, 0, 0, 0, 0, 0};
        const int got = sscanf(e, "%u:%u,%u,%u,%u,%u,%u", &l, &r[0], &r[1], &r[2], &r[3], &r[4], &r[5]);
        uint32_t sum = 0, k = 0;
        bool ok = got >= 2 && l >= 1 && l <= 32;
        for (; ok && k < 6 && r[k]; k++) {
            if (r[k] > 8) ok = false;               // no kernel is wider than radix 256
            sum += r[k];
        }
        if (ok && sum == l) {
            ctx->plan_log_n = l;
            ctx->plan_npass = k;
            for (uint32_t q = 0; q < 6; q++) ctx->plan_log_r[q] = r[q];
        }
    }
    // the device status word lives in page-locked host memory the device can write (fine-grained): reading it after a stream
    // synchronisation costs the host nothing
    if (hipHostMalloc((void **)&ctx->h_status, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&ctx->d_status, ctx->h_status, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (ctx->h_status) (void)hipHostFree(ctx->h_status);
        delete ctx;
        return WF_ERR_HIP;
    }
    *ctx->h_status = 0;
    // merkle_finish_kernel's ticket ring: allocated and zeroed here, on the creation stream, and waited for — no allocation and no
    // synchronisation on the launch path (round-3 advice), and a later wf_ctx_set_stream cannot find the fill still queued
    if (wf_dev_malloc(ctx, &ctx->d_tree_ticket, WF_TREE_TICKETS * sizeof(uint32_t)) != WF_OK ||
        hipMemsetAsync(ctx->d_tree_ticket, 0, WF_TREE_TICKETS * sizeof(uint32_t), ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)wf_ctx_destroy(ctx);
        return WF_ERR_HIP;
    }
    ctx->owned.push_back(ctx->d_tree_ticket);
    *out = ctx;
    return WF_OK;
}

extern "C" int wf_ctx_create(int device_id, wf_ctx **out) { return ctx_create(device_id, true, nullptr, out); }
extern "C" int wf_ctx_create_on_stream(int device_id, void *hip_stream, wf_ctx **out) { return ctx_create(device_id, false, hip_stream, out); }

extern "C" int wf_ctx_destroy(wf_ctx *ctx) {
    if (!ctx) return WF_ERR_INVALID_ARG;
    {
    // (no WF_ENTER here: its guard would outlive `delete ctx` below and unlock a mutex in freed memory)
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);      // a call still running on another thread finishes first
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (void *p : ctx->owned) (void)wf_dev_free(ctx, p);
    for (auto &kv : ctx->pool_free) (void)wf_dev_free(ctx, kv.second);
    for (auto &kv : ctx->pool_live) (void)wf_dev_free(ctx, kv.first);      // blocks the caller never returned
    for (int i = 0; i < 3; i++)
        if (ctx->scratch[i]) (void)wf_dev_free(ctx, ctx->scratch[i]);
    for (int i = 0; i < 2; i++) {
        if (ctx->h_bounce[i]) (void)hipHostFree(ctx->h_bounce[i]);
        if (ctx->bounce_ev[i]) (void)hipEventDestroy(ctx->bounce_ev[i]);
    }
    for (int i = 0; i < wf_ctx::STAGE_SLOTS; i++) {
        if (ctx->h_stage[i]) (void)hipHostFree(ctx->h_stage[i]);
        if (ctx->stage_ev[i]) (void)hipEventDestroy(ctx->stage_ev[i]);
    }
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    for (auto &r : ctx->prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    if (ctx->span_a) {
        (void)hipEventDestroy(ctx->span_a);
        (void)hipEventDestroy(ctx->span_b);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
    return WF_OK;
}

extern "C" int wf_ctx_set_stream(wf_ctx *ctx, void *hip_stream) {
    WF_ENTER(ctx);
    if ((hipStream_t)hip_stream == ctx->stream) return WF_OK;
    (void)hipStreamSynchronize(ctx->stream);       // cached pool blocks are ordered on the old stream
    if (ctx->own_stream) {
        (void)hipStreamDestroy(ctx->stream);
        ctx->own_stream = false;
    }
    ctx->stream = (hipStream_t)hip_stream;
    return WF_OK;
}

extern "C" int wf_ctx_get_stream(wf_ctx *ctx, void **hip_stream) {
    WF_ENTER(ctx);
    if (!hip_stream) return WF_ERR_INVALID_ARG;
    *hip_stream = (void *)ctx->stream;
    return WF_OK;
}

extern "C" int wf_ctx_sync(wf_ctx *ctx) {
    WF_ENTER(ctx);
    WF_HIP(hipStreamSynchronize(ctx->stream));
    return wf_check_status(ctx);
}

extern "C" int wf_last_hip_error(wf_ctx *ctx) { return ctx ? ctx->last_hip_error : 0; }
extern "C" uint32_t wf_last_device_status(wf_ctx *ctx) { return ctx ? ctx->last_device_status : 0; }

// size classes: multiples of 512 B below 1 MiB, of 2 MiB above (what hipMalloc rounds to anyway)
static size_t pool_round(size_t bytes) {
    if (bytes == 0) bytes = 1;
    const size_t g = bytes < (1u << 20) ? 512 : (2u << 20);
    return (bytes + g - 1) / g * g;
}

static void pool_release_cached(wf_ctx *ctx) {
    if (ctx->pool_free.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->pool_free) (void)wf_dev_free(ctx, kv.second);
    ctx->pool_free.clear();
    ctx->pool_free_bytes = 0;
}

extern "C" int wf_malloc(wf_ctx *ctx, size_t bytes, void **d_ptr) {
    WF_ENTER(ctx);
    if (!d_ptr) return WF_ERR_INVALID_ARG;
    if (guard_mode()) {
        // electric-fence sessions: no size classes and no caching — the block is exactly `bytes` long, so its last byte sits on the
        // last mapped byte (pool_round would leave up to 2 MiB - 1 mapped bytes behind it), and it is unmapped when it is released
        WF_HIP(hipSetDevice(ctx->device));
        WF_TRY(wf_dev_malloc(ctx, d_ptr, bytes ? bytes : 1));
        ctx->pool_live[*d_ptr] = bytes ? bytes : 1;
        return WF_OK;
    }
    const size_t want = pool_round(bytes);
    // best fit among the cached blocks, wasting at most a quarter of the block
    auto it = ctx->pool_free.lower_bound(want);
    if (it != ctx->pool_free.end() && it->first <= want + want / 4) {
        *d_ptr = it->second;
        ctx->pool_live[it->second] = it->first;
        ctx->pool_free_bytes -= it->first;
        ctx->pool_free.erase(it);
        return WF_OK;
    }
    WF_HIP(hipSetDevice(ctx->device));
    if (wf_dev_malloc(ctx, d_ptr, want) != WF_OK) {   // out of memory: give the cached blocks back to the driver and retry once
        (void)hipGetLastError();
        pool_release_cached(ctx);
        WF_TRY(wf_dev_malloc(ctx, d_ptr, want));
    }
    ctx->pool_live[*d_ptr] = want;
    return WF_OK;
}

extern "C" int wf_free(wf_ctx *ctx, void *d_ptr) {
    WF_ENTER(ctx);
    if (!d_ptr) return WF_OK;
    auto it = ctx->pool_live.find(d_ptr);
    // not a live block of this context's pool — another context's, a double free, a pointer into a block: refused, never passed on
    // to hipFree (which would pull memory from under whoever owns it)
    if (it == ctx->pool_live.end()) return WF_ERR_INVALID_ARG;
    const size_t sz = it->second;
    ctx->pool_live.erase(it);
    if (guard_mode()) {
        WF_HIP(hipStreamSynchronize(ctx->stream));       // queued kernels may still use the block
        return wf_dev_free(ctx, d_ptr);
    }
    ctx->pool_free.emplace(sz, d_ptr);
    ctx->pool_free_bytes += sz;
    // keep at most 64 GiB cached (of 288): beyond that return the largest blocks to the driver
    while (ctx->pool_free_bytes > (64ull << 30) && !ctx->pool_free.empty()) {
        auto big = std::prev(ctx->pool_free.end());
        WF_HIP(hipStreamSynchronize(ctx->stream));
        WF_TRY(wf_dev_free(ctx, big->second));
        ctx->pool_free_bytes -= big->first;
        ctx->pool_free.erase(big);
    }
    return WF_OK;
}

extern "C" int wf_ctx_trim(wf_ctx *ctx) {
    WF_ENTER(ctx);
    pool_release_cached(ctx);
    return WF_OK;
}

extern "C" int wf_memcpy_h2d(wf_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    WF_ENTER(ctx);
    if (bytes && (!d_dst || !h_src)) return WF_ERR_INVALID_ARG;
    return wf_copy_h2d(ctx, d_dst, h_src, bytes);     // returns after the copy: the caller may free h_src
}

extern "C" int wf_memcpy_d2h(wf_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    WF_ENTER(ctx);
    if (bytes && (!h_dst || !d_src)) return WF_ERR_INVALID_ARG;
    WF_TRY(wf_copy_d2h(ctx, h_dst, d_src, bytes));
    return wf_check_status(ctx);
}

extern "C" int wf_memcpy_d2d(wf_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
    WF_ENTER(ctx);
    if (bytes && (!d_dst || !d_src)) return WF_ERR_INVALID_ARG;
    WF_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return WF_OK;
}

extern "C" int wf_host_register(wf_ctx *ctx, void *h_ptr, size_t bytes) {
    WF_ENTER(ctx);
    if (!h_ptr || bytes == 0) return WF_ERR_INVALID_ARG;
    WF_HIP(hipHostRegister(h_ptr, bytes, hipHostRegisterDefault));
    return WF_OK;
}

extern "C" int wf_host_unregister(wf_ctx *ctx, void *h_ptr) {
    WF_ENTER(ctx);
    if (!h_ptr) return WF_ERR_INVALID_ARG;
    WF_HIP(hipStreamSynchronize(ctx->stream));      // no copy of this context may still be reading or writing the range
    WF_HIP(hipHostUnregister(h_ptr));
    return WF_OK;
}

// test hook: overwrite the ticket word the NEXT one-launch Merkle tree will use (tests/test_gpu_commit.py checks that the kernel
// reports a word it does not expect instead of silently leaving the top of the tree unwritten)
extern "C" int wf_debug_poke_tree_ticket(wf_ctx *ctx, uint32_t value) {
    WF_ENTER(ctx);
    uint32_t *tk = (uint32_t *)ctx->d_tree_ticket + (ctx->tree_ticket_next % WF_TREE_TICKETS);
    WF_HIP(hipMemcpyAsync(tk, &value, sizeof(value), hipMemcpyHostToDevice, ctx->stream));
    WF_HIP(hipStreamSynchronize(ctx->stream));
    return WF_OK;
}

// ---- measured shader clock ------------------------------------------------------------------------------
// One wavefront on a stream of its own spins for `spin_us` of the constant-rate clock (s_memrealtime) and counts shader cycles
// (s_memtime) meanwhile: the clock the chip actually runs at WHILE whatever the caller queued on the context's stream is executing
// (the part clocks to its power budget; bench.py reports this next to its issue ceilings instead of assuming 2.4 GHz).
namespace {
__global__ void clock_probe_kernel(uint64_t spin_ticks, uint64_t *out) {
    const uint64_t r0 = wall_clock64(), c0 = clock64();
    uint64_t r1 = r0;
    while (r1 - r0 < spin_ticks) r1 = wall_clock64();
    out[0] = r1 - r0;
    out[1] = clock64() - c0;
}
}  // namespace

extern "C" int wf_debug_shader_clock(wf_ctx *ctx, uint32_t spin_us, double *h_mhz) {
    WF_ENTER(ctx);
    if (!h_mhz || spin_us == 0 || spin_us > 100000) return WF_ERR_INVALID_ARG;
    int khz = 0;
    WF_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    if (khz <= 0) return WF_ERR_UNSUPPORTED;
    uint64_t *h = nullptr;
    WF_HIP(hipHostMalloc((void **)&h, 64, hipHostMallocMapped));
    h[0] = h[1] = 0;
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s, (uint64_t)spin_us * (uint64_t)khz / 1000, h);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
    }
    const double ticks = (double)h[0], cycles = (double)h[1];
    (void)hipHostFree(h);
    WF_HIP(e);
    if (ticks <= 0) return WF_ERR_HIP;
    *h_mhz = cycles / (ticks / ((double)khz * 1e3)) / 1e6;
    return WF_OK;
}

// ---- profiling hook ---------------------------------------------------------------------------------
extern "C" int wf_prof_enable(wf_ctx *ctx, int on) {
    WF_ENTER(ctx);
    ctx->prof_enabled = on == 1;
    ctx->prof_span = on == 2;
    ctx->span_open = false;
    ctx->span_launches = 0;
    return WF_OK;
}

// Synchronises, then writes one line per kernel name: "name count total_ms\n"; clears the records.
extern "C" int wf_prof_collect(wf_ctx *ctx, char *h_buf, size_t buf_len) {
    WF_ENTER(ctx);
    if (!h_buf || buf_len == 0) return WF_ERR_INVALID_ARG;
    WF_HIP(hipStreamSynchronize(ctx->stream));
    std::map<std::string, std::pair<uint64_t, double>> acc;
    for (auto &r : ctx->prof) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto &e = acc[r.name];
            e.first++;
            e.second += ms;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    ctx->prof.clear();
    size_t off = 0;
    h_buf[0] = 0;
    if (ctx->prof_span && ctx->span_open) {           // span mode: one line, the name no kernel has
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->span_a, ctx->span_b) == hipSuccess) {
            int w = snprintf(h_buf, buf_len, "__span__ %llu %.6f\n", (unsigned long long)ctx->span_launches, ms);
            if (w > 0 && (size_t)w < buf_len) off = (size_t)w;
        }
        ctx->span_open = false;
        ctx->span_launches = 0;
    }
    for (auto &kv : acc) {
        int w = snprintf(h_buf + off, buf_len - off, "%s %llu %.6f\n", kv.first.c_str(), (unsigned long long)kv.second.first,
                         kv.second.second);
        if (w < 0 || (size_t)w >= buf_len - off) break;
        off += (size_t)w;
    }
    return WF_OK;
}

// ---- persistent launches -----------------------------------------------------------------------------
// CUs x (256-thread workgroups of `kernel` the occupancy calculator admits per CU), cached per kernel.  The grid of a
// persistent kernel: every workgroup is resident from the start and walks its share of the tiles.
int wf_resident_blocks(wf_ctx *ctx, const void *kernel, uint32_t *out) {
    auto it = ctx->resident_blocks.find(kernel);
    if (it == ctx->resident_blocks.end()) {
        int cus = 0, per_cu = 0;
        WF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
        WF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0));
        it = ctx->resident_blocks.emplace(kernel, (uint32_t)(cus > 0 && per_cu > 0 ? cus * per_cu : 0)).first;
    }
    *out = it->second;
    return WF_OK;
}

// ---- scratch ---------------------------------------------------------------------------------------
int wf_scratch(wf_ctx *ctx, int slot, size_t bytes, void **out) {
    // electric-fence sessions: the slot is re-allocated at the exact size of every request (grow-only slots would leave a request
    // below the high-water mark with mapped memory behind it); the contents of a slot never outlive the library call that asked
    if (ctx->scratch_bytes[slot] < bytes || (guard_mode() && ctx->scratch_bytes[slot] != bytes)) {
        if (ctx->scratch[slot]) {
            WF_HIP(hipStreamSynchronize(ctx->stream));
            WF_TRY(wf_dev_free(ctx, ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_bytes[slot] = 0;
        }
        WF_TRY(wf_dev_malloc(ctx, &ctx->scratch[slot], bytes));
        ctx->scratch_bytes[slot] = bytes;
    }
    *out = ctx->scratch[slot];
    return WF_OK;
}

