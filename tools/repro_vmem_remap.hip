// Stand-alone reproducer (no library code) for the defect the round-4 electric-fence session ran into:
//   hipMemAddressReserve -> hipMemCreate -> hipMemMap -> hipMemSetAccess -> use -> hipMemUnmap -> hipMemRelease -> hipMemAddressFree,
// then the same again: the runtime hands out the SAME virtual address, now backed by a new allocation — and the GPU may keep
// using the translation of the old one (a fill of the new mapping reads back as the old contents / a kernel's stores are lost).
//   hipcc --offload-arch=gfx950 tools/repro_vmem_remap.hip -o tools/repro_vmem_remap.bin && tools/repro_vmem_remap.bin [rounds] [bytes]
// Prints one line per round that misbehaves and a summary; exit status 1 if any round did.  `keep` as a third argument keeps the
// reservations (never frees the address range): the mitigation csrc/context.hip's guard allocator uses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

__global__ void stamp(unsigned *p, size_t n, unsigned v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v + (unsigned)i;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    size_t bytes = argc > 2 ? (size_t)atoll(argv[2]) : 32000;
    const bool keep = argc > 3 && !strcmp(argv[3], "keep");
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (gran < 4096) gran = 4096;
    const size_t mapped = (bytes + gran - 1) / gran * gran, reserve = mapped + 2 * gran, words = mapped / 4;
    std::vector<unsigned> host(words);
    int bad_rounds = 0, same_va = 0;
    void *last = nullptr;
    for (int r = 0; r < rounds; r++) {
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, reserve, gran, nullptr, 0));
        same_va += va == last;
        last = va;
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, mapped, &prop, 0));
        char *base = (char *)va + gran;
        CK(hipMemMap(base, mapped, 0, h, 0));
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(base, mapped, &acc, 1));
        const unsigned v = 0x01000000u * (unsigned)(r + 1);
        hipLaunchKernelGGL(stamp, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, 0, (unsigned *)base, words, v);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(host.data(), base, mapped, hipMemcpyDeviceToHost));
        size_t wrong = 0, first = 0;
        for (size_t i = 0; i < words; i++)
            if (host[i] != v + (unsigned)i) {
                if (!wrong) first = i;
                wrong++;
            }
        if (wrong) {
            bad_rounds++;
            if (bad_rounds <= 10)
                printf("round %d at %p: %zu of %zu words are not what the kernel stored (first at word %zu: %08x, expected %08x)\n", r, (void *)base, wrong,
                       words, first, host[first], v + (unsigned)first);
        }
        CK(hipMemUnmap(base, mapped));
        CK(hipMemRelease(h));
        if (!keep) CK(hipMemAddressFree(va, reserve));
    }
    printf("%d rounds of %zu bytes (granularity %zu, reservations %s): %d misbehaved, %d got the previous round's address\n", rounds, mapped, gran,
           keep ? "kept" : "freed", bad_rounds, same_va);
    return bad_rounds ? 1 : 0;
}
