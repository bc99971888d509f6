// vmm_va_probe.hip -- how many address ranges of an arena's size can ONE process reserve?  csrc/mi_arena.hip never gives a range
// back (a mapping placed at an address that has just been unmapped ends in GPU faults on this ROCm: mi_alloc.hip), so the answer
// bounds how many walk-fed batches a process can create before its batches fall back to one allocation each (arena_reserve).
// usage: vmm_va_probe [GiB per range = 32] [map: 1 = put one 32 MiB piece into every range, touch it with a memset, unmap it]
//        LD_PRELOAD=<torch>/lib/libamdhip64.so:<torch>/lib/libhsa-runtime64.so vmm_va_probe    on the runtime PyTorch bundles
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? (size_t)atol(argv[1]) : 32;
    const bool map = argc > 2 && atoi(argv[2]) == 1;
    int ver = 0;
    (void)hipRuntimeGetVersion(&ver);
    if (hipSetDevice(0) != hipSuccess) { printf("no device\n"); return 1; }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t piece = 32u << 20;
    const double t0 = now_s();
    size_t n = 0;
    hipError_t e = hipSuccess;
    void* first = nullptr;
    void* last = nullptr;
    for (; n < 200000; ++n) {
        void* va = nullptr;
        e = hipMemAddressReserve(&va, gib << 30, 1ull << 30, nullptr, 0);
        if (e != hipSuccess) break;
        if (!first) first = va;
        last = va;
        if (map) {
            hipMemGenericAllocationHandle_t h;
            if ((e = hipMemCreate(&h, piece, &prop, 0)) != hipSuccess) { printf("range %zu: hipMemCreate: %s\n", n, hipGetErrorString(e)); break; }
            if ((e = hipMemMap(va, piece, 0, h, 0)) != hipSuccess) { printf("range %zu: hipMemMap: %s\n", n, hipGetErrorString(e)); break; }
            if ((e = hipMemSetAccess(va, piece, &acc, 1)) != hipSuccess) { printf("range %zu: hipMemSetAccess: %s\n", n, hipGetErrorString(e)); break; }
            if ((e = hipMemset(va, 0x5a, piece)) != hipSuccess) { printf("range %zu: hipMemset: %s\n", n, hipGetErrorString(e)); break; }
            (void)hipMemUnmap(va, piece);
            (void)hipMemRelease(h);
        }
        if (n == 99 || n == 999 || n == 9999) printf("  %zu ranges after %.2f s\n", n + 1, now_s() - t0);
    }
    printf("HIP runtime %d: %zu ranges of %zu GiB reserved (%s) in %.2f s; first %p, last %p; stopped by: %s\n", ver, n, gib,
           map ? "each mapped, touched, unmapped once" : "addresses only", now_s() - t0, first, last, e == hipSuccess ? "the probe's own limit" : hipGetErrorString(e));
    return 0;
}
