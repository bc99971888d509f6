// vmm_repro.hip -- which sequences of piece sizes can be mapped one after the other into ONE reserved address range?
// Round 6: under the HIP runtime PyTorch bundles (7.0.2; `import torch` before the library is loaded makes it the process's
// runtime) hipMemMap / hipMemSetAccess refused some pieces with "invalid argument" that the system's runtime (7.2.0) maps.
// usage: vmm_repro [hint]      hint: every piece gets a reservation of its own at the address the last one ended at
//        LD_PRELOAD=<torch>/lib/libamdhip64.so:<torch>/lib/libhsa-runtime64.so vmm_repro   runs it on the bundled runtime
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
int main(int argc, char** argv) {
    const bool hint = argc > 1 && !strcmp(argv[1], "hint");
    int ver = 0;
    (void)hipRuntimeGetVersion(&ver);
    printf("# HIP runtime %d%s\n", ver, hint ? ", one reservation per piece at the previous one's end" : "");
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    void* host; hipStream_t s;
    if (hipHostMalloc(&host, 8u << 20, hipHostMallocDefault) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 1;
    const std::vector<std::vector<size_t>> seqs = {
        {4, 2}, {2, 2, 2, 2, 2, 2, 2, 2, 2, 2}, {64, 64, 64, 64, 256}, {64, 64, 64, 64, 64, 64}, {256, 256, 256}, {2, 2, 4, 8, 16, 32, 64, 128, 256, 512},
        {2, 4}, {4, 4, 2}, {6, 2}, {2, 6}, {64, 128}, {128, 64}, {32, 32, 64}, {1024, 1024}};
    for (const auto& seq : seqs) {
        void* va = nullptr;
        size_t at = 0;
        std::string line;
        hipError_t e = hipSuccess;
        if (!hint) e = hipMemAddressReserve(&va, 8ull << 30, 2u << 20, nullptr, 0);
        for (size_t mb : seq) {
            const size_t n = mb << 20;
            char w[96];
            const char* call = "reserve";
            void* p = hint ? nullptr : (uint8_t*)va + at;
            if (e == hipSuccess && hint) {
                e = hipMemAddressReserve(&p, n, 2u << 20, va ? (uint8_t*)va + at : nullptr, 0);
                if (e == hipSuccess && !va) va = p;
                if (e == hipSuccess && p != (uint8_t*)va + at) { snprintf(w, sizeof w, " %zu@%zu:ELSEWHERE(%p)", mb, at >> 20, p); line += w; break; }
            }
            hipMemGenericAllocationHandle_t h;
            if (e == hipSuccess) { call = "create"; e = hipMemCreate(&h, n, &prop, 0); }
            if (e == hipSuccess) { call = "map"; e = hipMemMap(p, n, 0, h, 0); }
            if (e == hipSuccess) { call = "access"; e = hipMemSetAccess(p, n, &acc, 1); }
            if (e == hipSuccess) { call = "copy"; e = hipMemcpyAsync((uint8_t*)p + n - 4096, host, 4096, hipMemcpyHostToDevice, s); if (e == hipSuccess) e = hipStreamSynchronize(s); }
            if (e == hipSuccess && at) { call = "copy across the seam"; e = hipMemcpyAsync((uint8_t*)p - 4096, host, 8192, hipMemcpyHostToDevice, s); if (e == hipSuccess) e = hipStreamSynchronize(s); }
            snprintf(w, sizeof w, " %zu@%zu:%s%s%s", mb, at >> 20, e == hipSuccess ? "ok" : call, e == hipSuccess ? "" : " -> ", e == hipSuccess ? "" : hipGetErrorString(e));
            line += w;
            if (e != hipSuccess) break;
            at += n;
        }
        printf("%s\n", line.c_str());
    }
    return 0;
}
