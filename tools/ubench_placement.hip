// ubench_placement.hip -- where does the dispatcher put the workgroups of a persistent grid?
// 2 x n_cu workgroups of 256 threads with the SHA chunk pass's footprint (136 VGPRs: a CU has room for
// THREE), all co-resident for ~1 ms; every workgroup records the XCC / SE / CU it runs on.  Prints, per
// launch, how many CUs hold 0, 1, 2, 3 workgroups -- without and with the 54 KiB LDS request that
// makes two per CU the only possible placement.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_placement.hip -o tools/bin/ubench_placement && tools/bin/ubench_placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <map>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void where_kernel(uint32_t* out, uint64_t spin_ticks) {
    asm volatile("; footprint of sha256_items_kernel<0,false>" ::: "v135");
    extern __shared__ uint8_t pad[];
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));      // HW_REG_HW_ID
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));    // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (pad && spin_ticks == 1) pad[threadIdx.x] = 0;                       // keeps the LDS request alive
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, grid = 2 * ncu;
    uint32_t* d;
    CHK(hipMalloc(&d, grid * 8));
    std::vector<uint32_t> h(grid * 2);
    CHK(hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    int wall_khz = 100000;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    const uint64_t ticks = (uint64_t)wall_khz;                              // 1 ms
    for (size_t lds : {(size_t)0, (size_t)(54 * 1024 + 1024)}) {
        printf("## %d workgroups x 256 threads, 136 VGPRs, dynamic LDS %zu B\n", grid, lds);
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(where_kernel, dim3(grid), dim3(256), lds, 0, d, ticks);
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost));
            std::map<uint32_t, int> per_cu;
            std::map<uint32_t, int> per_xcc;
            for (int b = 0; b < grid; ++b) {
                const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
                per_cu[(xcc << 16) | ((hw >> 8) & 0xFF)] += 1;                // CU_ID 11:8, SH_ID 12, SE_ID 15:13
                per_xcc[xcc] += 1;
            }
            int hist[8] = {0};
            for (auto& kv : per_cu) hist[kv.second < 7 ? kv.second : 7]++;
            printf("launch %d: CUs seen %zu of %d; CUs holding 1/2/3/4+ workgroups: %d / %d / %d / %d; per XCC:", r,
                   per_cu.size(), ncu, hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
            for (auto& kv : per_xcc) printf(" %d", kv.second);
            printf("\n");
        }
    }
    return 0;
}
