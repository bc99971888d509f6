// ubench_stream.hip -- read bandwidth of the access patterns the kernels use, on gfx950.
//  coalesced : lane l reads 16 B at base + 16*l (+1 KiB per step)            -- the textbook stream
//  lane_run  : lane l owns a contiguous run of R bytes and reads it 16 B at a time, 8 loads
//              (one 128 B line) back to back -- the Gear kernel's pattern
//  lane_64   : lane l reads 64 B (4 x 16 B) per step from its own stream at random 64 KiB-apart
//              places -- the SHA kernel's pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t u32; typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void coalesced(const u32x4* __restrict__ in, u64 n16, u32* out) {
    u32x4 acc = {0, 0, 0, 0};
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) acc ^= in[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// each wave owns a 64*R-byte tile; lane l reads [l*R, (l+1)*R) in 128 B pieces
template <int R>
__global__ __launch_bounds__(256) void lane_run(const uint8_t* __restrict__ in, u64 n_tiles, u32* out) {
    const int lane = threadIdx.x & 63;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    u32x4 acc = {0, 0, 0, 0};
    for (u64 t = wave; t < n_tiles; t += nw) {
        const uint8_t* p = in + t * (64ull * R) + (u64)lane * R;
        for (int pc = 0; pc < R / 128; ++pc) {
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *(const u32x4*)(p + pc * 128 + 16 * i);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= v[i];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const u64 bytes = 6ull << 30;
    uint8_t* buf; u32* out;
    CHK(hipMalloc(&buf, bytes + 4096)); CHK(hipMalloc(&out, 64));
    CHK(hipMemset(buf, 0x5a, bytes));
    auto time = [&](const char* name, auto launch) {
        launch(); CHK(hipDeviceSynchronize());
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        CHK(hipEventRecord(a, 0)); for (int r = 0; r < 5; ++r) launch(); CHK(hipEventRecord(b, 0)); CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        printf("%-28s %8.3f ms  %7.1f GB/s (nominal 6 GiB)\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e9);
    };
    for (int wpc : {4, 8, 12, 16}) {
        char nm[64];
        const int grid = ncu * wpc / 4;
        snprintf(nm, sizeof nm, "coalesced  %2d waves/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL(coalesced, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, bytes / 16, out); });
        snprintf(nm, sizeof nm, "lane_run1K %2d waves/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL(lane_run<1024>, dim3(grid), dim3(256), 0, 0, buf, bytes / 65536, out); });
        snprintf(nm, sizeof nm, "lane_run256 %2d waves/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL(lane_run<256>, dim3(grid), dim3(256), 0, 0, buf, bytes / 16384, out); });
        snprintf(nm, sizeof nm, "lane_run384 %2d waves/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL(lane_run<384>, dim3(grid), dim3(256), 0, 0, buf, bytes / (64 * 384), out); });
        snprintf(nm, sizeof nm, "lane_run640 %2d waves/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL(lane_run<640>, dim3(grid), dim3(256), 0, 0, buf, bytes / (64 * 640), out); });
        snprintf(nm, sizeof nm, "lane_run896 %2d waves/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL(lane_run<896>, dim3(grid), dim3(256), 0, 0, buf, bytes / (64 * 896), out); });
        snprintf(nm, sizeof nm, "lane_run1152 %2d waves/CU", wpc);
        time(nm, [&] { hipLaunchKernelGGL(lane_run<1152>, dim3(grid), dim3(256), 0, 0, buf, bytes / (64 * 1152), out); });
    }
    return 0;
}
