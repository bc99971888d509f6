// ubench_valu64.hip -- issue rate of 64-bit VALU forms on gfx950 (candidates for rotate / Gear update).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef unsigned long long u64; typedef uint32_t u32;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define REP8(S) S S S S S S S S
#define K64(NAME, ASM)                                                                   \
__global__ __launch_bounds__(256) void NAME(u64* out, int iters) {                        \
    u64 a0 = threadIdx.x * 0x9E3779B97F4A7C15ull + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7; \
    u64 b = a0 * 11 + 99; u32 c = threadIdx.x | 1;                                         \
    for (int i = 0; i < iters; ++i) {                                                     \
        REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) \
    }                                                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;                       \
}
K64(lshr_b64,  "v_lshrrev_b64 %0, 7, %0\n\tv_lshrrev_b64 %1, 7, %1\n\tv_lshrrev_b64 %2, 7, %2\n\tv_lshrrev_b64 %3, 7, %3")
K64(lshl_b64,  "v_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3")
K64(lshl_add64,"v_lshl_add_u64 %0, %0, 1, %4\n\tv_lshl_add_u64 %1, %1, 1, %4\n\tv_lshl_add_u64 %2, %2, 1, %4\n\tv_lshl_add_u64 %3, %3, 1, %4")
K64(mov_b64,   "v_mov_b64 %0, %4\n\tv_mov_b64 %1, %4\n\tv_mov_b64 %2, %4\n\tv_mov_b64 %3, %4")
K64(pk_mov,    "v_pk_mov_b32 %0, %4, %4 op_sel:[0,1]\n\tv_pk_mov_b32 %1, %4, %4 op_sel:[0,1]\n\tv_pk_mov_b32 %2, %4, %4 op_sel:[0,1]\n\tv_pk_mov_b32 %3, %4, %4 op_sel:[0,1]")
K64(mad_u64,   "v_mad_u64_u32 %0, vcc, %5, %5, %0\n\tv_mad_u64_u32 %1, vcc, %5, %5, %1\n\tv_mad_u64_u32 %2, vcc, %5, %5, %2\n\tv_mad_u64_u32 %3, vcc, %5, %5, %3")
struct E { const char* n; void (*f)(u64*, int); int per; };
int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    u64* out; CHK(hipMalloc(&out, 8ull * 256 * ncu * 8));
    E es[] = {{"lshr_b64", lshr_b64, 32}, {"lshl_b64", lshl_b64, 32}, {"lshl_add64", lshl_add64, 32}, {"mov_b64", mov_b64, 32}, {"pk_mov", pk_mov, 32}, {"mad_u64u32", mad_u64, 32}};
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(lshl_b64, dim3(ncu * 8), dim3(256), 0, 0, out, 32768);
    CHK(hipDeviceSynchronize());
    printf("instr        ns per wave64 instr per SIMD at W=1, 2, 4, 8\n");
    for (auto& e : es) {
        printf("%-12s", e.n);
        for (int W = 1; W <= 8; W *= 2) {
            const int iters = 16384;
            hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
            CHK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(e.f, dim3(ncu * W), dim3(256), 0, 0, out, iters);
            CHK(hipEventRecord(b, 0)); CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            printf("  %6.2f", ms * 1e6 / ((double)iters * e.per * W));
        }
        printf("\n");
    }
    return 0;
}
