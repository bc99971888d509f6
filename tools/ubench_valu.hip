// ubench_valu.hip -- VALU instruction issue-rate microbenchmark for gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip ; run on the GPU box.
// For each instruction: cycles per wave64 instruction per SIMD, with 8 independent
// chains (throughput) and 1 chain (dependent latency), at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP8(S) S S S S S S S S
#define KERNEL(NAME, ASM_INDEP, ASM_DEP)                                                     \
__global__ __launch_bounds__(256) void NAME##_ind(uint32_t* out, int iters, long long* cyc) { \
    uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3,            \
             a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;          \
    uint32_t b = a0 * 0x9E3779B9u + 12345u, c = a0 * 0x85EBCA6Bu + 999u;                      \
    long long t0 = clock64();                                                                \
    for (int i = 0; i < iters; ++i) {                                                         \
        REP8(asm volatile(ASM_INDEP : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4),      \
                          "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)                    \
    }                                                                                         \
    long long t1 = clock64();                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                          \
}                                                                                             \
__global__ __launch_bounds__(256) void NAME##_dep(uint32_t* out, int iters, long long* cyc) { \
    uint32_t a0 = threadIdx.x;                                                                \
    uint32_t b = a0 * 0x9E3779B9u + 12345u, c = a0 * 0x85EBCA6Bu + 999u;                      \
    long long t0 = clock64();                                                                \
    for (int i = 0; i < iters; ++i) {                                                         \
        REP8(asm volatile(ASM_DEP : "+v"(a0) : "v"(b), "v"(c));)                              \
    }                                                                                         \
    long long t1 = clock64();                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;                                          \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                          \
}

#define I8(op, tail) op " %0, %0, " tail "\n\t" op " %1, %1, " tail "\n\t" op " %2, %2, " tail "\n\t" op " %3, %3, " tail "\n\t" \
                     op " %4, %4, " tail "\n\t" op " %5, %5, " tail "\n\t" op " %6, %6, " tail "\n\t" op " %7, %7, " tail
#define D8(op, tail) op " %0, %0, " tail "\n\t" op " %0, %0, " tail "\n\t" op " %0, %0, " tail "\n\t" op " %0, %0, " tail "\n\t" \
                     op " %0, %0, " tail "\n\t" op " %0, %0, " tail "\n\t" op " %0, %0, " tail "\n\t" op " %0, %0, " tail

// independent variants reference %8 (b) and %9 (c); dependent ones %1 and %2
KERNEL(add_u32,   I8("v_add_u32_e32", "%8"),           D8("v_add_u32_e32", "%1"))
KERNEL(xor_b32,   I8("v_xor_b32_e32", "%8"),           D8("v_xor_b32_e32", "%1"))
KERNEL(lshr_b32,  I8("v_lshrrev_b32_e32", "%8")  ,     D8("v_lshrrev_b32_e32", "%1"))
KERNEL(alignbit,  I8("v_alignbit_b32", "%8, 7"),       D8("v_alignbit_b32", "%1, 7"))
KERNEL(alignself, "v_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %1, %1, %1, 7\n\tv_alignbit_b32 %2, %2, %2, 7\n\tv_alignbit_b32 %3, %3, %3, 7\n\tv_alignbit_b32 %4, %4, %4, 7\n\tv_alignbit_b32 %5, %5, %5, 7\n\tv_alignbit_b32 %6, %6, %6, 7\n\tv_alignbit_b32 %7, %7, %7, 7",
                  "v_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %0, %0, %0, 7\n\tv_alignbit_b32 %0, %0, %0, 7")
KERNEL(bitop3,    I8("v_bitop3_b32", "%8, %9 bitop3:0x96"), D8("v_bitop3_b32", "%1, %2 bitop3:0x96"))
KERNEL(add3,      I8("v_add3_u32", "%8, %9"),          D8("v_add3_u32", "%1, %2"))
KERNEL(xad,       I8("v_xad_u32", "%8, %9"),           D8("v_xad_u32", "%1, %2"))
KERNEL(bfi,       I8("v_bfi_b32", "%8, %9"),           D8("v_bfi_b32", "%1, %2"))
KERNEL(perm,      I8("v_perm_b32", "%8, %9"),          D8("v_perm_b32", "%1, %2"))
KERNEL(lshl_add,  I8("v_lshl_add_u32", "1, %8"),       D8("v_lshl_add_u32", "1, %1"))
KERNEL(and_or,    I8("v_and_or_b32", "%8, %9"),        D8("v_and_or_b32", "%1, %2"))
KERNEL(min3,      I8("v_min3_u32", "%8, %9"),          D8("v_min3_u32", "%1, %2"))
KERNEL(xor_e64,   I8("v_xor_b32_e64", "%8"),           D8("v_xor_b32_e64", "%1"))
KERNEL(mul_lo,    I8("v_mul_lo_u32", "%8"),            D8("v_mul_lo_u32", "%1"))
KERNEL(mad_u24,   I8("v_mad_u32_u24", "%8, %9"),       D8("v_mad_u32_u24", "%1, %2"))
KERNEL(bfe,       I8("v_bfe_u32", "8, 8"),             D8("v_bfe_u32", "8, 8"))
KERNEL(pk_add16,  I8("v_pk_add_u16", "%8"),            D8("v_pk_add_u16", "%1"))

struct Entry { const char* name; void (*ind)(uint32_t*, int, long long*); void (*dep)(uint32_t*, int, long long*); };
#define E(n) {#n, n##_ind, n##_dep}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d MHz\n", prop.gcnArchName, ncu, prop.clockRate / 1000);
    Entry es[] = {E(add_u32), E(xor_b32), E(lshr_b32), E(xor_e64), E(alignbit), E(alignself), E(bitop3), E(add3), E(xad),
                  E(bfi), E(perm), E(lshl_add), E(and_or), E(min3), E(bfe), E(mul_lo), E(mad_u24), E(pk_add16)};
    uint32_t* out; long long* cyc;
    CHK(hipMalloc(&out, sizeof(uint32_t) * 256 * ncu * 8));
    CHK(hipMalloc(&cyc, sizeof(long long) * ncu * 8));
    std::vector<long long> h(ncu * 8);
    const int iters = 32768;         // x 64 instructions per iteration
    // ramp the clocks: ~0.3 s of saturated VALU before measuring
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(add_u32_ind, dim3(ncu * 8), dim3(256), 0, 0, out, iters, cyc);
    CHK(hipDeviceSynchronize());
    printf("%-10s | WALL ns per wave64 instruction per SIMD (W waves share it); last col: T lane-ops/s at W=8 | clock64 ticks/instr at W=8\n", "instr");
    printf("%-10s | ind W=1  W=2  W=4  W=8 | dep W=1  W=2  W=4  W=8 |\n", "");
    for (auto& e : es) {
        double res[2][4]; double gops = 0, ticks = 0;
        for (int v = 0; v < 2; ++v) for (int wi = 0; wi < 4; ++wi) {
            const int W = 1 << wi;
            auto fn = v == 0 ? e.ind : e.dep;
            hipLaunchKernelGGL(fn, dim3(ncu * W), dim3(256), 0, 0, out, 16, cyc);   // warm
            CHK(hipDeviceSynchronize());
            hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
            CHK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(fn, dim3(ncu * W), dim3(256), 0, 0, out, iters, cyc);
            CHK(hipEventRecord(b, 0));
            CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            CHK(hipMemcpy(h.data(), cyc, sizeof(long long) * ncu * W, hipMemcpyDeviceToHost));
            double avg = 0; for (int i = 0; i < ncu * W; ++i) avg += (double)h[i]; avg /= ncu * W;
            const double n_inst = (double)iters * 64.0;              // per wave
            res[v][wi] = (double)ms * 1e6 / (n_inst * W);            // wall ns per instr per SIMD (W waves share it)
            if (v == 0 && wi == 3) { gops = (double)ncu * W * 4 * 64 * n_inst / (ms * 1e-3) / 1e12; ticks = avg / (n_inst * W); }
        }
        printf("%-10s |   %5.2f %5.2f %5.2f %5.2f |   %5.2f %5.2f %5.2f %5.2f | %6.1f T | %5.2f\n", e.name,
               res[0][0], res[0][1], res[0][2], res[0][3], res[1][0], res[1][1], res[1][2], res[1][3], gops, ticks);
    }
    return 0;
}
