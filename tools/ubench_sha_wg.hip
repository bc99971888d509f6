// ubench_sha_wg.hip -- the SHA-256 compression alone (no memory traffic) by workgroup shape: do two waves of ONE
// workgroup on a SIMD, re-aligned by a barrier per block, share the SIMD better than free-running waves?
// (2-pass VALU ops reach their rate only when two waves issue them together: profiles/r03_ubench_mix*.txt)
// Measures the VALU roof of the batched SHA kernel: ns per 64-byte block per wave per SIMD, and the
// chip-wide hashing rate that implies, for a fully unrolled body vs a 16-round rolled body.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t u32;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ u32 rotr(u32 x, u32 n) { return __builtin_amdgcn_alignbit(x, x, n); }
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ u32 ch3(u32 e, u32 f, u32 g)  { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ u32 maj3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

__device__ constexpr u32 kK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
__constant__ u32 cK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

#define ROUND(a,b,c,d,e,f,g,h,kw) { const u32 t1 = (h + ch3(e,f,g) + (kw)) + xor3(rotr(e,6),rotr(e,11),rotr(e,25)); \
    const u32 t2 = xor3(rotr(a,2),rotr(a,13),rotr(a,22)) + maj3(a,b,c); d += t1; h = t1 + t2; }

template <int BAR>
__device__ __forceinline__ void compress_unrolled(u32 (&st)[8], u32 (&w)[16]) {
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (BAR > 0 && i % BAR == 0) __builtin_amdgcn_s_barrier();
        u32 wi;
        if (i < 16) wi = w[i];
        else {
            const u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            wi = w[i & 15] + xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3) + w[(i + 9) & 15] + xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
            w[i & 15] = wi;
        }
        const u32 t1 = (h + ch3(e, f, g) + (wi + kK[i])) + xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
        const u32 t2 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22)) + maj3(a, b, c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}


template <int WG, int BAR>
__global__ __launch_bounds__(WG) void sha_loop(u32* out, int blocks) {
    extern __shared__ unsigned char pad[];
    u32 st[8], w[16];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 8; ++i) st[i] = t * 0x9E3779B9u + i;
    u32 x = t * 0x85EBCA6Bu + 1;
    for (int b = 0; b < blocks; ++b) {
        for (int i = 0; i < 16; ++i) { x = x * 1664525u + 1013904223u; w[i] = x ^ st[i & 7]; }
        compress_unrolled<BAR>(st, w);
    }
    u32 r = 0;
    for (int i = 0; i < 8; ++i) r ^= st[i];
    out[t] = r;
    if (blocks == -1) pad[threadIdx.x] = 0;
}

struct Cfg { const char* name; void (*fn)(u32*, int); int wg; int g; };

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    u32* out;
    CHK(hipMalloc(&out, sizeof(u32) * 256 * ncu * 16));
    const Cfg cfgs[] = {
        {"wg256 free", sha_loop<256, 0>, 256, 2}, {"wg256 free", sha_loop<256, 0>, 256, 4}, {"wg256 free", sha_loop<256, 0>, 256, 6},
        {"wg512 bar/64", sha_loop<512, 64>, 512, 1}, {"wg512 bar/64", sha_loop<512, 64>, 512, 2}, {"wg512 bar/64", sha_loop<512, 64>, 512, 3},
        {"wg512 bar/32", sha_loop<512, 32>, 512, 1}, {"wg512 bar/32", sha_loop<512, 32>, 512, 2}, {"wg512 bar/32", sha_loop<512, 32>, 512, 3},
        {"wg512 bar/16", sha_loop<512, 16>, 512, 1}, {"wg512 bar/16", sha_loop<512, 16>, 512, 2}, {"wg512 bar/16", sha_loop<512, 16>, 512, 3},
        {"wg512 bar/8", sha_loop<512, 8>, 512, 1}, {"wg512 bar/8", sha_loop<512, 8>, 512, 2}, {"wg512 bar/8", sha_loop<512, 8>, 512, 3},
        {"wg512 bar/4", sha_loop<512, 4>, 512, 1}, {"wg512 bar/4", sha_loop<512, 4>, 512, 2}, {"wg512 bar/4", sha_loop<512, 4>, 512, 3},
        {"wg512 bar/2", sha_loop<512, 2>, 512, 1}, {"wg512 bar/2", sha_loop<512, 2>, 512, 2}, {"wg512 bar/2", sha_loop<512, 2>, 512, 3},
        {"wg1024 bar/16", sha_loop<1024, 16>, 1024, 1}, {"wg1024 bar/4", sha_loop<1024, 4>, 1024, 1},
    };
    for (auto& c : cfgs) CHK(hipFuncSetAttribute((const void*)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((sha_loop<256, 0>), dim3(ncu * 8), dim3(256), 0, 0, out, 512);
    CHK(hipDeviceSynchronize());
    const int reps = argc > 1 ? atoi(argv[1]) : 9;
    printf("# SHA-256 compression only; W = waves per SIMD; us per wave-block per SIMD over %d launches; TB/s = chip-wide at the median\n", reps);
    printf("%-12s %2s %2s | %7s %7s %7s | %6s\n", "shape", "G", "W", "min", "median", "max", "TB/s");
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (auto& c : cfgs) {
        const int W = c.g * c.wg / 256;
        const size_t lds = (160u * 1024u) / (size_t)(c.g + 1) + 1024u;
        const int blocks = 1024;
        double v[64];
        for (int r = 0; r < reps; ++r) {
            CHK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(c.fn, dim3(ncu * c.g), dim3(c.wg), lds, 0, out, blocks);
            CHK(hipEventRecord(b, 0));
            CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            v[r] = ms * 1e3 / blocks / W;
        }
        for (int i = 0; i < reps; ++i) for (int j = i + 1; j < reps; ++j) if (v[j] < v[i]) { double t = v[i]; v[i] = v[j]; v[j] = t; }
        const double med = v[reps / 2];
        printf("%-12s %2d %2d | %7.3f %7.3f %7.3f | %6.3f\n", c.name, c.g, W, v[0], med, v[reps - 1],
               (double)ncu * 4 * 64 * 64.0 / (med * 1e-6) / 1e12);
    }
    return 0;
}
