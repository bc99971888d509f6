// corun_probe.hip -- round 4: what does each pass of the engine lose to a co-runner of ONE kind?
// A C2 step (one batch at a time, through the C ABI) is timed alone and beside a persistent side-stream kernel that does
// nothing but one thing: dependent VALU ops, conflict-free ds_read_b64, streaming global loads, scalar ALU ops, or s_sleep
// (waves that only occupy slots).  The co-runner is timed alone and beside the step as well (its iteration count is fixed,
// so its duration says how much of ITS resource the engine's passes took).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/corun_probe.hip -Lmakisu_amd -lmakisu_mi -o tools/bin/corun_probe
//   tools/bin/corun_probe [waves per SIMD of the co-runner: 1] [mask of co-runner kinds: 0x1F]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "makisu_mi.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define MIOK(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, mi_last_error(ctx)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_valu(uint32_t iters, uint32_t* out) {
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9E3779B9u;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {                       // one dependent chain, half alignbit half xor
            asm volatile("v_alignbit_b32 %0, %0, %0, 7\n\tv_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));
        }
    }
    if (x == 0x12345678u) out[0] = x;
}

__global__ __launch_bounds__(256) void k_lds(uint32_t iters, uint32_t* out) {
    __shared__ uint64_t tab[8192];                           // 64 KiB
    for (int i = threadIdx.x; i < 8192; i += 256) tab[i] = i;
    __syncthreads();
    // lane l reads entry (l % 32) of a 256-byte row: 32 lanes x 8 bytes = every bank once -- no conflicts
    uint32_t addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint64_t*)tab + (threadIdx.x & 31) * 8 + ((threadIdx.x >> 5) & 7) * 256;
    uint64_t a, b, c, d;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:2048\n\tds_read_b64 %2, %4 offset:4096\n\t"
                         "ds_read_b64 %3, %4 offset:6144\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
        }
    }
    if ((a ^ b ^ c ^ d) == 0x12345678u) out[0] = 1;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mem(uint32_t iters, const uint8_t* buf, uint64_t bytes, uint32_t* out) {
    // every wave streams its own window, coalesced: 64 lanes x 16 bytes = 1 KiB per load, 8 loads in flight
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint64_t n_waves = (uint64_t)gridDim.x * 4, window = bytes / n_waves & ~8191ull;
    const uint8_t* base = buf + wave * window + lane * 16;
    u32x4 acc = {0, 0, 0, 0};
    uint64_t off = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load((const u32x4*)(base + off + (uint64_t)k * 1024));
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
        off += 8192;
        if (off >= window) off = 0;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

__global__ __launch_bounds__(256) void k_salu(uint32_t iters, uint32_t* out) {
    uint32_t s = blockIdx.x;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 32; ++k) asm volatile("s_add_u32 %0, %0, 0x9E3779B9\n\ts_xor_b32 %0, %0, 0x85EBCA6B" : "+s"(s) : : "scc");
    }
    if (s == 0x12345678u) out[0] = s;
}

__global__ __launch_bounds__(256) void k_sleep(uint32_t iters, uint32_t* out) {
    for (uint32_t i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
    if (iters == 0xFFFFFFFFu) out[0] = 1;
}

struct Probe { const char* name; int kind; uint32_t iters; };

int main(int argc, char** argv) {
    const int per_simd = argc > 1 ? atoi(argv[1]) : 1;
    const unsigned kinds = argc > 2 ? (unsigned)strtoul(argv[2], nullptr, 0) : 0x1Fu;   // bit k: run co-runner kind k
    mi_ctx* ctx = nullptr;
    mi_config cfg;
    mi_config_default(&cfg);
    MIOK(mi_ctx_create(&cfg, &ctx));
    int32_t n_cu = 0, mhz = 0;
    uint64_t hbm = 0;
    char name[128];
    MIOK(mi_device_info(ctx, &n_cu, &mhz, &hbm, name, sizeof name));
    const uint64_t n_files = 100000;
    std::vector<uint64_t> sizes(n_files, 65536), ids(n_files);
    for (uint64_t i = 0; i < n_files; ++i) ids[i] = i;
    mi_batch* b = nullptr;
    MIOK(mi_batch_begin(ctx, n_files, n_files * 65536, &b));
    MIOK(mi_batch_add_synthetic(b, n_files, sizes.data(), ids.data(), 0x4D414B49));
    MIOK(mi_batch_run(b));
    hipStream_t side;
    HIPOK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    HIPOK(hipEventCreate(&e0));
    HIPOK(hipEventCreate(&e1));
    uint32_t* d_out = nullptr;
    uint8_t* d_buf = nullptr;
    const uint64_t buf_bytes = 4ull << 30;
    HIPOK(hipMalloc(&d_out, 64));
    HIPOK(hipMalloc(&d_buf, buf_bytes));
    HIPOK(hipMemset(d_buf, 1, buf_bytes));
    const dim3 grid((unsigned)(n_cu * per_simd)), block(256);   // 256 threads = one wave per SIMD and workgroup
    auto launch = [&](int kind, uint32_t iters) {
        switch (kind) {
            case 0: hipLaunchKernelGGL(k_valu, grid, block, 0, side, iters, d_out); break;
            case 1: hipLaunchKernelGGL(k_lds, grid, block, 0, side, iters, d_out); break;
            case 2: hipLaunchKernelGGL(k_mem, grid, block, 0, side, iters, (const uint8_t*)d_buf, buf_bytes, d_out); break;
            case 3: hipLaunchKernelGGL(k_salu, grid, block, 0, side, iters, d_out); break;
            default: hipLaunchKernelGGL(k_sleep, grid, block, 0, side, iters, d_out); break;
        }
    };
    auto time_alone = [&](int kind, uint32_t iters) {
        HIPOK(hipEventRecord(e0, side));
        launch(kind, iters);
        HIPOK(hipEventRecord(e1, side));
        HIPOK(hipEventSynchronize(e1));
        float ms = 0;
        HIPOK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms;
    };
    auto step = [&](mi_stats* st) {
        MIOK(mi_batch_submit(b));
        MIOK(mi_batch_wait(b));
        MIOK(mi_get_stats(ctx, st));
    };
    mi_stats st;
    double cdc0 = 0, sha0 = 0, tot0 = 0;
    for (int i = 0; i < 8; ++i) { step(&st); if (i >= 3) { cdc0 += st.ms_cdc / 5; sha0 += st.ms_sha_chunks / 5; tot0 += st.ms_total / 5; } }
    printf("%s, %d CUs; co-runner: %d wave(s) per SIMD (%u workgroups of 256)\n", name, n_cu, per_simd, grid.x);
    printf("step alone: marking %.3f ms, hashing %.3f ms, total %.3f ms\n", cdc0, sha0, tot0);
    Probe probes[] = {{"dependent VALU ops", 0, 2000}, {"ds_read_b64, no bank conflicts", 1, 2000}, {"streaming global loads", 2, 2000},
                      {"scalar ALU ops", 3, 2000}, {"s_sleep", 4, 2000}};
    for (Probe& p : probes) {
        if (!(kinds >> p.kind & 1u)) continue;
        // size the co-runner to ~30 ms alone: it must outlast three steps
        double t = time_alone(p.kind, p.iters);
        p.iters = (uint32_t)(p.iters * 30.0 / (t > 0.01 ? t : 0.01));
        if (p.iters < 16) p.iters = 16;
        const double alone = time_alone(p.kind, p.iters);
        HIPOK(hipEventRecord(e0, side));
        launch(p.kind, p.iters);
        HIPOK(hipEventRecord(e1, side));
        double cdc = 0, sha = 0, tot = 0;
        int n = 0;
        for (int i = 0; i < 3; ++i) {                          // three steps beside it (they end before it does)
            step(&st);
            cdc += st.ms_cdc; sha += st.ms_sha_chunks; tot += st.ms_total; ++n;
        }
        const bool covered = hipEventQuery(e1) == hipErrorNotReady;
        HIPOK(hipEventSynchronize(e1));
        float beside = 0;
        HIPOK(hipEventElapsedTime(&beside, e0, e1));
        printf("%-32s alone %6.2f ms, beside three steps %6.2f ms (+%.2f)%s | marking %.3f (x%.2f) hashing %.3f (x%.2f) total %.3f (x%.2f)\n",
               p.name, alone, beside, beside - alone, covered ? "" : " [ended before the steps did]", cdc / n, cdc / n / cdc0,
               sha / n, sha / n / sha0, tot / n, tot / n / tot0);
        fflush(stdout);
    }
    mi_batch_free(b);
    mi_ctx_destroy(ctx);
    return 0;
}
