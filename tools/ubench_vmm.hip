// ubench_vmm.hip -- can the arena be a reserved address range that is MAPPED PIECE BY PIECE (hipMemAddressReserve /
// hipMemCreate / hipMemMap / hipMemSetAccess) instead of one hipMalloc that moves when it grows?
// Measures, on the box it runs on:
//   1. hipMalloc + hipFree of S bytes (what an arena costs today: on some boxes of the pool 47-68 ms per GiB);
//   2. the same S bytes as pieces of C bytes mapped one after the other into one reserved range (per piece: create, map,
//      set access), for several C;
//   3. host-to-device copies from pinned 8 MiB slabs into either kind of memory (GB/s), and the same copies WHILE another
//      thread maps further pieces behind them (does a page-table update stall copies in flight?);
//   4. a coalesced 16-byte-per-lane read kernel and a lane-owned-128-byte-line kernel (the Gear pattern) over either kind of
//      memory (a piecewise mapping may get smaller page-table fragments: TLB misses would show here).
// usage: ubench_vmm [GiB = 6]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
typedef uint32_t u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d: %s\n", hipGetErrorString(e_), __LINE__, #x); exit(1); } } while (0)

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(256) void coalesced(const u32x4* __restrict__ in, u64 n16, u32* out) {
    u32x4 acc = {0, 0, 0, 0};
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) acc ^= in[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
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

struct Vmm {
    void* va = nullptr;
    size_t reserved = 0, mapped = 0;
    std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> pieces;
    hipMemAllocationProp prop = {};
    void reserve(size_t bytes, size_t align) {
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        CHK(hipMemAddressReserve(&va, bytes, align, nullptr, 0));
        reserved = bytes;
    }
    void map_piece(size_t bytes) {
        hipMemGenericAllocationHandle_t h;
        CHK(hipMemCreate(&h, bytes, &prop, 0));
        CHK(hipMemMap((uint8_t*)va + mapped, bytes, 0, h, 0));
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CHK(hipMemSetAccess((uint8_t*)va + mapped, bytes, &acc, 1));
        pieces.push_back({h, bytes});
        mapped += bytes;
    }
    void release() {                                         // (the range itself stays reserved: see mi_alloc.hip on reusing one)
        size_t at = 0;
        for (auto& p : pieces) { CHK(hipMemUnmap((uint8_t*)va + at, p.second)); CHK(hipMemRelease(p.first)); at += p.second; }
        pieces.clear();
        mapped = 0;
    }
};

static double copy_rate(uint8_t* dev, size_t bytes, void** slabs, hipStream_t* streams, int n, size_t slab) {
    std::atomic<size_t> next{0};
    const double t0 = now_s();
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i)
        th.emplace_back([&, i] {
            CHK(hipSetDevice(0));
            for (;;) {
                const size_t at = next.fetch_add(slab);
                if (at >= bytes) break;
                const size_t len = bytes - at < slab ? bytes - at : slab;
                CHK(hipMemcpyAsync(dev + at, slabs[i], len, hipMemcpyHostToDevice, streams[i]));
                CHK(hipStreamSynchronize(streams[i]));
            }
        });
    for (auto& t : th) t.join();
    return bytes / (now_s() - t0) / 1e9;
}

int main(int argc, char** argv) {
    const size_t S = (size_t)(argc > 1 ? atof(argv[1]) : 6.0) << 30;
    CHK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    size_t fr = 0, tot = 0;
    CHK(hipMemGetInfo(&fr, &tot));
    printf("# %s, %d CUs, %.1f GiB free of %.1f; S = %.1f GiB\n", prop.gcnArchName, ncu, fr / 1073741824.0, tot / 1073741824.0, S / 1073741824.0);
    hipMemAllocationProp ap = {};
    ap.type = hipMemAllocationTypePinned;
    ap.location.type = hipMemLocationTypeDevice;
    size_t gmin = 0, grec = 0;
    CHK(hipMemGetAllocationGranularity(&gmin, &ap, hipMemAllocationGranularityMinimum));
    CHK(hipMemGetAllocationGranularity(&grec, &ap, hipMemAllocationGranularityRecommended));
    printf("# allocation granularity: minimum %zu, recommended %zu\n", gmin, grec);

    const int NT = 8;
    const size_t slab = 8u << 20;
    void* slabs[NT];
    hipStream_t streams[NT];
    for (int i = 0; i < NT; ++i) { CHK(hipHostMalloc(&slabs[i], slab, hipHostMallocDefault)); memset(slabs[i], 0x5a + i, slab); CHK(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking)); }
    u32* out;
    CHK(hipMalloc(&out, 64));

    auto kernels = [&](const char* what, uint8_t* buf) {
        for (int k = 0; k < 2; ++k) {
            auto launch = [&] {
                if (k == 0) hipLaunchKernelGGL(coalesced, dim3(ncu * 4), dim3(256), 0, 0, (const u32x4*)buf, S / 16, out);
                else hipLaunchKernelGGL(lane_run<1024>, dim3(ncu * 4), dim3(256), 0, 0, buf, S / 65536, out);
            };
            launch();
            CHK(hipDeviceSynchronize());
            hipEvent_t a, b;
            CHK(hipEventCreate(&a));
            CHK(hipEventCreate(&b));
            CHK(hipEventRecord(a, 0));
            for (int r = 0; r < 5; ++r) launch();
            CHK(hipEventRecord(b, 0));
            CHK(hipDeviceSynchronize());
            float ms;
            CHK(hipEventElapsedTime(&ms, a, b));
            printf("  %-34s %-22s %8.3f ms %8.1f GB/s\n", what, k == 0 ? "coalesced 16 B/lane" : "lane-owned 1 KiB runs", ms / 5, S / (ms / 5 * 1e-3) / 1e9);
        }
    };

    // 1. hipMalloc
    for (int rep = 0; rep < 2; ++rep) {
        uint8_t* p;
        double t0 = now_s();
        CHK(hipMalloc((void**)&p, S));
        const double t_alloc = now_s() - t0;
        const double r1 = copy_rate(p, S, slabs, streams, NT, slab);
        const double r2 = copy_rate(p, S, slabs, streams, NT, slab);
        printf("hipMalloc(%.1f GiB) %.3f s = %.1f ms/GiB; H2D first pass %.1f GB/s, second %.1f GB/s\n", S / 1073741824.0, t_alloc, t_alloc * 1e3 / (S / 1073741824.0), r1, r2);
        if (rep == 1) kernels("hipMalloc", p);
        t0 = now_s();
        CHK(hipFree(p));
        printf("  hipFree %.3f s\n", now_s() - t0);
    }
    // 2. pieces
    for (size_t C : {(size_t)2 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30}) {
        Vmm v;
        double t0 = now_s();
        v.reserve(S + (1ull << 30), 1ull << 30);
        const double t_res = now_s() - t0;
        t0 = now_s();
        double t_first = 0;
        while (v.mapped < S) { v.map_piece(C); if (!t_first) t_first = now_s() - t0; }
        const double t_map = now_s() - t0;
        const double r1 = copy_rate((uint8_t*)v.va, S, slabs, streams, NT, slab);
        const double r2 = copy_rate((uint8_t*)v.va, S, slabs, streams, NT, slab);
        printf("pieces of %4zu MiB: reserve %.4f s, map all %.3f s = %.1f ms/GiB (first piece %.2f ms); H2D first pass %.1f GB/s, second %.1f GB/s\n", C >> 20, t_res, t_map,
               t_map * 1e3 / (S / 1073741824.0), t_first * 1e3, r1, r2);
        char nm[64];
        snprintf(nm, sizeof nm, "pieces of %zu MiB", C >> 20);
        kernels(nm, (uint8_t*)v.va);
        t0 = now_s();
        v.release();
        printf("  unmap + release %.3f s\n", now_s() - t0);
    }
    // 3. copies while another thread maps ahead of them (the arena as it would be used: pieces of 256 MiB)
    for (size_t C : {(size_t)64 << 20, (size_t)256 << 20}) {
        Vmm v;
        v.reserve(S + (1ull << 30), 1ull << 30);
        std::atomic<size_t> mapped{0};
        const double t0 = now_s();
        std::thread mapper([&] { CHK(hipSetDevice(0)); while (v.mapped < S) { v.map_piece(C); mapped.store(v.mapped); } });
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        double t_first_landed = 0;
        for (int i = 0; i < NT; ++i)
            th.emplace_back([&, i] {
                CHK(hipSetDevice(0));
                for (;;) {
                    const size_t at = next.fetch_add(slab);
                    if (at >= S) break;
                    while (mapped.load() < at + slab) std::this_thread::yield();
                    CHK(hipMemcpyAsync((uint8_t*)v.va + at, slabs[i], slab, hipMemcpyHostToDevice, streams[i]));
                    CHK(hipStreamSynchronize(streams[i]));
                    if (at == 0) t_first_landed = now_s() - t0;
                }
            });
        for (auto& t : th) t.join();
        const double t_all = now_s() - t0;
        mapper.join();
        printf("map (pieces of %zu MiB) and copy at once: first slab landed after %.2f ms, all %.1f GiB after %.3f s = %.1f GB/s\n", C >> 20, t_first_landed * 1e3, S / 1073741824.0, t_all,
               S / t_all / 1e9);
        v.release();
    }
    return 0;
}
