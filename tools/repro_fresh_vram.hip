// repro_fresh_vram.hip -- the staging hypothesis of DESIGN.md 4.5 tested OUTSIDE the library (VERDICT r3, weak #4a):
// "a host-to-device copy into VRAM that hipMalloc has just handed out can lose bytes, because the driver is still
// wiping that memory on the SDMA engines the copy uses".  Per round: allocate CHURN GB, write them, free them, allocate
// FRESH GB and AT ONCE copy a known pattern into all of it from 8 host threads (8 MiB pinned slabs, one stream each,
// as mi_stage.hip's readers do), then check every word on the device.  Prints one line per bad round and a summary.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/repro_fresh_vram tools/repro_fresh_vram.hip -lpthread
//   tools/bin/repro_fresh_vram [churn_GB=110] [rounds=200] [fresh_GB=4] [max_seconds=400]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

__host__ __device__ static inline uint64_t word_at(uint64_t idx, uint64_t round) {   // splitmix64 of (word index, round)
    uint64_t z = idx * 0x9E3779B97F4A7C15ull + round * 0xD1B54A32D192ED03ull + 1;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// out[0] = wrong words, out[1] = of them zero, out[2] = lowest wrong word index + 1, out[3] = highest + 1
__global__ void check_kernel(const uint64_t* p, uint64_t n_words, uint64_t round, unsigned long long* out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = p[i];
        if (v != word_at(i, round)) {
            atomicAdd(&out[0], 1ull);
            if (v == 0) atomicAdd(&out[1], 1ull);
            atomicMin(&out[2], (unsigned long long)i + 1);
            atomicMax(&out[3], (unsigned long long)i + 1);
        }
    }
}

int main(int argc, char** argv) {
    const double churn_gb = argc > 1 ? atof(argv[1]) : 110, fresh_gb = argc > 3 ? atof(argv[3]) : 4;
    int rounds = argc > 2 ? atoi(argv[2]) : 200;
    const int kThreads = 8;
    const double max_s = argc > 4 ? atof(argv[4]) : 400;
    const uint64_t slab = 8ull << 20, fresh = (uint64_t)(fresh_gb * (1ull << 30)) / slab * slab, n_slabs = fresh / slab;
    const uint64_t churn = (uint64_t)(churn_gb * (1ull << 30));
    CK(hipSetDevice(0));
    std::vector<hipStream_t> streams(kThreads);
    std::vector<uint64_t*> slabs(kThreads);
    for (int t = 0; t < kThreads; ++t) {
        CK(hipStreamCreateWithFlags(&streams[t], hipStreamNonBlocking));
        CK(hipHostMalloc((void**)&slabs[t], slab, hipHostMallocDefault));
    }
    unsigned long long *d_out, h_out[4], bad_rounds = 0, bad_words = 0;
    CK(hipMalloc((void**)&d_out, 32));
    const auto t_start = std::chrono::steady_clock::now();
    double copy_s = 0, churn_s = 0;
    for (int r = 0; r < rounds; ++r) {
        void* big = nullptr;
        const auto tc = std::chrono::steady_clock::now();
        if (std::chrono::duration<double>(tc - t_start).count() > max_s) { rounds = r; break; }   // the time budget
        if (churn) {                                            // what a freed resident batch leaves behind
            CK(hipMalloc(&big, churn));
            CK(hipMemsetAsync(big, 0x5A, churn, 0));
            CK(hipDeviceSynchronize());
            CK(hipFree(big));
        }
        churn_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
        uint64_t* dev = nullptr;
        CK(hipMalloc((void**)&dev, fresh));                     // fresh VRAM: copies start the moment it exists
        std::atomic<uint64_t> next{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < kThreads; ++t)
            th.emplace_back([&, t] {
                CK(hipSetDevice(0));
                for (uint64_t s; (s = next.fetch_add(1)) < n_slabs;) {
                    const uint64_t w0 = s * (slab / 8);
                    for (uint64_t i = 0; i < slab / 8; ++i) slabs[t][i] = word_at(w0 + i, (uint64_t)r);
                    CK(hipMemcpyAsync((char*)dev + s * slab, slabs[t], slab, hipMemcpyHostToDevice, streams[t]));
                    CK(hipStreamSynchronize(streams[t]));
                }
            });
        for (auto& x : th) x.join();
        copy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        h_out[0] = h_out[1] = h_out[3] = 0; h_out[2] = ~0ull;
        CK(hipMemcpy(d_out, h_out, 32, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, dev, fresh / 8, (uint64_t)r, d_out);
        CK(hipMemcpy(h_out, d_out, 32, hipMemcpyDeviceToHost));
        if (h_out[0]) {
            ++bad_rounds; bad_words += h_out[0];
            printf("round %d: %llu wrong words (%llu zero) in words [%llu, %llu] of %llu\n", r, h_out[0], h_out[1],
                   h_out[2] - 1, h_out[3] - 1, (unsigned long long)(fresh / 8));
            fflush(stdout);
        }
        if (r % 20 == 19) { printf("... %d rounds, %llu bad so far\n", r + 1, bad_rounds); fflush(stdout); }
        CK(hipFree(dev));
    }
    const double total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    printf("repro_fresh_vram: %d rounds, churn %.0f GB written and freed per round, %.1f GB copied per round into a fresh "
           "allocation by %d threads (8 MiB slabs) at %.1f GB/s: %llu bad round(s), %llu wrong word(s); %.0f s, of which "
           "%.0f s allocating, writing and freeing the churn\n", rounds,
           churn_gb, fresh / 1e9, kThreads, fresh * (double)rounds / copy_s / 1e9, bad_rounds, bad_words, total_s, churn_s);
    return bad_rounds ? 1 : 0;
}
