// mi_hip_stub.cpp -- a TEST DOUBLE for the HIP runtime: the thirty hip* entry points libmakisu_mi.so imports, implemented
// on host memory and host threads, so that the library's HOST side -- reader threads, pinned slabs, the inline window,
// arena growth, the order of copies, events and waits, the API's state machine -- runs on a box without a GPU, and runs
// under ThreadSanitizer.  Test infrastructure only (LD_PRELOAD=<this .so>, tests/test_host_hip_double.py); nothing in
// the product knows about it.
//
// What it models, and how strictly:
//   * device memory is host memory (hipMalloc = aligned_alloc), filled with 0xDD so that a byte no copy ever wrote is
//     not a plausible zero;
//   * a stream is a THREAD with a queue: hipMemcpyAsync / hipMemsetAsync / event records / event waits / kernel
//     launches are queued and executed in order by that thread, asynchronously to the caller -- also for pageable host
//     memory, which the real runtime stages synchronously; the caller's buffer must stay untouched until the stream has
//     been synchronised, or the race detector sees the stream's thread and the caller on the same bytes;
//   * streams are independent of each other (all are "non-blocking"): only hipStreamSynchronize, hipEventSynchronize and
//     hipStreamWaitEvent order work across them.  The synchronous hipMemcpy copies at once, on the calling thread,
//     WITHOUT waiting for any stream -- the NULL stream's implicit synchronisation does not exist for non-blocking
//     streams, so relying on it is a bug the detector should see;
//   * kernels do not run: a launch is a queued no-op (device results stay 0xDD).  MI_HIP_STUB_KERNEL_US=<n> makes each
//     launch take n microseconds of stream time, to move the interleavings;
//   * MI_HIP_STUB_COPY_US=<n>: every queued copy sleeps n microseconds before it copies (widens race windows).
//   * MI_HIP_STUB_MALLOC_LIMIT_MB=<n>: hipMalloc of more than n MiB fails with hipErrorOutOfMemory (a tree larger than the device).
//   * the virtual-memory calls (the arena, mi_arena.hip): hipMemAddressReserve = mmap(PROT_NONE); a physical piece (hipMemCreate)
//     is a memfd filled with 0xDD, hipMemMap maps it MAP_FIXED | MAP_SHARED into the range, hipMemUnmap puts PROT_NONE back --
//     so a byte touched before the mapper got there, or after the pieces were let go of, is a SEGFAULT here, not a plausible
//     value, and a piece that is mapped again elsewhere (an arena that outgrew its range) still holds its bytes.  With MI_HIP_STUB_MALLOC_LIMIT_MB the pieces of all arenas together
//     may not exceed n MiB (hipMemCreate fails beyond, hipMemGetInfo reports what is left).  MI_HIP_STUB_MAP_US=<n>: every
//     hipMemCreate takes n microseconds per MiB (a box whose driver charges fresh device memory by the byte).
#include <hip/hip_runtime_api.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace {

long env_us(const char* name) {
    const char* e = getenv(name);
    return e ? atol(e) : 0;
}
const long kCopyUs = env_us("MI_HIP_STUB_COPY_US");
const long kKernelUs = env_us("MI_HIP_STUB_KERNEL_US");

struct Stream {
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::deque<std::function<void()>> q;
    bool busy = false, stop = false;
    std::thread th;
    Stream() : th([this] { run(); }) {}
    ~Stream() {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
        }
        cv_work.notify_all();
        th.join();
    }
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            f();
            {
                std::lock_guard<std::mutex> g(mu);
                busy = false;
            }
            cv_idle.notify_all();
        }
    }
    void push(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> g(mu);
            q.push_back(std::move(f));
        }
        cv_work.notify_one();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(mu);
        cv_idle.wait(lk, [&] { return q.empty() && !busy; });
    }
};

struct Event {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t recorded = 0, done = 0;                // generations: a record is complete when done >= its number
    std::chrono::steady_clock::time_point at;
};

Stream* null_stream() {                              // the library never uses it; kept for completeness
    static Stream* s = new Stream();
    return s;
}
Stream* S(hipStream_t s) { return s ? (Stream*)s : null_stream(); }

struct LaunchCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local LaunchCfg t_cfg;
thread_local int t_device = 0;

const long kMapUs = env_us("MI_HIP_STUB_MAP_US");
// MI_HIP_STUB_VM_FAIL=create:k | map:k | access:k : the k-th (0-based) hipMemCreate fails with hipErrorOutOfMemory, the k-th
// hipMemMap / hipMemSetAccess with hipErrorInvalidValue -- a device that runs out under the mapper, a runtime that refuses a piece;
// reserve:k : every hipMemAddressReserve from the k-th on fails -- a process that has used up its address space (ranges are never
// given back, mi_arena.hip)
struct VmFail { int what = 0; long at = -1; };
VmFail vm_fail() {
    static const VmFail f = [] {
        VmFail v;
        const char* e = getenv("MI_HIP_STUB_VM_FAIL");
        if (!e) return v;
        if (!strncmp(e, "create:", 7)) { v.what = 1; v.at = atol(e + 7); }
        if (!strncmp(e, "map:", 4)) { v.what = 2; v.at = atol(e + 4); }
        if (!strncmp(e, "access:", 7)) { v.what = 3; v.at = atol(e + 7); }
        if (!strncmp(e, "reserve:", 8)) { v.what = 4; v.at = atol(e + 8); }
        return v;
    }();
    return f;
}
std::atomic<long> g_vm_calls[5];
const long kLimitMb = env_us("MI_HIP_STUB_MALLOC_LIMIT_MB");
std::atomic<long long> g_vm_bytes{0};                 // physical pieces alive (hipMemCreate - hipMemRelease)
std::atomic<long> g_vm_pieces{0}, g_vm_ranges{0};
struct VmHandle { size_t bytes; int fd; };
std::atomic<long> g_live_allocs{0};
std::atomic<long> g_big_mallocs{0};                  // hipMalloc calls of a MiB and more: arenas (re)allocated

}  // namespace

extern "C" {

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d != 0) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int d) {
    if (d != 0) return hipErrorInvalidDevice;
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "HIP test double (no GPU)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = 288ull << 30;
    p->clockRate = 2400000;
    p->warpSize = 64;
    p->maxThreadsPerBlock = 1024;
    p->sharedMemPerBlock = 160 << 10;
    p->maxSharedMemoryPerMultiProcessor = 160 << 10;
    return hipSuccess;
}
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "error (HIP test double)"; }
hipError_t hipGetLastError(void) { return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) {
    static const long limit_mb = env_us("MI_HIP_STUB_MALLOC_LIMIT_MB");           // a device with little memory: larger requests fail
    if (limit_mb > 0 && n > ((size_t)limit_mb << 20)) return hipErrorOutOfMemory;
    void* q = aligned_alloc(4096, (n + 4095) / 4096 * 4096 + 4096);
    if (!q) return hipErrorOutOfMemory;
    memset(q, 0xDD, n);
    *p = q;
    ++g_live_allocs;
    if (n >= (1u << 20)) ++g_big_mallocs;
    return hipSuccess;
}
hipError_t hipFree(void* p) { if (p) { free(p); --g_live_allocs; } return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) {
    void* q = aligned_alloc(4096, (n + 4095) / 4096 * 4096 + 4096);
    if (!q) return hipErrorOutOfMemory;
    memset(q, 0xCC, n);
    *p = q;
    ++g_live_allocs;
    return hipSuccess;
}
hipError_t hipHostFree(void* p) { if (p) { free(p); --g_live_allocs; } return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = (hipStream_t) new Stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { if (s) { ((Stream*)s)->sync(); delete (Stream*)s; } return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { S(s)->sync(); return hipSuccess; }

hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) {
    if (n) memmove(dst, src, n);                     // at once, on the caller: no implicit wait for any stream
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s) {
    S(s)->push([=] {
        if (kCopyUs) usleep((useconds_t)kCopyUs);
        if (n) memmove(dst, src, n);
    });
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s) {
    S(s)->push([=] { if (n) memset(dst, v, n); });
    return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t) new Event(); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { delete (Event*)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    Event* ev = (Event*)e;
    uint64_t gen;
    {
        std::lock_guard<std::mutex> g(ev->mu);
        gen = ++ev->recorded;
    }
    S(s)->push([ev, gen] {
        std::lock_guard<std::mutex> g(ev->mu);       // (notified UNDER the lock: a waiter that destroys the event when it returns
        if (ev->done < gen) ev->done = gen;          //  cannot do so while the broadcast is still touching the condition variable)
        ev->at = std::chrono::steady_clock::now();
        ev->cv.notify_all();
    });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    Event* ev = (Event*)e;
    std::unique_lock<std::mutex> lk(ev->mu);
    const uint64_t gen = ev->recorded;
    ev->cv.wait(lk, [&] { return ev->done >= gen; });
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int) {
    Event* ev = (Event*)e;
    uint64_t gen;
    {
        std::lock_guard<std::mutex> g(ev->mu);
        gen = ev->recorded;                          // the most recent record at the time of the call
    }
    S(s)->push([ev, gen] {
        std::unique_lock<std::mutex> lk(ev->mu);
        ev->cv.wait(lk, [&] { return ev->done >= gen; });
    });
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    Event *x = (Event*)a, *y = (Event*)b;
    std::chrono::steady_clock::time_point ta, tb;
    {
        std::lock_guard<std::mutex> g(x->mu);
        ta = x->at;
    }
    {
        std::lock_guard<std::mutex> g(y->mu);
        tb = y->at;
    }
    *ms = std::chrono::duration<float, std::milli>(tb - ta).count();
    return hipSuccess;
}

hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t s) {
    S(s)->push([] { if (kKernelUs) usleep((useconds_t)kKernelUs); });
    return hipSuccess;
}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t s) {
    t_cfg = LaunchCfg{grid, block, shmem, s};
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* s) {
    *grid = t_cfg.grid; *block = t_cfg.block; *shmem = t_cfg.shmem; *s = t_cfg.stream;
    return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* h[1]; return h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned int, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}

// ---- virtual memory: an address range, physical pieces, mappings ----
hipError_t hipMemGetInfo(size_t* fr, size_t* total) {
    const size_t tot = kLimitMb > 0 ? (size_t)kLimitMb << 20 : (size_t)288 << 30;
    const size_t used = (size_t)g_vm_bytes.load();
    *total = tot;
    *fr = used < tot ? tot - used : 0;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }     // (callers have synchronised their streams: nothing device-wide here)
hipError_t hipMemGetAllocationGranularity(size_t* g, const hipMemAllocationProp*, hipMemAllocationGranularity_flags) { *g = 4096; return hipSuccess; }
hipError_t hipMemAddressReserve(void** p, size_t n, size_t, void*, unsigned long long) {
    if (vm_fail().what == 4 && g_vm_calls[4]++ >= vm_fail().at) return hipErrorOutOfMemory;
    void* q = mmap(nullptr, n, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (q == MAP_FAILED) return hipErrorOutOfMemory;
    *p = q;
    ++g_vm_ranges;
    return hipSuccess;
}
hipError_t hipMemAddressFree(void* p, size_t n) { munmap(p, n); --g_vm_ranges; return hipSuccess; }
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t* h, size_t n, const hipMemAllocationProp*, unsigned long long) {
    if (vm_fail().what == 1 && g_vm_calls[1]++ == vm_fail().at) return hipErrorOutOfMemory;
    if (kLimitMb > 0 && (size_t)g_vm_bytes.load() + n > ((size_t)kLimitMb << 20)) return hipErrorOutOfMemory;
    if (kMapUs) usleep((useconds_t)(kMapUs * (long)((n + (1u << 20) - 1) >> 20)));
    const int fd = memfd_create("mi_hip_stub_piece", MFD_CLOEXEC);
    if (fd < 0 || ftruncate(fd, (off_t)n) != 0) { if (fd >= 0) close(fd); return hipErrorOutOfMemory; }
    void* q = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (q == MAP_FAILED) { close(fd); return hipErrorOutOfMemory; }
    memset(q, 0xDD, n);
    munmap(q, n);
    g_vm_bytes += (long long)n;
    ++g_vm_pieces;
    *h = (hipMemGenericAllocationHandle_t) new VmHandle{n, fd};
    return hipSuccess;
}
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) {
    VmHandle* v = (VmHandle*)h;
    g_vm_bytes -= (long long)v->bytes;
    --g_vm_pieces;
    close(v->fd);
    delete v;
    return hipSuccess;
}
hipError_t hipMemMap(void* p, size_t n, size_t, hipMemGenericAllocationHandle_t h, unsigned long long) {
    VmHandle* v = (VmHandle*)h;
    if (v->bytes != n) return hipErrorInvalidValue;
    if (vm_fail().what == 2 && g_vm_calls[2]++ == vm_fail().at) return hipErrorInvalidValue;
    return mmap(p, n, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, v->fd, 0) == p ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipMemSetAccess(void*, size_t, const hipMemAccessDesc*, size_t) {
    if (vm_fail().what == 3 && g_vm_calls[3]++ == vm_fail().at) return hipErrorInvalidValue;
    return hipSuccess;
}
hipError_t hipMemUnmap(void* p, size_t n) {
    return mmap(p, n, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0) == p ? hipSuccess : hipErrorInvalidValue;
}
long mi_hip_stub_vm_pieces(void) { return g_vm_pieces.load(); }
long mi_hip_stub_vm_ranges(void) { return g_vm_ranges.load(); }
long long mi_hip_stub_vm_bytes(void) { return g_vm_bytes.load(); }

long mi_hip_stub_live_allocations(void) { return g_live_allocs.load(); }
long mi_hip_stub_big_mallocs(void) { return g_big_mallocs.load(); }

}  // extern "C"
