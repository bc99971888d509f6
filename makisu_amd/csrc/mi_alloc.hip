// mi_alloc.hip -- every device allocation of the library goes through dev_alloc / dev_free.
//
// Normally that is hipMalloc / hipFree.  With MI_GUARD_ALLOC=1 (the over-read audit, tests/test_gpu_overread.py)
// an allocation is built from the HIP virtual-memory calls instead: a reserved address range whose LAST page
// stays unmapped, physical memory mapped in front of it, and the buffer placed so that its last byte is the last
// mapped byte.  The kernels of this library read past the end of a string by design -- up to 63 bytes behind a
// SHA-256 string (67 with the cooperative loads), up to 127 behind a file in the Gear marking, whole 16-byte units
// in the CRC tiles -- and rely on slack the host code adds to every buffer (DevBuf: 256 bytes, the arena: 4 KiB).
// Under the guard that reliance is checked by the hardware: a load that leaves the slack hits an unmapped page
// and the process dies with "Memory access fault by GPU", naming the address.  A buffer gets exactly the bytes
// that were asked for (the callers' growth margins are switched off, see guard_alloc()), so the only slack is
// the documented one.  What the reference guarantees at this place: tario.WriteEntry copies exactly h.Size bytes
// (lib/tario/write.go:43-45), no more.
#include "mi_internal.h"

#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include <mutex>
#include <unordered_map>

namespace mi {

namespace {
struct GuardRec { void* va; size_t reserved, mapped; hipMemGenericAllocationHandle_t handle; unsigned long long seq; size_t bytes; };
std::atomic<unsigned long long> g_seq{0};
std::mutex g_mu;
std::unordered_map<void*, GuardRec> g_recs;
}  // namespace

// MI_GUARD_ALLOC=1: guarded allocations, and a freed buffer's address range is NEVER handed out again -- it stays
// reserved and unmapped, so a use after free faults too, whatever is allocated later.  (MI_GUARD_ALLOC=3 gives the
// range back with hipMemAddressFree.  On ROCm 7.0.2 that breaks batches whose arena grows: a range released and
// reserved again within microseconds -- a new mapping at an address that has just been unmapped -- ends in faults at
// addresses near no buffer of the trace, or in bytes that are not the ones copied, while the same run with the
// ranges kept (every stale pointer of this library would fault there) is clean: profiles/r04_overread_audit.txt.
// The product never does this -- it uses hipMalloc / hipFree -- so it is kept only as a way to show the effect.)
// MI_GUARD_TRACE=1 writes one line per allocation and free to stderr (sequence number, range, bytes), so that the
// address a fault names can be placed: behind a live buffer's slack, or inside a freed one.
static int guard_mode() {
    static const int m = [] { const char* v = getenv("MI_GUARD_ALLOC"); return v && *v ? atoi(v) : 0; }();
    return m;
}
static bool guard_trace() {
    static const bool on = [] { const char* v = getenv("MI_GUARD_TRACE"); return v && *v && *v != '0'; }();
    return on;
}
bool guard_alloc() { return guard_mode() > 0; }

hipError_t dev_alloc(void** p, size_t bytes) {
    if (!guard_alloc()) return hipMalloc(p, bytes);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran < 4096) gran = 4096;
    const size_t need = (bytes + 255) & ~(size_t)255;            // the buffer keeps its 256-byte alignment
    const size_t mapped = (need + gran - 1) / gran * gran;
    GuardRec r{nullptr, mapped + gran, mapped, {}, g_seq.fetch_add(1), bytes};
    e = hipMemAddressReserve(&r.va, r.reserved, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    e = hipMemCreate(&r.handle, mapped, &prop, 0);
    if (e != hipSuccess) { (void)hipMemAddressFree(r.va, r.reserved); return e; }
    e = hipMemMap(r.va, mapped, 0, r.handle, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(r.va, mapped, &acc, 1);
        if (e != hipSuccess) (void)hipMemUnmap(r.va, mapped);
    }
    if (e != hipSuccess) {
        (void)hipMemRelease(r.handle);
        (void)hipMemAddressFree(r.va, r.reserved);
        return e;
    }
    // the buffer ends within 255 bytes of the guard page, and exactly on it whenever `bytes` is a multiple of 256
    // (under the guard the arena and every DevBuf are)
    *p = (u8*)r.va + (mapped - need);
    if (guard_trace())
        fprintf(stderr, "mi_guard alloc #%llu [%p, +%zu) guard page at %p\n", r.seq, *p, bytes, (void*)((u8*)r.va + mapped));
    std::lock_guard<std::mutex> g(g_mu);
    g_recs[*p] = r;
    return hipSuccess;
}

hipError_t dev_free(void* p) {
    if (!p) return hipSuccess;
    GuardRec r;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_recs.find(p);
        if (it == g_recs.end()) return hipFree(p);
        r = it->second;
        g_recs.erase(it);
    }
    (void)hipDeviceSynchronize();                                // hipFree's implicit wait for work that still uses it
    if (guard_trace()) fprintf(stderr, "mi_guard free  #%llu [%p, +%zu)\n", r.seq, p, r.bytes);
    hipError_t e = hipMemUnmap(r.va, r.mapped);
    (void)hipMemRelease(r.handle);
    if (guard_mode() == 3) (void)hipMemAddressFree(r.va, r.reserved);  // else the range stays reserved, unmapped, for good
    return e;
}

}  // namespace mi
