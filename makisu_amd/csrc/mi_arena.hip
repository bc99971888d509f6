// mi_arena.hip -- the arena of a batch as a reserved address range that is mapped piece by piece.
//
// Until round 5 the arena was ONE hipMalloc.  An arena that had to grow while a walk was staging files into it was allocated
// again, larger; the reader threads were drained first (copies in flight target the old allocation), what the arena held was
// copied device to device, and on boxes whose driver charges fresh device memory by the byte (47-68 ms per GiB,
// tools/first_use_probe.py) every step paid for its full size -- all of it on the thread that walks the tree, before the commit's
// tar writer could start (48 x 128 MiB all new: 0.24 s of a 2.86 s commit).
//
// Now: hipMemAddressReserve once (addresses cost nothing), and a thread of the arena's own -- the mapper -- puts physical memory
// behind the front of the range as the promise grows: hipMemCreate + hipMemMap + hipMemSetAccess per piece -- pieces of ONE size
// per arena (32 MiB; MI_ARENA_PIECE_MB), in a range aligned to 1 GiB: hipMemSetAccess of both runtimes this library meets (the
// system's 7.2.0 and the 7.0.2 PyTorch bundles, which `import torch` makes the process's) answers "invalid argument" for some
// sequences of UNEQUAL piece sizes -- 2 MiB then 4, 6 then 2, 64 then 128 ... -- and for none of equal ones, thousands of them
// (tools/vmm_repro.hip, profiles/r06_vmm_repro.txt).  The adder's thread only moves
// a number (arena_promise); whoever is about to touch device memory at arena offset x -- a reader thread before its host-to-device
// copy, the inline window before its flush, stage_batch before the first kernel -- waits until the mapper has passed x
// (arena_wait_mapped), which a reader thread that has just read 8 MiB from a file practically never does.  Growing changes no
// address: nothing is drained, nothing is copied.  Measured (tools/ubench_vmm.hip, profiles/r06_ubench_vmm.txt): mapping 6 GiB
// in 256 MiB pieces 0.1 ms per GiB on a box whose hipMalloc is free too; host-to-device copies into pieces 56 GB/s (hipMalloc: 55),
// the first slab landed 0.6 ms after the first call, a coalesced read kernel 6.0 TB/s over pieces against 5.6 over one hipMalloc.
//
// Address ranges are RETIRED, never given back: on this ROCm a mapping placed at an address that has just been unmapped ends in
// GPU faults or in bytes that are not the ones copied (profiles/r04_overread_audit.txt, mi_alloc.hip on MI_GUARD_ALLOC=3), so an
// arena's range stays reserved for the life of the process after arena_release (physical memory IS given back).  A range is
// 32 GiB at least and four times the first promise; an arena that outgrows it has its pieces mapped again in a larger range (no
// copy: the same physical pieces), after the caller drained whatever targets it.  A process has 4 094 ranges of 32 GiB
// (tools/vmm_va_probe.hip, profiles/r06_vmm_va_probe.txt); a batch that finds none left gets a plain arena (arena_reserve,
// mi_api.hip: arena_promise reports *no_addresses instead of failing the batch).
//
// What the reference does here: nothing -- tario.WriteEntry (lib/tario/write.go:28-52) streams a file through a 32 KiB buffer.
// The arena exists because the GPU scans a whole batch at once.
#include "mi_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace mi {

namespace {
constexpr u64 kMinRange = 32ull << 30;                      // (addresses cost nothing; a 15 GB tree outgrew 8 GiB once: profiles/r06_real_tree_commit.txt)
// every piece of every arena (the driver's own minimum is 4 KiB): small enough that a batch of a few files does not hold much
// more than it needs, large enough that a 100 GB arena is a few thousand mappings
u64 piece_bytes() {
    static const u64 v = [] {
        const char* e = getenv("MI_ARENA_PIECE_MB");
        const long mb = e && *e ? atol(e) : 32;
        return (u64)(mb >= 1 && mb <= 4096 ? mb : 32) << 20;
    }();
    return v;
}

bool arena_trace() {
    static const bool on = [] { const char* v = getenv("MI_ARENA_TRACE"); return v && *v == '1'; }();
    return on;
}
u64 round_up(u64 v, u64 a) { return (v + a - 1) / a * a; }
}  // namespace

struct ArenaVm {
    int device = 0;
    bool fill = false;                                         // MI_FLAG_VERIFY_STAGING: new memory reads 0xA5 until a copy lands
    hipMemAllocationProp prop = {};
    void* va = nullptr;
    u64 reserved = 0;
    struct Piece { hipMemGenericAllocationHandle_t h; u64 bytes; };
    std::vector<Piece> pieces;
    std::mutex mu;
    std::condition_variable cv_work, cv_mapped;
    u64 piece = 0;                                             // every piece's size
    u64 target = 0, mapped = 0;                                // target: the promise (whole pieces)
    bool busy = false, stop = false;
    int err = MI_OK;
    std::string err_msg;
    std::thread th;
    hipStream_t fill_stream = nullptr;

    hipError_t map_at(void* base, u64 at, const Piece& pc) {
        hipError_t e = hipMemMap((u8*)base + at, pc.bytes, 0, pc.h, 0);
        if (e != hipSuccess) return e;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess((u8*)base + at, pc.bytes, &acc, 1);
        if (e != hipSuccess) (void)hipMemUnmap((u8*)base + at, pc.bytes);
        return e;
    }
    void run() {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || (mapped < target && err == MI_OK); });
            if (stop) break;
            const u64 at = mapped, take = piece;
            busy = true;
            lk.unlock();
            Piece pc{{}, take};
            const char* call = "hipMemCreate";
            hipError_t e = hipMemCreate(&pc.h, take, &prop, 0);
            bool made = e == hipSuccess;
            if (made) { call = "hipMemMap / hipMemSetAccess"; e = map_at(va, at, pc); }
            if (e == hipSuccess && fill) {
                if (!fill_stream) e = hipStreamCreateWithFlags(&fill_stream, hipStreamNonBlocking);
                if (e == hipSuccess) e = hipMemsetAsync((u8*)va + at, 0xA5, take, fill_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(fill_stream);
                if (e != hipSuccess) (void)hipMemUnmap((u8*)va + at, take);
            }
            if (e != hipSuccess && made) (void)hipMemRelease(pc.h);
            lk.lock();
            busy = false;
            if (e != hipSuccess) {
                err = e == hipErrorOutOfMemory ? MI_ERR_NOMEM : MI_ERR_HIP;
                char buf[256];
                snprintf(buf, sizeof buf, "the arena's next %llu bytes of device memory (%llu mapped so far, range %p + %llu): %s: %s",
                         (unsigned long long)take, (unsigned long long)at, va, (unsigned long long)reserved, call, hipGetErrorString(e));
                err_msg = buf;
            } else {
                pieces.push_back(pc);
                mapped = at + take;
                if (arena_trace()) fprintf(stderr, "mi_arena: mapped %.1f MB (+%.1f) of %.1f promised\n", mapped / 1e6, take / 1e6, target / 1e6);
            }
            cv_mapped.notify_all();
        }
    }
};

bool arena_is_plain() {
    static const bool plain = [] { const char* v = getenv("MI_ARENA"); return guard_alloc() || (v && !strcmp(v, "malloc")); }();
    return plain;
}
bool arena_always_pieces() {
    static const bool on = [] { const char* v = getenv("MI_ARENA"); return !guard_alloc() && v && !strcmp(v, "pieces"); }();
    return on;
}

u64 arena_piece_bytes(const Arena* a) { return a->vm ? a->vm->piece : 0; }

bool arena_outgrown(const Arena* a, u64 want) { return a->vm && round_up(want, a->vm->piece) > a->vm->reserved; }

void arena_counts(const Arena* a, u64* mapped, u64* pieces, u64* reserved) {
    u64 m = 0, n = 0, r = 0;
    if (a->vm) {
        std::lock_guard<std::mutex> g(a->vm->mu);
        m = a->vm->mapped;
        n = a->vm->pieces.size();
        r = a->vm->reserved;
    }
    if (mapped) *mapped = m;
    if (pieces) *pieces = n;
    if (reserved) *reserved = r;
}

int arena_promise(mi_ctx* c, Arena* a, u64 want, bool* no_addresses) {
    if (no_addresses) *no_addresses = false;
    if (want <= a->bytes) return MI_OK;
    // The promise moves in whole pieces: a batch that is filled file by file asks here for every file, and neither the free-memory
    // question below nor a wake-up of the mapper is asked per file.  (The arena of rounds 1-5 took a half on top of what was asked.)
    const u64 P = a->vm ? a->vm->piece : piece_bytes();
    const u64 need = round_up(want, P);
    const u64 cap = round_up((u64)c->prop.totalGlobalMem, 1ull << 30) + (1ull << 30);       // no arena can be larger than the device
    if (need > cap) return fail(c, MI_ERR_NOMEM, "an arena of %llu bytes on a device of %llu", (unsigned long long)want, (unsigned long long)c->prop.totalGlobalMem);
    ArenaVm* vm = a->vm;
    if (!vm) {
        vm = new ArenaVm();
        vm->device = c->device;
        vm->fill = c->verify_staging;
        vm->prop.type = hipMemAllocationTypePinned;
        vm->prop.location.type = hipMemLocationTypeDevice;
        vm->prop.location.id = c->device;
        vm->piece = P;
        u64 range = round_up(4 * need > kMinRange ? 4 * need : kMinRange, 1ull << 30);
        if (const char* e = getenv("MI_ARENA_RANGE_MB")) { const u64 v = (u64)atoll(e) << 20; if (v) range = round_up(v > need ? v : need, P); }   // (tests: a range that is outgrown)
        if (range > cap) range = cap;
        const hipError_t e = hipMemAddressReserve(&vm->va, range, 1ull << 30, nullptr, 0);
        if (e != hipSuccess) {
            delete vm;
            if (no_addresses) *no_addresses = true;            // (the caller may still take one allocation that moves)
            return fail(c, e == hipErrorOutOfMemory ? MI_ERR_NOMEM : MI_ERR_HIP, "hipMemAddressReserve of %llu bytes for an arena: %s", (unsigned long long)range, hipGetErrorString(e));
        }
        vm->reserved = range;
        vm->th = std::thread([vm] { vm->run(); });
        a->vm = vm;
        a->p = vm->va;
        if (arena_trace()) fprintf(stderr, "mi_arena: reserved %.1f MB of addresses at %p\n", range / 1e6, vm->va);
    }
    std::unique_lock<std::mutex> lk(vm->mu);
    if (vm->err) return fail(c, vm->err, "%s", vm->err_msg.c_str());
    // what is promised and not yet mapped will be taken from what is free now; tables of about a sixteenth of the arena's size
    // follow at mi_batch_run -- an arena the device cannot hold is refused HERE, while the caller can still go window by window
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            const u64 more = need > vm->mapped ? need - vm->mapped : 0;
            if (more > (u64)fr)
                return fail(c, MI_ERR_NOMEM, "the arena would need %llu more bytes of device memory, %llu are free", (unsigned long long)more, (unsigned long long)fr);
        }
    }
    if (need > vm->reserved) {
        // OUTGROWN: the same pieces, mapped again in a larger range (the caller drained what targets the arena).  The old
        // range stays reserved -- see the head of this file.
        vm->cv_mapped.wait(lk, [&] { return !vm->busy; });
        u64 range = round_up(4 * need > 2 * vm->reserved ? 4 * need : 2 * vm->reserved, 1ull << 30);
        if (range > cap) range = cap;
        void* nva = nullptr;
        hipError_t e = hipMemAddressReserve(&nva, range, 1ull << 30, nullptr, 0);
        if (e != hipSuccess) return fail(c, MI_ERR_HIP, "hipMemAddressReserve of %llu bytes for an arena that grew: %s", (unsigned long long)range, hipGetErrorString(e));
        (void)hipDeviceSynchronize();
        u64 at = 0;
        for (const ArenaVm::Piece& pc : vm->pieces) {
            if (e == hipSuccess) e = hipMemUnmap((u8*)vm->va + at, pc.bytes);
            if (e == hipSuccess) e = vm->map_at(nva, at, pc);
            at += pc.bytes;
        }
        if (e != hipSuccess) {
            vm->err = MI_ERR_HIP;
            vm->err_msg = std::string("moving the arena's pieces to a larger address range: ") + hipGetErrorString(e);
            return fail(c, vm->err, "%s", vm->err_msg.c_str());
        }
        if (arena_trace()) fprintf(stderr, "mi_arena: outgrew %.1f MB of addresses at %p: %zu pieces now at %p (%.1f MB reserved)\n", vm->reserved / 1e6, vm->va, vm->pieces.size(), nva, range / 1e6);
        vm->va = nva;
        vm->reserved = range;
        a->p = nva;
        ++a->moves;
    }
    if (arena_trace()) fprintf(stderr, "mi_arena: promise %.1f MB -> %.1f MB (%.1f mapped)\n", a->bytes / 1e6, need / 1e6, vm->mapped / 1e6);
    vm->target = need;
    a->bytes = need;
    lk.unlock();
    vm->cv_work.notify_one();
    return MI_OK;
}

int arena_wait_mapped(mi_ctx* c, Arena* a, u64 upto, std::string* msg) {
    ArenaVm* vm = a->vm;
    if (!vm) return MI_OK;                                     // the plain arena: allocated when it was reserved
    std::unique_lock<std::mutex> lk(vm->mu);
    if (upto > vm->target) upto = vm->target;                  // (slack the caller added: the promise's end is as far as anything reaches)
    vm->cv_mapped.wait(lk, [&] { return vm->mapped >= upto || vm->err != MI_OK; });
    if (vm->mapped >= upto) return MI_OK;
    if (msg) { *msg = vm->err_msg; return vm->err; }           // (a reader thread: the message goes to its batch, not to the ctx)
    return fail(c, vm->err, "%s", vm->err_msg.c_str());
}

void arena_release(Arena* a) {
    ArenaVm* vm = a->vm;
    if (!vm) {
        if (a->p) (void)dev_free(a->p);
    } else {
        (void)hipDeviceSynchronize();                          // (hipFree's implicit wait for work that still uses the memory)
        {
            std::lock_guard<std::mutex> g(vm->mu);
            vm->stop = true;
        }
        vm->cv_work.notify_all();
        if (vm->th.joinable()) vm->th.join();
        u64 at = 0;
        for (const ArenaVm::Piece& pc : vm->pieces) {
            (void)hipMemUnmap((u8*)vm->va + at, pc.bytes);
            (void)hipMemRelease(pc.h);
            at += pc.bytes;
        }
        if (vm->fill_stream) (void)hipStreamDestroy(vm->fill_stream);
        if (arena_trace()) fprintf(stderr, "mi_arena: released %.1f MB in %zu pieces; %.1f MB of addresses at %p retired\n", at / 1e6, vm->pieces.size(), vm->reserved / 1e6, vm->va);
        delete vm;
    }
    a->p = nullptr;
    a->bytes = 0;
    a->vm = nullptr;
}

}  // namespace mi
