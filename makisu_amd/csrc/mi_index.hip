// mi_index.hip -- a persistent, device-resident set of chunk digests: dedup ACROSS batches.
//
// The reference remembers what it has already built as content-addressed layers
// (CAS link + IsExist, lib/builder/step/common.go:88-91) and as cacheID -> "tarHex,gzipHex"
// entries behind keyvalue.Store (lib/cache/cache_manager.go:239-252,
// lib/cache/keyvalue/store.go:22-26).  The chunk-granular analogue: a set of
// chunk digests that outlives a batch -- "which chunks of this layer did an earlier layer
// already contain?" -- and that can be exported/imported as a flat byte blob so the shim can
// keep it behind the same keyvalue.Store next to the existing entries.
//
// Device side: open addressing, slot = {64-bit tag, 32-byte digest}; tag = the digest's first 8
// bytes (0 is stored as 1), 0 = empty.  mi_index_add_batch marks the batch's digests
// (mi_dedup_mark) and probes the UNIQUE rows only (dup_of == -1), in two kernels per round:
//   probe   walks from the row's slot comparing TAGS only -- the value atomicCAS returns, coherent
//           by itself: an empty slot is claimed (CAS 0 -> tag) and the digest stored, a slot with
//           an equal tag is remembered for the next kernel, anything else is walked past;
//   verify  (after the kernel boundary, when every digest stored by `probe` is visible) compares
//           all 32 bytes at the remembered slot: equal = "known"; different = a 64-bit tag
//           collision, the row goes on probing from the next slot in another round.
// No thread ever reads digest bytes another thread of the same launch may still be writing, and
// never trusts a slot's bytes without its tag, so recycled device memory cannot fake a match.
// Duplicate rows inherit the flag of the row they point to.  The table is rebuilt at twice
// the size when it gets more than half full.
#include "mi_internal.h"

#include <string.h>

using namespace mi;

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

namespace {

// a DevBuf that frees itself: temporaries stay leak-free on every error return
struct ScopedBuf : DevBuf {
    ~ScopedBuf() { release(); }
};

__device__ __forceinline__ bool digest_eq32(const u8* a, const u8* b) {
    const u32x4 a0 = ((const u32x4*)a)[0], a1 = ((const u32x4*)a)[1];
    const u32x4 b0 = ((const u32x4*)b)[0], b1 = ((const u32x4*)b)[1];
    const u32x4 d0 = a0 ^ b0, d1 = a1 ^ b1;
    return (d0.x | d0.y | d0.z | d0.w | d1.x | d1.y | d1.z | d1.w) == 0;
}

__device__ __forceinline__ u64 tag_of(const u8* d) {
    const u64 t = *(const u64*)d;
    return t ? t : 1ull;
}

// Row states while an insert is in progress.
enum : u8 { kRowDone = 0, kRowVerify = 1, kRowProbe = 2 };

// first round: rows with dup_of == -1 (or all rows when dup_of == nullptr) start probing
__global__ __launch_bounds__(256)
void index_begin_kernel(const u8* __restrict__ digests, const i64* __restrict__ dup_of, u64 n, u64 mask,
                        u8* __restrict__ row_state, u64* __restrict__ row_slot) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool probe = !(dup_of && dup_of[i] >= 0);
    row_state[i] = probe ? kRowProbe : kRowDone;
    row_slot[i] = tag_of(digests + 32 * i) & mask;
}

// known[i] = 0 if the row's digest was inserted now (known may be nullptr)
__global__ __launch_bounds__(256)
void index_probe_kernel(const u8* __restrict__ digests, u64 n, u64* __restrict__ tags,
                        u8* __restrict__ slots, u64 mask, u8* __restrict__ row_state,
                        u64* __restrict__ row_slot, u8* __restrict__ known, u64* __restrict__ n_new) {
    __shared__ u32 wg_new;
    if (threadIdx.x == 0) wg_new = 0;
    __syncthreads();
    u32 mine_new = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        if (row_state[i] != kRowProbe) continue;
        const u8* d = digests + 32 * i;
        const u64 tag = tag_of(d);
        u64 slot = row_slot[i];
        for (;;) {
            const u64 old = atomicCAS((unsigned long long*)&tags[slot], 0ull, (unsigned long long)tag);
            if (old == 0ull) {                               // empty: mine now
                ((u32x4*)(slots + 32 * slot))[0] = ((const u32x4*)d)[0];
                ((u32x4*)(slots + 32 * slot))[1] = ((const u32x4*)d)[1];
                if (known) known[i] = 0;
                row_state[i] = kRowDone;
                ++mine_new;
                break;
            }
            if (old == tag) {                                // full compare after the kernel boundary
                row_slot[i] = slot;
                row_state[i] = kRowVerify;
                break;
            }
            slot = (slot + 1) & mask;
        }
    }
    if (mine_new) atomicAdd(&wg_new, mine_new);
    __syncthreads();
    if (threadIdx.x == 0 && wg_new) atomicAdd((unsigned long long*)n_new, (unsigned long long)wg_new);
}

__global__ __launch_bounds__(256)
void index_verify_kernel(const u8* __restrict__ digests, u64 n, const u8* __restrict__ slots, u64 mask,
                         u8* __restrict__ row_state, u64* __restrict__ row_slot, u8* __restrict__ known,
                         u64* __restrict__ n_retry) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || row_state[i] != kRowVerify) return;
    const u64 slot = row_slot[i];
    if (digest_eq32(slots + 32 * slot, digests + 32 * i)) {
        if (known) known[i] = 1;
        row_state[i] = kRowDone;
    } else {                                                 // equal tags, different digests
        row_slot[i] = (slot + 1) & mask;
        row_state[i] = kRowProbe;
        atomicAdd((unsigned long long*)n_retry, 1ull);
    }
}

__global__ __launch_bounds__(256)
void index_inherit_kernel(const i64* __restrict__ dup_of, u64 n, u8* __restrict__ known) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const i64 d = dup_of[i];
    if (d >= 0) known[i] = known[d];                         // d < i and d is a unique row
}

__global__ __launch_bounds__(256)
void index_export_kernel(const u64* __restrict__ state, const u8* __restrict__ slots, u64 cap,
                         u8* __restrict__ out, u64* __restrict__ cursor) {
    const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap || state[s] == 0ull) return;
    const u64 at = atomicAdd((unsigned long long*)cursor, 1ull);
    ((u32x4*)(out + 32 * at))[0] = ((const u32x4*)(slots + 32 * s))[0];
    ((u32x4*)(out + 32 * at))[1] = ((const u32x4*)(slots + 32 * s))[1];
}

}  // namespace

struct mi_index {
    mi_ctx* ctx;
    DevBuf state, slots, counter, scratch, dup, row_state, row_slot;
    u64 cap = 0;           // slots, power of two
    u64 count = 0;         // digests held
};

namespace {

int index_alloc(mi_index* x, u64 cap) {
    mi_ctx* c = x->ctx;
    HIPCHK(c, x->state.ensure(cap * 8));
    HIPCHK(c, x->slots.ensure(cap * 32));
    HIPCHK(c, x->counter.ensure(16));
    HIPCHK(c, hipMemsetAsync(x->state.p, 0, cap * 8, c->stream));
    x->cap = cap;
    return MI_OK;
}

// inserts n device-resident digests (rows filtered by dup_of when given); grows first if needed
int index_insert(mi_index* x, const u8* d_digests, const i64* d_dup_of, u64 n, u8* d_known, u64* n_new);

int index_grow(mi_index* x, u64 min_cap) {
    mi_ctx* c = x->ctx;
    u64 cap = x->cap ? x->cap : 1024;
    while (cap < min_cap) cap <<= 1;
    if (cap == x->cap) return MI_OK;
    // export the current content, reallocate, re-insert
    ScopedBuf old;
    const u64 have = x->count;
    if (have) {
        HIPCHK(c, old.ensure(have * 32));
        HIPCHK(c, hipMemsetAsync(x->counter.p, 0, 8, c->stream));
        hipLaunchKernelGGL(index_export_kernel, dim3((u32)((x->cap + 255) / 256)), dim3(256), 0, c->stream,
                           x->state.as<u64>(), x->slots.as<u8>(), x->cap, old.as<u8>(), x->counter.as<u64>());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    x->state.release();
    x->slots.release();
    int rc = index_alloc(x, cap);
    if (rc) { x->cap = 0; x->count = 0; return rc; }     // out of memory: an empty, still usable index
    x->count = 0;
    if (have) {
        u64 n_new = 0;
        rc = index_insert(x, old.as<u8>(), nullptr, have, nullptr, &n_new);
        if (rc) return rc;
        if (n_new != have) return fail(c, MI_ERR_HIP, "index rebuild lost entries (%llu of %llu)",
                                       (unsigned long long)n_new, (unsigned long long)have);
    }
    return MI_OK;
}

int index_insert(mi_index* x, const u8* d_digests, const i64* d_dup_of, u64 n, u8* d_known, u64* n_new) {
    mi_ctx* c = x->ctx;
    *n_new = 0;
    if (n == 0) return MI_OK;
    if ((x->count + n) * 2 > x->cap) {                      // keep the load factor under 1/2
        int rc = index_grow(x, (x->count + n) * 2);
        if (rc) return rc;
    }
    HIPCHK(c, x->row_state.ensure(n + 16));
    HIPCHK(c, x->row_slot.ensure(n * 8 + 16));
    HIPCHK(c, hipMemsetAsync(x->counter.p, 0, 16, c->stream));   // [0] inserted, [1] rows to re-probe
    const u32 per_row = (u32)((n + 255) / 256);
    const u32 grid = per_row < 2048u ? per_row : 2048u;
    u64* d_cnt = x->counter.as<u64>();
    hipLaunchKernelGGL(index_begin_kernel, dim3(per_row), dim3(256), 0, c->stream, d_digests, d_dup_of, n,
                       x->cap - 1, x->row_state.as<u8>(), x->row_slot.as<u64>());
    for (int round = 0;; ++round) {
        hipLaunchKernelGGL(index_probe_kernel, dim3(grid), dim3(256), 0, c->stream, d_digests, n,
                           x->state.as<u64>(), x->slots.as<u8>(), x->cap - 1, x->row_state.as<u8>(),
                           x->row_slot.as<u64>(), d_known, d_cnt);
        hipLaunchKernelGGL(index_verify_kernel, dim3(per_row), dim3(256), 0, c->stream, d_digests, n,
                           x->slots.as<u8>(), x->cap - 1, x->row_state.as<u8>(), x->row_slot.as<u64>(),
                           d_known, d_cnt + 1);
        HIPCHK(c, hipMemcpyAsync(c->h_word, x->counter.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->h_word[1] == 0) break;                       // no 64-bit tag collisions (the normal case)
        if (round > 64) return fail(c, MI_ERR_HIP, "chunk index: probing does not converge");
        HIPCHK(c, hipMemsetAsync(d_cnt + 1, 0, 8, c->stream));
    }
    if (d_dup_of && d_known)
        hipLaunchKernelGGL(index_inherit_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, c->stream,
                           d_dup_of, n, d_known);
    HIPCHK(c, hipMemcpyAsync(c->h_word, x->counter.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    const u64 added = c->h_word[0];
    x->count += added;
    *n_new = added;
    return MI_OK;
}

}  // namespace

extern "C" {

int mi_index_create(mi_ctx* c, uint64_t capacity_hint, mi_index** out) {
    if (!c || !out) return MI_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    mi_index* x = new mi_index();
    x->ctx = c;
    ++c->live_children;                                 // mi_index_free undoes it
    u64 cap = 1024;
    while (cap < 2 * capacity_hint) cap <<= 1;
    int rc = index_alloc(x, cap);
    if (rc) { mi_index_free(x); return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *out = x;
    return MI_OK;
}

void mi_index_free(mi_index* x) {
    if (!x) return;
    (void)hipSetDevice(x->ctx->device);
    (void)hipStreamSynchronize(x->ctx->stream);
    x->state.release(); x->slots.release(); x->counter.release(); x->scratch.release(); x->dup.release();
    x->row_state.release(); x->row_slot.release();
    --x->ctx->live_children;
    delete x;
}

int mi_index_count(mi_index* x, uint64_t* n) {
    if (!x || !n) return MI_ERR_INVALID;
    *n = x->count;
    return MI_OK;
}

int mi_index_add_batch(mi_index* x, mi_batch* b, uint8_t* known_out, uint64_t cap, uint64_t* n_new,
                       uint64_t* n_known) {
    if (!x || !b || b->ctx != x->ctx) return MI_ERR_INVALID;
    mi_ctx* c = x->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!b->ran || b->in_flight) return fail(c, MI_ERR_STATE, "the batch must have run (and been waited for)");
    const u64 n = b->n_chunks;
    if (known_out && cap < n) return fail(c, MI_ERR_CAPACITY, "known buffer holds %llu rows, need %llu",
                                          (unsigned long long)cap, (unsigned long long)n);
    HIPCHK(c, x->scratch.ensure(n + 16));
    HIPCHK(c, x->dup.ensure(n * 8 + 16));
    // a fresh in-batch marking: the batch's own dup_of may be absent (MI_FLAG_NO_DEDUP) or hold
    // job-wide indices (mi_batch_set_global_dedup)
    u64 added = 0, uniq = 0;
    int rc = n ? mi_dedup_mark(c, b->digests.p, n, x->dup.p, &uniq) : MI_OK;
    if (rc) return rc;
    rc = index_insert(x, b->digests.as<u8>(), x->dup.as<i64>(), n, x->scratch.as<u8>(), &added);
    if (rc) return rc;
    std::vector<u8> known(n);
    if (n) HIPCHK(c, hipMemcpy(known.data(), x->scratch.p, n, hipMemcpyDeviceToHost));
    u64 nk = 0;
    for (u8 k : known) nk += k;
    if (known_out && n) memcpy(known_out, known.data(), n);
    if (n_new) *n_new = added;
    if (n_known) *n_known = nk;
    return MI_OK;
}

// mi_index_add_batch for digests that lie in HOST memory -- a batch that ran on another GPU than the index's (the commit over
// several ctxs feeds one index: 32 bytes per chunk cross the host, 0.4 % of the data).  known_out: n flags.  (hidden: mi_local.h)
int mi_index_add_digests(mi_index* x, const void* digests, uint64_t n, uint8_t* known_out, uint64_t* n_new, uint64_t* n_known) {
    if (!x || (!digests && n)) return MI_ERR_INVALID;
    mi_ctx* c = x->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (n_new) *n_new = 0;
    if (n_known) *n_known = 0;
    if (n == 0) return MI_OK;
    ScopedBuf d;
    HIPCHK(c, d.ensure(n * 32));
    HIPCHK(c, x->scratch.ensure(n + 16));
    HIPCHK(c, x->dup.ensure(n * 8 + 16));
    HIPCHK(c, hipMemcpy(d.p, digests, n * 32, hipMemcpyHostToDevice));
    u64 added = 0, uniq = 0;
    int rc = mi_dedup_mark(c, d.p, n, x->dup.p, &uniq);
    if (!rc) rc = index_insert(x, d.as<u8>(), x->dup.as<i64>(), n, x->scratch.as<u8>(), &added);
    if (rc) return rc;
    std::vector<u8> known(n);
    HIPCHK(c, hipMemcpy(known.data(), x->scratch.p, n, hipMemcpyDeviceToHost));
    u64 nk = 0;
    for (u8 k : known) nk += k;
    if (known_out) memcpy(known_out, known.data(), n);
    if (n_new) *n_new = added;
    if (n_known) *n_known = nk;
    return MI_OK;
}

int mi_index_same_ctx(mi_index* x, mi_batch* b) { return x && b && b->ctx == x->ctx ? 1 : 0; }

int mi_index_export(mi_index* x, void* out, uint64_t cap_digests) {
    if (!x || (!out && cap_digests)) return MI_ERR_INVALID;
    mi_ctx* c = x->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (cap_digests < x->count) return fail(c, MI_ERR_CAPACITY, "export buffer holds %llu digests, need %llu",
                                            (unsigned long long)cap_digests, (unsigned long long)x->count);
    if (x->count == 0) return MI_OK;
    ScopedBuf tmp;
    HIPCHK(c, tmp.ensure(x->count * 32));
    HIPCHK(c, hipMemsetAsync(x->counter.p, 0, 8, c->stream));
    hipLaunchKernelGGL(index_export_kernel, dim3((u32)((x->cap + 255) / 256)), dim3(256), 0, c->stream,
                       x->state.as<u64>(), x->slots.as<u8>(), x->cap, tmp.as<u8>(), x->counter.as<u64>());
    hipError_t e = hipMemcpyAsync(out, tmp.p, x->count * 32, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, MI_ERR_HIP, "mi_index_export: %s", hipGetErrorString(e));
    return MI_OK;
}

int mi_index_import(mi_index* x, const void* digests, uint64_t n, uint64_t* n_new) {
    if (!x || (!digests && n)) return MI_ERR_INVALID;
    mi_ctx* c = x->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (n_new) *n_new = 0;
    if (n == 0) return MI_OK;
    // imported digests may repeat each other or the table's content: mark them first so only
    // unique rows probe (the kernel's no-equal-probers precondition)
    ScopedBuf d, dup;
    HIPCHK(c, d.ensure(n * 32));
    HIPCHK(c, dup.ensure(n * 8));
    HIPCHK(c, hipMemcpy(d.p, digests, n * 32, hipMemcpyHostToDevice));
    uint64_t uniq = 0;
    int rc = mi_dedup_mark(c, d.p, n, dup.p, &uniq);
    u64 added = 0;
    if (!rc) rc = index_insert(x, d.as<u8>(), dup.as<i64>(), n, nullptr, &added);
    if (!rc && n_new) *n_new = added;
    return rc;
}

}  // extern "C"
