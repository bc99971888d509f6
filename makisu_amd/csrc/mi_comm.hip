// mi_comm.hip -- the ONE collective of the path, inside the C ABI: an RCCL all-gather of
// the per-GPU chunk-digest arrays over xGMI, then global duplicate marking.
//
// A Go host has no torch: the library itself must be able to exchange the digest sets.
// RCCL is bound at run time with dlopen("librccl.so.1") -- no link-time dependency, so a
// process that already carries an RCCL (e.g. PyTorch's) keeps exactly one copy.
//   multi-process (one rank per GPU):  rank 0 calls mi_comm_unique_id, ships the 128 bytes
//     to its peers by any side channel, every rank calls mi_comm_init_rank, then
//     mi_dedup_allgather(batch) after each mi_batch_run/wait;
//   single process, n GPUs (the natural shape for a Go host: one ctx per device):
//     mi_comm_init_all(ctxs, n) and mi_dedup_allgather_all(batches, n) -- the row counts are host
//     knowledge there (no counts collective, no host sync before the slabs), every allocation and
//     pad copy happens BEFORE the group, and the group holds nothing but the n ncclAllGather calls.
// Exchange: counts first (one u64 per rank), then slabs padded to the largest count in one
// ncclAllGather -- every xGMI link carries exactly one peer's slab, no ring of 7 hops --
// then the padding is squeezed out with device copies and the rank marks ITS OWN rows against
// the gathered set (mi_batch_mark_global); dup_of of the batch is rewritten with GLOBAL row
// indices (rank-major); a last 8-byte all-gather sums the ranks' first-occurrence counts.
// Host synchronisations per exchange: TWO -- the counts (the slab size is a host decision) and the
// end (round 2: three; the marking's first-occurrence count now goes from the kernel's device word
// straight into the summing all-gather); scalars travel through one pinned block, never through
// stack memory; the digest array is sent from where the batch keeps it whenever the padded slab
// fits its allocation.  MI_RCCL_LIB=<path> loads that library instead of librccl (the test double of
// tests/rccl_stub: n ranks on ONE GPU, which RCCL itself refuses).
#include "mi_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

using namespace mi;

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;   // (the all-to-all form only)
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        if (const char* over = getenv("MI_RCCL_LIB")) {         // a test double, or a differently named RCCL
            r.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) { r.err = std::string("cannot load MI_RCCL_LIB: ") + dlerror(); return; }
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) { r.err = std::string("cannot load librccl: ") + dlerror(); return; }
#define MI_SYM(field, sym)                                                    \
        r.field = (decltype(r.field))dlsym(r.lib, sym);                          \
        if (!r.field) { r.err = std::string("librccl lacks ") + sym; r.lib = nullptr; return; }
        MI_SYM(GetUniqueId, "ncclGetUniqueId")
        MI_SYM(CommInitRank, "ncclCommInitRank")
        MI_SYM(CommInitAll, "ncclCommInitAll")
        MI_SYM(CommDestroy, "ncclCommDestroy")
        MI_SYM(CommCount, "ncclCommCount")
        MI_SYM(AllGather, "ncclAllGather")
        MI_SYM(GroupStart, "ncclGroupStart")
        MI_SYM(GroupEnd, "ncclGroupEnd")
        MI_SYM(GetErrorString, "ncclGetErrorString")
#undef MI_SYM
        r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");   // absent from a library older than point-to-point: the
        r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");   // all-gather form does not need them
    });
    return &r;
}

#define NCCLCHK(c, call)                                                                     \
    do {                                                                                     \
        ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess)                                                               \
            return fail((c), MI_ERR_HIP, "%s failed: %s", #call, rccl()->GetErrorString(r_)); \
    } while (0)

int need_rccl(mi_ctx* c) {
    Rccl* r = rccl();
    if (!r->lib) return fail(c, MI_ERR_NO_DEVICE, "RCCL unavailable: %s", r->err.c_str());
    return MI_OK;
}

struct Shares;                          // the all-to-all form's buffers (below)

// per-exchange scratch kept on the ctx side of things (one exchange at a time per ctx)
struct Exchange {
    Shares* sh = nullptr;
    DevBuf counts, slab, gathered, compact, dup;
    u64* pin = nullptr;                 // pinned: [0] this rank's scalar, [1 .. nranks] gathered scalars
    size_t pin_words = 0;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // before the slab all-gather / after it / after the marking
    bool timed = false;                 // the three events of the last exchange were recorded
    bool a2a = false;                   // ... the last exchange was the all-to-all form: its seven events (Shares::ev) count
};
int ensure_events(mi_ctx* c, Exchange* x) {
    for (auto& e : x->ev)
        if (!e) HIPCHK(c, hipEventCreate(&e));
    return MI_OK;
}
int ensure_pin(mi_ctx* c, Exchange* x) {
    const size_t want = (size_t)c->comm_nranks + 1;
    if (x->pin_words >= want) return MI_OK;
    if (x->pin) (void)hipHostFree(x->pin);
    x->pin = nullptr;
    HIPCHK(c, hipHostMalloc((void**)&x->pin, want * 8, hipHostMallocDefault));
    x->pin_words = want;
    return MI_OK;
}
Exchange* exchange_of(mi_ctx* c) {
    if (!c->comm_scratch) c->comm_scratch = new Exchange();
    return (Exchange*)c->comm_scratch;
}

// phase 1: counts.  phase 2: slabs.  Split so the single-process form can group them.
int exchange_counts_enqueue(mi_batch* b) {
    mi_ctx* c = b->ctx;
    Exchange* x = exchange_of(c);
    HIPCHK(c, x->counts.ensure(8 * (size_t)(c->comm_nranks + 1)));
    int rc = ensure_pin(c, x);
    if (rc) return rc;
    u64* d_counts = x->counts.as<u64>();
    x->pin[0] = b->n_chunks;                               // pinned: the async copy needs no sync
    HIPCHK(c, hipMemcpyAsync(d_counts + c->comm_nranks, x->pin, 8, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, rccl()->AllGather(d_counts + c->comm_nranks, d_counts, 1, ncclUint64,
                                 (ncclComm_t)c->comm, c->stream));
    return MI_OK;
}

// the gathered counts on the host (sync 1 of 2 of the multi-process form: the slab size is a host decision)
int exchange_counts_read(mi_batch* b, std::vector<u64>& counts, u64* max_out) {
    mi_ctx* c = b->ctx;
    Exchange* x = exchange_of(c);
    counts.resize((size_t)c->comm_nranks);
    HIPCHK(c, hipMemcpyAsync(x->pin + 1, x->counts.p, 8 * counts.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    u64 m = 0;
    for (size_t r = 0; r < counts.size(); ++r) { counts[r] = x->pin[1 + r]; m = counts[r] > m ? counts[r] : m; }
    *max_out = m;
    return MI_OK;
}

// Everything the slab all-gather needs that is not the collective itself: the receive buffer, and -- when a
// peer holds more rows than this batch's digest buffer has room for -- a padded copy to send from.  No
// collective call in here, so it may run outside any group.
int exchange_slabs_prepare(mi_batch* b, u64 n_ranks, u64 m, const void** send_out) {
    mi_ctx* c = b->ctx;
    Exchange* x = exchange_of(c);
    int rc = ensure_events(c, x);
    if (rc) return rc;
    x->timed = false;
    *send_out = b->digests.p;                              // rows past n_chunks are padding nobody reads
    if (m == 0) return MI_OK;
    HIPCHK(c, x->gathered.ensure(m * 32 * n_ranks));
    if (m * 32 > b->digests.bytes) {                       // a peer holds more rows than this batch could
        HIPCHK(c, x->slab.ensure(m * 32));
        if (b->n_chunks)
            HIPCHK(c, hipMemcpyAsync(x->slab.p, b->digests.p, b->n_chunks * 32, hipMemcpyDeviceToDevice, c->stream));
        *send_out = x->slab.p;
    }
    HIPCHK(c, hipEventRecord(x->ev[0], c->stream));
    return MI_OK;
}

int exchange_slabs_enqueue(mi_batch* b, const void* send, u64 m) {
    mi_ctx* c = b->ctx;
    Exchange* x = exchange_of(c);
    if (m == 0) return MI_OK;
    NCCLCHK(c, rccl()->AllGather(send, x->gathered.p, m * 32, ncclUint8, (ncclComm_t)c->comm, c->stream));
    return MI_OK;
}

int mark_and_rewrite(mi_batch* b, const std::vector<u64>& counts, u64 m, uint64_t* n_total,
                     uint64_t* first_global) {
    mi_ctx* c = b->ctx;
    Exchange* x = exchange_of(c);
    u64 total = 0, first = 0;
    for (size_t r = 0; r < counts.size(); ++r) {
        if ((int)r < c->comm_rank) first += counts[r];
        total += counts[r];
    }
    const u8* glob = x->gathered.as<u8>();
    if (m) HIPCHK(c, hipEventRecord(x->ev[1], c->stream));
    bool ragged = false;
    for (u64 v : counts) ragged |= (v != m);
    if (ragged && total) {                                   // squeeze the padding out, rank-major
        HIPCHK(c, x->compact.ensure(total * 32));
        u64 at = 0;
        for (size_t r = 0; r < counts.size(); ++r) {
            if (counts[r])
                HIPCHK(c, hipMemcpyAsync(x->compact.as<u8>() + at * 32, glob + r * m * 32, counts[r] * 32,
                                         hipMemcpyDeviceToDevice, c->stream));
            at += counts[r];
        }
        glob = x->compact.as<u8>();
    }
    // no sync: the marking is enqueued on the same stream and leaves its first-occurrence count in the
    // ctx's device word; this rank answers for its own rows only, straight into the batch's dup_of column
    HIPCHK(c, b->dup_of.ensure(b->n_chunks * 8 + 16));      // absent when the ctx has MI_FLAG_NO_DEDUP
    int rc = mi_dedup_mark_range_enqueue(c, glob, total, first, b->n_chunks, b->dup_of.p);
    if (rc) return rc;
    if (m) {
        HIPCHK(c, hipEventRecord(x->ev[2], c->stream));
        x->timed = true;
        x->a2a = false;
    }
    b->results_valid = false;
    if (n_total) *n_total = total;
    if (first_global) *first_global = first;
    return MI_OK;
}

// sums the ranks' first-occurrence counts over the communicator: an all-gather of the marking kernel's
// device word (no host round trip in between), added on the host -- no second collective type to bind
int sum_over_ranks_enqueue(mi_ctx* c) {
    Exchange* x = exchange_of(c);
    NCCLCHK(c, rccl()->AllGather(c->dd_nuniq.p, x->counts.p, 1, ncclUint64, (ncclComm_t)c->comm, c->stream));
    return MI_OK;
}
int sum_over_ranks_finish(mi_ctx* c, u64* sum) {
    Exchange* x = exchange_of(c);
    HIPCHK(c, hipMemcpyAsync(x->pin + 1, x->counts.p, 8 * (size_t)c->comm_nranks, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));            // sync 2 of 2
    HIPCHK(c, hipGetLastError());
    *sum = 0;
    for (int r = 0; r < c->comm_nranks; ++r) *sum += x->pin[1 + r];
    return MI_OK;
}


// ---- the hash-partitioned form: two all-to-alls instead of the all-gather ("the optimisation") -------------
// The all-gather hands every rank ALL n_total digests (7/8 of them over xGMI at 8 ranks: 224 bytes received per own row) and
// every rank walks the rows of all earlier ranks through its table.  Here every digest has ONE owner -- the rank its second
// 8 bytes select (the marking's table slots by the first 8) -- and travels there once:
//   1. a rank splits its rows by owner, STABLY (part_hist / part_scan / part_scatter: per 256-row block a histogram, a scan
//      of the histograms per owner, a scatter that ranks equal owners within a wave by ballot) into send order: digests and
//      the rows' LOCAL indices (u32);
//   2. all-gather of n + 1 words per rank (its row count and how many rows go to each owner) -- the one host decision;
//   3. all-to-all #1 (grouped ncclSend / ncclRecv; the own share is a device copy): 36 bytes per row, each xGMI link carries
//      one peer's share.  An owner receives its shares in rank order, each in row order: the received set is in GLOBAL row order;
//   4. the owner marks the received set with the in-batch marking (launch_dedup_mark: the smallest equal index, or -1) --
//      every distinct digest of the job is counted exactly once, by its owner -- and turns the indices into global rows
//      (answer_kernel: the source rank's first global row + the local index that came with the digest);
//   5. all-to-all #2: the answers (8 bytes per row) go back the way the digests came; the rank scatters them into dup_of;
//   6. the owners' unique counts are summed as in the all-gather form.
// Per own row at 8 ranks: 44 bytes x 7/8 over xGMI instead of 224, and every rank's table work is n_total / n rows instead of
// own rows + a probe per earlier row.  Same results, bit for bit (tests/test_gpu_native_exchange.py runs both forms on every case).
}  // namespace

namespace mi {          // (named kernels: a profile lists them as mi::part_... / mi::answer_...)
constexpr int kMaxOwners = 64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32 owner_of(const u8* dg, u32 n) {
    const u64 w = ((const u64*)dg)[1];
    return (u32)(((w >> 32) * (u64)n) >> 32);
}

__global__ __launch_bounds__(256)
void part_hist_kernel(const u8* __restrict__ dg, u64 n_rows, u32 n, u32* __restrict__ hist) {
    __shared__ u32 wcnt[4][kMaxOwners];                        // counted by ballot, as the scatter ranks: no LDS atomics
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const u32 mine = i < n_rows ? owner_of(dg + 32 * i, n) : 0xFFFFFFFFu;
    for (u32 q = 0; q < n; ++q) {
        const u64 m = __ballot(mine == q);
        if (lane == 0) wcnt[wave][q] = (u32)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < n)
        hist[(u64)blockIdx.x * n + threadIdx.x] = wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
}

// hist[block][owner] -> the block's first position within the owner's share; to[owner] = the share's rows.  One workgroup per
// owner; its sixteen waves take a sixteenth of the blocks each: sum it, learn what lies before it, scan it (64 blocks a step).
__global__ __launch_bounds__(1024)
void part_scan_kernel(u32* __restrict__ hist, u32 n_blocks, u32 n, u64* __restrict__ to) {
    __shared__ u32 wsum[16];
    const u32 o = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 per = (n_blocks + 15) / 16;
    const u32 lo = wave * per < n_blocks ? wave * per : n_blocks, hi = lo + per < n_blocks ? lo + per : n_blocks;
    u32 acc = 0;
    for (u32 b = lo + lane; b < hi; b += 64) acc += hist[(u64)b * n + o];
    for (int d = 32; d; d >>= 1) acc += (u32)__shfl_xor((int)acc, d, 64);
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    u32 carry = 0, all = 0;
    for (u32 v = 0; v < 16; ++v) { if (v < wave) carry += wsum[v]; all += wsum[v]; }
    for (u32 b0 = lo; b0 < hi; b0 += 64) {
        const u32 b = b0 + lane;
        const u32 v = b < hi ? hist[(u64)b * n + o] : 0u;
        u32 incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const u32 t = (u32)__shfl_up((int)incl, d, 64);
            if ((int)lane >= d) incl += t;
        }
        if (b < hi) hist[(u64)b * n + o] = carry + incl - v;
        carry += (u32)__shfl((int)incl, 63, 64);
    }
    if (threadIdx.x == 0) to[o] = all;
}

__global__ __launch_bounds__(256)
void part_scatter_kernel(const u8* __restrict__ dg, u64 n_rows, u32 n, const u32* __restrict__ hist,
                         const u64* __restrict__ to, u8* __restrict__ out_dg, u32* __restrict__ out_row) {
    __shared__ u32 wcnt[4][kMaxOwners], base[kMaxOwners];     // base: where each owner's share begins in send order
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (u32 q = 0; q < n; ++q) { base[q] = run; run += (u32)to[q]; }
    }
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n_rows;
    const u32 mine = live ? owner_of(dg + 32 * i, n) : 0xFFFFFFFFu;
    u32 before = 0;                                            // rows of my owner in earlier lanes of my wave
    for (u32 q = 0; q < n; ++q) {
        const u64 m = __ballot(mine == q);
        if (mine == q) before = (u32)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave][q] = (u32)__popcll(m);
    }
    __syncthreads();
    if (!live) return;
    u32 pos = base[mine] + hist[(u64)blockIdx.x * n + mine] + before;
    for (u32 w = 0; w < wave; ++w) pos += wcnt[w][mine];
    const u32x4 a = ((const u32x4*)(dg + 32 * i))[0], b = ((const u32x4*)(dg + 32 * i))[1];
    ((u32x4*)(out_dg + 32ull * pos))[0] = a;
    ((u32x4*)(out_dg + 32ull * pos))[1] = b;
    out_row[pos] = (u32)i;
}

// dup[j]: index of the first equal row WITHIN the received set, or -1  ->  that row's GLOBAL index (seg: where each source
// rank's share begins in the received set, n + 1 entries; first: each rank's first global row)
__global__ __launch_bounds__(256)
void answer_kernel(i64* __restrict__ dup, const u32* __restrict__ row, const u64* __restrict__ seg,
                   const u64* __restrict__ first, u32 n, u64 n_recv) {
    __shared__ u64 s_seg[kMaxOwners + 1], s_first[kMaxOwners];
    if (threadIdx.x <= n) s_seg[threadIdx.x] = seg[threadIdx.x];
    if (threadIdx.x < n) s_first[threadIdx.x] = first[threadIdx.x];
    __syncthreads();
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_recv) return;
    const i64 d = dup[j];
    if (d < 0) return;
    u32 q = 0;
    while (q + 1 < n && s_seg[q + 1] <= (u64)d) ++q;
    dup[j] = (i64)(s_first[q] + row[d]);
}

__global__ __launch_bounds__(256)
void answer_scatter_kernel(const i64* __restrict__ back, const u32* __restrict__ row, u64 n_rows, i64* __restrict__ dup_of) {
    const u64 p = (u64)blockIdx.x * 256 + threadIdx.x;
    if (p < n_rows) dup_of[row[p]] = back[p];
}

}  // namespace mi

namespace {

struct Shares {                          // the all-to-all form's state of one ctx (next to its Exchange)
    DevBuf hist, cnt_send, cnt_all, send_dg, send_row, recv_dg, recv_row, ans, back, meta;
    u64* pin = nullptr;
    size_t pin_words = 0;
    std::vector<u64> to, from, soff, roff, first;             // rows per peer and where they lie (send order / received set)
    u64 n_recv = 0, n_total = 0;
    // before / after: the split | all-to-all #1 | (marking + answers) | all-to-all #2 | (scatter)
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void release() {
        for (auto& e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        for (DevBuf* d : {&hist, &cnt_send, &cnt_all, &send_dg, &send_row, &recv_dg, &recv_row, &ans, &back, &meta}) d->release();
        if (pin) (void)hipHostFree(pin);
        pin = nullptr;
        pin_words = 0;
    }
};
Shares* shares_of(mi_ctx* c) {
    Exchange* x = exchange_of(c);
    if (!x->sh) x->sh = new Shares();
    return x->sh;
}

int need_p2p(mi_ctx* c) {
    int rc = need_rccl(c);
    if (rc) return rc;
    if (!rccl()->Send || !rccl()->Recv) return fail(c, MI_ERR_NO_DEVICE, "the collective library has no ncclSend / ncclRecv");
    if (c->comm_nranks > kMaxOwners) return fail(c, MI_ERR_INVALID, "the all-to-all form takes at most %d ranks", kMaxOwners);
    return MI_OK;
}

// step 1: the rows in send order and, in cnt_send[1 .. n], how many go to each owner (cnt_send[0]: the batch's rows)
int shares_split_enqueue(mi_batch* b) {
    mi_ctx* c = b->ctx;
    Shares* s = shares_of(c);
    const u32 n = (u32)c->comm_nranks;
    const u64 rows = b->n_chunks;
    if (rows >= 0xFFFFFFFFull) return fail(c, MI_ERR_INVALID, "dedup set too large");
    const size_t words = (size_t)(n + 1) * n + 2 * (size_t)n + 2;
    if (s->pin_words < words) {
        if (s->pin) (void)hipHostFree(s->pin);
        s->pin = nullptr;
        s->pin_words = 0;
        HIPCHK(c, hipHostMalloc((void**)&s->pin, words * 8, hipHostMallocDefault));
        s->pin_words = words;
    }
    const u32 n_blocks = (u32)((rows + 255) / 256);
    HIPCHK(c, s->cnt_send.ensure(8 * (size_t)(n + 1)));
    HIPCHK(c, s->cnt_all.ensure(8 * (size_t)(n + 1) * n));
    HIPCHK(c, s->hist.ensure(4 * (size_t)n * (n_blocks ? n_blocks : 1)));
    HIPCHK(c, s->send_dg.ensure(rows * 32 + 32));
    HIPCHK(c, s->send_row.ensure(rows * 4 + 16));
    for (auto& e : s->ev)
        if (!e) HIPCHK(c, hipEventCreate(&e));
    exchange_of(c)->timed = false;
    s->pin[0] = rows;
    HIPCHK(c, hipMemcpyAsync(s->cnt_send.p, s->pin, 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(s->ev[0], c->stream));
    if (rows == 0) {
        HIPCHK(c, hipMemsetAsync(s->cnt_send.as<u64>() + 1, 0, 8 * (size_t)n, c->stream));
        HIPCHK(c, hipEventRecord(s->ev[1], c->stream));
        return MI_OK;
    }
    hipLaunchKernelGGL(part_hist_kernel, dim3(n_blocks), dim3(256), 0, c->stream, b->digests.as<u8>(), rows, n, s->hist.as<u32>());
    hipLaunchKernelGGL(part_scan_kernel, dim3(n), dim3(1024), 0, c->stream, s->hist.as<u32>(), n_blocks, n,
                       s->cnt_send.as<u64>() + 1);
    hipLaunchKernelGGL(part_scatter_kernel, dim3(n_blocks), dim3(256), 0, c->stream, b->digests.as<u8>(), rows, n,
                       s->hist.as<u32>(), s->cnt_send.as<u64>() + 1, s->send_dg.as<u8>(), s->send_row.as<u32>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(s->ev[1], c->stream));
    return MI_OK;
}

// step 2's host side: from the job's matrix m[q * (n + 1)] = rank q's rows, m[q * (n + 1) + 1 + o] = rows of q owned by o --
// what this rank sends, receives, and where; the received set's layout goes to the device for answer_kernel
int shares_plan(mi_batch* b, const u64* m) {
    mi_ctx* c = b->ctx;
    Shares* s = shares_of(c);
    const size_t n = (size_t)c->comm_nranks, me = (size_t)c->comm_rank;
    s->to.assign(n, 0); s->from.assign(n, 0); s->soff.assign(n + 1, 0); s->roff.assign(n + 1, 0); s->first.assign(n, 0);
    u64 total = 0;
    for (size_t q = 0; q < n; ++q) {
        s->first[q] = total;
        total += m[q * (n + 1)];
        s->to[q] = m[me * (n + 1) + 1 + q];
        s->from[q] = m[q * (n + 1) + 1 + me];
        s->soff[q + 1] = s->soff[q] + s->to[q];
        s->roff[q + 1] = s->roff[q] + s->from[q];
    }
    if (s->soff[n] != b->n_chunks)
        return fail(c, MI_ERR_STATE, "the owners' shares of this rank add up to %llu rows, the batch has %llu",
                    (unsigned long long)s->soff[n], (unsigned long long)b->n_chunks);
    if (total >= 0xFFFFFFFFull) return fail(c, MI_ERR_INVALID, "dedup set too large");
    s->n_total = total;
    s->n_recv = s->roff[n];
    HIPCHK(c, s->recv_dg.ensure(s->n_recv * 32 + 32));
    HIPCHK(c, s->recv_row.ensure(s->n_recv * 4 + 16));
    HIPCHK(c, s->ans.ensure(s->n_recv * 8 + 16));
    HIPCHK(c, s->back.ensure(b->n_chunks * 8 + 16));
    HIPCHK(c, s->meta.ensure(8 * (2 * n + 1)));
    u64* pm = s->pin + (n + 1) * n + 1;                        // (behind the matrix's place in the pinned block)
    for (size_t q = 0; q <= n; ++q) pm[q] = s->roff[q];
    for (size_t q = 0; q < n; ++q) pm[n + 1 + q] = s->first[q];
    HIPCHK(c, hipMemcpyAsync(s->meta.p, pm, 8 * (2 * n + 1), hipMemcpyHostToDevice, c->stream));
    // the marking's scratch, before any group opens
    u64 cap = 1024;
    while (cap < 2 * s->n_recv) cap <<= 1;
    HIPCHK(c, c->dd_table.ensure(cap * 8));
    HIPCHK(c, c->dd_slot.ensure(s->n_recv * 4 + 16));
    HIPCHK(c, c->dd_nuniq.ensure(8));
    HIPCHK(c, b->dup_of.ensure(b->n_chunks * 8 + 16));       // absent when the ctx has MI_FLAG_NO_DEDUP
    HIPCHK(c, hipEventRecord(s->ev[2], c->stream));
    return MI_OK;
}

// all-to-all #1 (back == false): shares out, the received set in; #2 (back == true): the answers the other way.  The caller
// holds the group open; the own share is a device copy on the ctx's stream.
int shares_exchange_enqueue(mi_batch* b, bool back) {
    mi_ctx* c = b->ctx;
    Shares* s = shares_of(c);
    const int n = c->comm_nranks, me = c->comm_rank;
    ncclComm_t comm = (ncclComm_t)c->comm;
    for (int q = 0; q < n; ++q) {
        const u64 out = back ? s->from[(size_t)q] : s->to[(size_t)q], in = back ? s->to[(size_t)q] : s->from[(size_t)q];
        const u64 oo = back ? s->roff[(size_t)q] : s->soff[(size_t)q], io = back ? s->soff[(size_t)q] : s->roff[(size_t)q];
        if (q == me) {
            if (!out) continue;
            if (back) {
                HIPCHK(c, hipMemcpyAsync(s->back.as<u8>() + io * 8, s->ans.as<u8>() + oo * 8, out * 8, hipMemcpyDeviceToDevice, c->stream));
            } else {
                HIPCHK(c, hipMemcpyAsync(s->recv_dg.as<u8>() + io * 32, s->send_dg.as<u8>() + oo * 32, out * 32, hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(s->recv_row.as<u8>() + io * 4, s->send_row.as<u8>() + oo * 4, out * 4, hipMemcpyDeviceToDevice, c->stream));
            }
            continue;
        }
        if (back) {
            if (out) NCCLCHK(c, rccl()->Send(s->ans.as<u8>() + oo * 8, out * 8, ncclUint8, q, comm, c->stream));
            if (in) NCCLCHK(c, rccl()->Recv(s->back.as<u8>() + io * 8, in * 8, ncclUint8, q, comm, c->stream));
        } else {
            if (out) {
                NCCLCHK(c, rccl()->Send(s->send_dg.as<u8>() + oo * 32, out * 32, ncclUint8, q, comm, c->stream));
                NCCLCHK(c, rccl()->Send(s->send_row.as<u8>() + oo * 4, out * 4, ncclUint8, q, comm, c->stream));
            }
            if (in) {
                NCCLCHK(c, rccl()->Recv(s->recv_dg.as<u8>() + io * 32, in * 32, ncclUint8, q, comm, c->stream));
                NCCLCHK(c, rccl()->Recv(s->recv_row.as<u8>() + io * 4, in * 4, ncclUint8, q, comm, c->stream));
            }
        }
    }
    return MI_OK;
}

// step 4: the owner's marking of what it received, answers as global rows (in s->ans); its unique count in c->dd_nuniq
int shares_mark_enqueue(mi_batch* b) {
    mi_ctx* c = b->ctx;
    Shares* s = shares_of(c);
    HIPCHK(c, hipEventRecord(s->ev[3], c->stream));
    u64 cap = 1024;
    while (cap < 2 * s->n_recv) cap <<= 1;
    launch_dedup_mark(s->recv_dg.as<u8>(), s->n_recv, nullptr, c->dd_table.as<u32>(), c->dd_slot.as<u32>(), cap,
                      s->ans.as<i64>(), c->dd_nuniq.as<u64>(), true, c->stream);
    if (s->n_recv) {
        const u32 n = (u32)c->comm_nranks;
        hipLaunchKernelGGL(answer_kernel, dim3((u32)((s->n_recv + 255) / 256)), dim3(256), 0, c->stream, s->ans.as<i64>(),
                           s->recv_row.as<u32>(), s->meta.as<u64>(), s->meta.as<u64>() + n + 1, n, s->n_recv);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(s->ev[4], c->stream));
    return MI_OK;
}

// step 5's end: the answers that came back, into the batch's dup_of column
int shares_finish_enqueue(mi_batch* b) {
    mi_ctx* c = b->ctx;
    Shares* s = shares_of(c);
    Exchange* x = exchange_of(c);
    HIPCHK(c, hipEventRecord(s->ev[5], c->stream));
    if (b->n_chunks)
        hipLaunchKernelGGL(answer_scatter_kernel, dim3((u32)((b->n_chunks + 255) / 256)), dim3(256), 0, c->stream,
                           s->back.as<i64>(), s->send_row.as<u32>(), b->n_chunks, b->dup_of.as<i64>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(s->ev[6], c->stream));
    x->timed = true;
    x->a2a = true;
    b->results_valid = false;
    return MI_OK;
}

int group_end(mi_ctx* c, int rc) {                            // always closed, also after a failed enqueue
    const ncclResult_t r_ = rccl()->GroupEnd();
    if (rc) return rc;
    if (r_ != ncclSuccess) return fail(c, MI_ERR_HIP, "ncclGroupEnd failed: %s", rccl()->GetErrorString(r_));
    return MI_OK;
}

}  // namespace

extern "C" {

int mi_comm_unique_id(void* id_out) {
    if (!id_out) return MI_ERR_INVALID;
    int rc = need_rccl(nullptr);
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = rccl()->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, MI_ERR_HIP, "ncclGetUniqueId: %s", rccl()->GetErrorString(r));
    static_assert(sizeof id == MI_COMM_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof id);
    return MI_OK;
}

int mi_comm_init_rank(mi_ctx* c, int nranks, int rank, const void* id) {
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return MI_ERR_INVALID;
    int rc = need_rccl(c);
    if (rc) return rc;
    if (c->comm) return fail(c, MI_ERR_STATE, "communicator already initialised");
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    NCCLCHK(c, rccl()->CommInitRank(&comm, nranks, uid, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_nranks = nranks;
    return MI_OK;
}

int mi_comm_init_all(mi_ctx** ctxs, int n) {
    if (!ctxs || n < 1) return MI_ERR_INVALID;
    int rc = need_rccl(ctxs[0]);
    if (rc) return rc;
    std::vector<int> devs((size_t)n);
    std::vector<ncclComm_t> comms((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return MI_ERR_INVALID;
        if (ctxs[i]->comm) return fail(ctxs[i], MI_ERR_STATE, "communicator already initialised");
        devs[(size_t)i] = ctxs[i]->device;
    }
    NCCLCHK(ctxs[0], rccl()->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        ctxs[i]->comm = comms[(size_t)i];
        ctxs[i]->comm_rank = i;
        ctxs[i]->comm_nranks = n;
    }
    return MI_OK;
}

int mi_comm_ranks(mi_ctx* c, int* n_ranks) {
    if (!c || !n_ranks) return MI_ERR_INVALID;
    *n_ranks = 0;
    if (!c->comm) return MI_OK;                // no communicator: 0 ranks
    NCCLCHK(c, rccl()->CommCount((ncclComm_t)c->comm, n_ranks));
    return MI_OK;
}

int mi_comm_destroy(mi_ctx* c) {
    if (!c) return MI_ERR_INVALID;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)rccl()->CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    if (c->comm_scratch) {
        Exchange* x = (Exchange*)c->comm_scratch;
        x->counts.release(); x->slab.release(); x->gathered.release(); x->compact.release(); x->dup.release();
        if (x->sh) { x->sh->release(); delete x->sh; }
        if (x->pin) (void)hipHostFree(x->pin);
        for (auto e : x->ev) if (e) (void)hipEventDestroy(e);
        delete x;
        c->comm_scratch = nullptr;
    }
    c->comm_rank = 0;
    c->comm_nranks = 1;
    return MI_OK;
}

int mi_comm_exchange_ms(mi_ctx* c, double* ms_gather, double* ms_marking) {
    if (!c) return MI_ERR_INVALID;
    if (ms_gather) *ms_gather = 0;
    if (ms_marking) *ms_marking = 0;
    Exchange* x = (Exchange*)c->comm_scratch;
    if (!x || !x->timed) return MI_OK;                     // no exchange yet, or one without rows
    HIPCHK(c, hipSetDevice(c->device));
    if (x->a2a) {                                              // the wire: both all-to-alls; the rest: split, marking + answers, scatter
        const hipEvent_t* e = x->sh->ev;
        HIPCHK(c, hipEventSynchronize(e[6]));
        float t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; ++i)
            if (i != 1) HIPCHK(c, hipEventElapsedTime(&t[i], e[i], e[i + 1]));      // (e[1] -> e[2]: the counts and the host's plan)
        if (ms_gather) *ms_gather = t[2] + t[4];
        if (ms_marking) *ms_marking = t[0] + t[3] + t[5];
        return MI_OK;
    }
    HIPCHK(c, hipEventSynchronize(x->ev[2]));
    float a = 0, b = 0;
    HIPCHK(c, hipEventElapsedTime(&a, x->ev[0], x->ev[1]));
    HIPCHK(c, hipEventElapsedTime(&b, x->ev[1], x->ev[2]));
    if (ms_gather) *ms_gather = a;
    if (ms_marking) *ms_marking = b;
    return MI_OK;
}

int mi_dedup_allgather(mi_batch* b, uint64_t* n_total, uint64_t* n_unique, uint64_t* first_global) {
    if (!b) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->comm) return fail(c, MI_ERR_STATE, "mi_dedup_allgather before mi_comm_init_rank/_all");
    if (!b->ran || b->in_flight) return fail(c, MI_ERR_STATE, "the batch must have run (and been waited for)");
    int rc = exchange_counts_enqueue(b);
    if (rc) return rc;
    std::vector<u64> counts;
    u64 m = 0;
    rc = exchange_counts_read(b, counts, &m);
    if (rc) return rc;
    const void* send = nullptr;
    rc = exchange_slabs_prepare(b, counts.size(), m, &send);
    if (rc) return rc;
    rc = exchange_slabs_enqueue(b, send, m);
    if (rc) return rc;
    rc = mark_and_rewrite(b, counts, m, n_total, first_global);
    if (rc) return rc;
    rc = sum_over_ranks_enqueue(c);
    if (rc) return rc;
    u64 sum = 0;
    rc = sum_over_ranks_finish(c, &sum);
    if (rc) return rc;
    if (n_unique) *n_unique = sum;
    return MI_OK;
}

int mi_dedup_allgather_all(mi_batch** batches, int n, uint64_t* n_total, uint64_t* n_unique) {
    if (!batches || n < 1) return MI_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        if (!batches[i]) return MI_ERR_INVALID;
        mi_ctx* c = batches[i]->ctx;
        if (!c->comm || c->comm_nranks != n || c->comm_rank != i)
            return fail(c, MI_ERR_STATE, "batch %d does not belong to rank %d of an %d-rank mi_comm_init_all", i, i, n);
        if (!batches[i]->ran || batches[i]->in_flight)
            return fail(c, MI_ERR_STATE, "every batch must have run (and been waited for)");
    }
    mi_ctx* c0 = batches[0]->ctx;
    int rc = MI_OK;
    // every rank's row count is known to this process: no counts collective, no host sync before the slabs
    std::vector<u64> counts((size_t)n);
    u64 m = 0;
    for (int i = 0; i < n; ++i) { counts[(size_t)i] = batches[i]->n_chunks; m = counts[(size_t)i] > m ? counts[(size_t)i] : m; }
    std::vector<const void*> send((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) {              // allocations and pad copies: outside the group
        HIPCHK(batches[i]->ctx, hipSetDevice(batches[i]->ctx->device));
        rc = exchange_slabs_prepare(batches[i], (u64)n, m, &send[(size_t)i]);
        if (rc) return rc;
    }
    NCCLCHK(c0, rccl()->GroupStart());
    for (int i = 0; i < n && !rc; ++i) {       // the group holds the n collectives and nothing else
        (void)hipSetDevice(batches[i]->ctx->device);
        rc = exchange_slabs_enqueue(batches[i], send[(size_t)i], m);
    }
    {
        const ncclResult_t r_ = rccl()->GroupEnd();        // always closed, also after a failed enqueue
        if (rc) return rc;
        if (r_ != ncclSuccess) return fail(c0, MI_ERR_HIP, "ncclGroupEnd failed: %s", rccl()->GetErrorString(r_));
    }
    for (int i = 0; i < n; ++i) {              // every rank's marking enqueued on its own stream ...
        (void)hipSetDevice(batches[i]->ctx->device);
        uint64_t nt = 0;
        rc = mark_and_rewrite(batches[i], counts, m, &nt, nullptr);
        if (rc) return rc;
        if (n_total) *n_total = nt;
    }
    uint64_t unique_sum = 0;
    for (int i = 0; i < n; ++i) {              // ... then read: the counts are in this process, no collective
        mi_ctx* c = batches[i]->ctx;
        (void)hipSetDevice(c->device);
        HIPCHK(c, hipMemcpyAsync(c->h_word, c->dd_nuniq.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        unique_sum += *c->h_word;              // every rank counted its own first occurrences
    }
    if (n_unique) *n_unique = unique_sum;
    return MI_OK;
}

// The all-to-all form of mi_dedup_allgather: same arguments, same results (see the comment above owner_of).
int mi_dedup_alltoall(mi_batch* b, uint64_t* n_total, uint64_t* n_unique, uint64_t* first_global) {
    if (!b) return MI_ERR_INVALID;
    mi_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->comm) return fail(c, MI_ERR_STATE, "mi_dedup_alltoall before mi_comm_init_rank/_all");
    if (!b->ran || b->in_flight) return fail(c, MI_ERR_STATE, "the batch must have run (and been waited for)");
    int rc = need_p2p(c);
    if (rc) return rc;
    const size_t n = (size_t)c->comm_nranks;
    rc = shares_split_enqueue(b);
    if (rc) return rc;
    Shares* s = shares_of(c);
    NCCLCHK(c, rccl()->AllGather(s->cnt_send.p, s->cnt_all.p, n + 1, ncclUint64, (ncclComm_t)c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(s->pin + 1, s->cnt_all.p, 8 * (n + 1) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                // sync 1 of 2: what goes where is a host decision
    rc = shares_plan(b, s->pin + 1);
    if (rc) return rc;
    for (int back = 0; back < 2; ++back) {
        NCCLCHK(c, rccl()->GroupStart());
        rc = group_end(c, shares_exchange_enqueue(b, back != 0));
        if (rc) return rc;
        rc = back ? shares_finish_enqueue(b) : shares_mark_enqueue(b);
        if (rc) return rc;
    }
    Exchange* x = exchange_of(c);
    HIPCHK(c, x->counts.ensure(8 * (n + 1)));
    rc = ensure_pin(c, x);
    if (rc) return rc;
    rc = sum_over_ranks_enqueue(c);
    if (rc) return rc;
    u64 sum = 0;
    rc = sum_over_ranks_finish(c, &sum);                       // sync 2 of 2
    if (rc) return rc;
    if (n_total) *n_total = s->n_total;
    if (n_unique) *n_unique = sum;
    if (first_global) *first_global = s->first[(size_t)c->comm_rank];
    return MI_OK;
}

// ... and of mi_dedup_allgather_all: n ctxs of one process.  The rows per owner are counted on the devices, so every rank is
// read once (no collective) before the two grouped all-to-alls.
int mi_dedup_alltoall_all(mi_batch** batches, int n, uint64_t* n_total, uint64_t* n_unique) {
    if (!batches || n < 1) return MI_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        if (!batches[i]) return MI_ERR_INVALID;
        mi_ctx* c = batches[i]->ctx;
        if (!c->comm || c->comm_nranks != n || c->comm_rank != i)
            return fail(c, MI_ERR_STATE, "batch %d does not belong to rank %d of an %d-rank mi_comm_init_all", i, i, n);
        if (!batches[i]->ran || batches[i]->in_flight)
            return fail(c, MI_ERR_STATE, "every batch must have run (and been waited for)");
    }
    mi_ctx* c0 = batches[0]->ctx;
    int rc = need_p2p(c0);
    if (rc) return rc;
    const size_t w = (size_t)n + 1;
    std::vector<u64> m(w * (size_t)n);
    for (int i = 0; i < n; ++i) {
        HIPCHK(batches[i]->ctx, hipSetDevice(batches[i]->ctx->device));
        rc = shares_split_enqueue(batches[i]);
        if (rc) return rc;
    }
    for (int i = 0; i < n; ++i) {
        mi_ctx* c = batches[i]->ctx;
        Shares* s = shares_of(c);
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipMemcpyAsync(s->pin + 1, s->cnt_send.p, 8 * w, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        memcpy(&m[(size_t)i * w], s->pin + 1, 8 * w);
    }
    for (int i = 0; i < n; ++i) {                              // allocations and uploads: outside the groups
        HIPCHK(batches[i]->ctx, hipSetDevice(batches[i]->ctx->device));
        rc = shares_plan(batches[i], m.data());
        if (rc) return rc;
    }
    for (int back = 0; back < 2; ++back) {
        NCCLCHK(c0, rccl()->GroupStart());
        rc = MI_OK;
        for (int i = 0; i < n && !rc; ++i) {
            (void)hipSetDevice(batches[i]->ctx->device);
            rc = shares_exchange_enqueue(batches[i], back != 0);
        }
        rc = group_end(c0, rc);
        if (rc) return rc;
        for (int i = 0; i < n; ++i) {
            (void)hipSetDevice(batches[i]->ctx->device);
            rc = back ? shares_finish_enqueue(batches[i]) : shares_mark_enqueue(batches[i]);
            if (rc) return rc;
        }
    }
    uint64_t unique_sum = 0;
    for (int i = 0; i < n; ++i) {                              // every owner counted the distinct digests it owns
        mi_ctx* c = batches[i]->ctx;
        (void)hipSetDevice(c->device);
        HIPCHK(c, hipMemcpyAsync(c->h_word, c->dd_nuniq.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        unique_sum += *c->h_word;
    }
    if (n_total) *n_total = shares_of(c0)->n_total;
    if (n_unique) *n_unique = unique_sum;
    return MI_OK;
}

}  // extern "C"
