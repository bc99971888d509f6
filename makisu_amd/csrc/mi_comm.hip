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

// per-exchange scratch kept on the ctx side of things (one exchange at a time per ctx)
struct Exchange {
    DevBuf counts, slab, gathered, compact, dup;
    u64* pin = nullptr;                 // pinned: [0] this rank's scalar, [1 .. nranks] gathered scalars
    size_t pin_words = 0;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // before the slab all-gather / after it / after the marking
    bool timed = false;                 // the three events of the last exchange were recorded
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

}  // extern "C"
