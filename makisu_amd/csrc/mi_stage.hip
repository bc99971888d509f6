// mi_stage.hip -- host->HBM staging for host-fed batches: a pool of pinned slabs filled by
// reader threads (pread from the page cache, or memcpy from caller memory), one hipMemcpyAsync
// per filled slab, copies of different threads' slabs in flight on different streams.
//
// What it replaces: the byte loop of tario.WriteEntry (lib/tario/write.go:28-52: open, then
// io.CopyN(w, f, h.Size) in 32 KiB pieces on one goroutine).  Here the files of a batch are read
// by several threads at once, consecutive small files share one slab (one PCIe transfer per
// ~8 MiB, not per file), and the transfer of one thread's slab overlaps the other threads' reads.
// Host code only (no kernels); lives next to the engine because it owns HIP streams.
#include "mi_internal.h"

#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

using namespace mi;

namespace mi {

struct StageFile {                       // a source file shared by its pieces
    int fd;                              // open (mi_batch_add_path: checked at the call), or -1:
    std::string path;                    // ... the reader thread opens `path` for each piece itself
    ~StageFile() { if (fd >= 0) close(fd); }
};

struct StageLatch {                      // completion of one blocking mi_batch_add_bytes call
    std::mutex mu;
    std::condition_variable cv;
    u64 left = 0;
};

struct StageItem {
    mi_batch* batch;
    u64 arena_off, len;
    const u8* src;                       // caller memory (valid until its latch opens) or nullptr
    std::shared_ptr<StageFile> file;     // ... or a file range
    u64 file_off;
    StageLatch* latch;
};

struct Stager {
    mi_ctx* ctx;
    u64 slab_bytes;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<StageItem> queue;
    bool stop = false;
};

namespace {

void batch_fail(Stager* st, mi_batch* b, const std::string& msg) {
    std::lock_guard<std::mutex> g(st->mu);
    if (b->stage_err.empty()) b->stage_err = msg;
}

// the item's source bytes sit in a pinned slab: a blocking adder may return (cgo pointer rule)
void item_consumed(const StageItem& it) {
    if (!it.latch) return;
    std::lock_guard<std::mutex> g(it.latch->mu);
    if (--it.latch->left == 0) it.latch->cv.notify_all();
}

// the item's bytes are in HBM (or its batch has been marked failed)
void item_landed(Stager* st, const StageItem& it) {
    bool wake = false;
    {
        std::lock_guard<std::mutex> g(st->mu);
        wake = --it.batch->stage_pending == 0;
    }
    if (wake) st->cv_done.notify_all();
}

// One reader thread: pops a run of queued items whose arena span fits its slab, fills the slab
// (slab offset = arena offset - span start, so alignment gaps between files travel as they are),
// issues ONE H2D copy for the span on its own stream and waits for it; the other threads read and
// copy meanwhile, so PCIe stays busy without any cross-thread event hand-over.  A blocking
// mi_batch_add_bytes returns as soon as its pieces sit in slabs, before their transfers finish.
void worker(Stager* st) {
    mi_ctx* c = st->ctx;
    (void)hipSetDevice(c->device);
    hipStream_t stream = nullptr;
    void* slab = nullptr;
    bool ok = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipHostMalloc(&slab, st->slab_bytes, hipHostMallocDefault) == hipSuccess;
    std::vector<StageItem> run;
    for (;;) {
        run.clear();
        {
            std::unique_lock<std::mutex> lk(st->mu);
            st->cv_work.wait(lk, [&] { return st->stop || !st->queue.empty(); });
            if (st->queue.empty()) break;                      // stop requested and nothing left
            const mi_batch* b = st->queue.front().batch;
            const u64 start = st->queue.front().arena_off;
            u64 prev_end = start;
            while (!st->queue.empty()) {
                const StageItem& f = st->queue.front();
                if (f.batch != b || f.arena_off < start || f.arena_off + f.len - start > st->slab_bytes) break;
                // only alignment padding may lie between two items of a run: the span travels as ONE
                // copy, and a larger hole is somebody else's bytes (the batch's inline window, a file
                // another thread is staging) that this copy must not overwrite
                if (f.arena_off < prev_end || f.arena_off - prev_end >= kFileAlign) break;
                prev_end = f.arena_off + f.len;
                run.push_back(f);
                st->queue.pop_front();
            }
        }
        mi_batch* b = run.front().batch;
        const u64 start = run.front().arena_off;
        std::string err;
        if (!ok) err = "staging thread could not allocate its pinned slab";
        u64 end = start;
        for (const StageItem& it : run) {
            if (!err.empty()) break;
            u8* dst = (u8*)slab + (it.arena_off - start);
            if (it.arena_off > end) memset((u8*)slab + (end - start), 0, it.arena_off - end);   // alignment gap
            if (it.src) {
                memcpy(dst, it.src, it.len);
            } else {
                // a deferred file (bulk adds, tree walks) is opened HERE, by one of several threads
                int fd = it.file->fd, own = -1;
                if (fd < 0) {
                    own = fd = open(it.file->path.c_str(), O_RDONLY | O_CLOEXEC);
                    if (fd < 0) { err = "open " + it.file->path + ": " + strerror(errno); break; }
                }
                u64 got = 0;
                while (got < it.len) {
                    const ssize_t r = pread(fd, dst + got, it.len - got, (off_t)(it.file_off + got));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) {
                        err = "read " + it.file->path + ": " +
                              (r == 0 ? std::string("file shorter than the size given") : std::string(strerror(errno)));
                        break;
                    }
                    got += (u64)r;
                }
                if (own >= 0) close(own);
            }
            end = it.arena_off + it.len;
        }
        for (const StageItem& it : run) item_consumed(it);
        if (err.empty()) {
            hipError_t e = hipMemcpyAsync((u8*)b->arena.p + start, slab, end - start, hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) err = std::string("host-to-device copy (staging): ") + hipGetErrorString(e);
        }
        if (!err.empty()) batch_fail(st, b, err);
        for (const StageItem& it : run) item_landed(st, it);
    }
    if (slab) (void)hipHostFree(slab);
    if (stream) (void)hipStreamDestroy(stream);
}

}  // namespace

Stager* stager_create(mi_ctx* c, u32 n_threads, u64 slab_bytes) {
    Stager* st = new Stager();
    st->ctx = c;
    st->slab_bytes = slab_bytes;
    for (u32 i = 0; i < n_threads; ++i) st->threads.emplace_back(worker, st);
    return st;
}

void stager_destroy(Stager* st) {
    if (!st) return;
    {
        std::lock_guard<std::mutex> g(st->mu);
        st->stop = true;
    }
    st->cv_work.notify_all();
    for (auto& t : st->threads) t.join();
    delete st;
}

// Queues [arena_off, +len) of the batch's arena, split into pieces of at most one slab.
static void enqueue(Stager* st, mi_batch* b, u64 arena_off, u64 len, const u8* src,
                    const std::shared_ptr<StageFile>& file, u64 file_off, StageLatch* latch) {
    std::vector<StageItem> items;
    for (u64 done = 0; done < len;) {
        const u64 take = len - done < st->slab_bytes ? len - done : st->slab_bytes;
        items.push_back({b, arena_off + done, take, src ? src + done : nullptr, file, file_off + done, latch});
        done += take;
    }
    if (latch) latch->left = items.size();
    {
        std::lock_guard<std::mutex> g(st->mu);
        b->stage_pending += items.size();
        for (auto& it : items) st->queue.push_back(std::move(it));
    }
    if (items.size() == 1) st->cv_work.notify_one();      // one small file: one reader, not the whole pool
    else st->cv_work.notify_all();
}

int stager_put_bytes(Stager* st, mi_batch* b, u64 arena_off, const void* src, u64 len) {
    if (len == 0) return MI_OK;
    StageLatch latch;
    enqueue(st, b, arena_off, len, (const u8*)src, nullptr, 0, &latch);
    std::unique_lock<std::mutex> lk(latch.mu);
    latch.cv.wait(lk, [&] { return latch.left == 0; });          // the engine never retains caller memory
    return MI_OK;
}

int stager_put_file(Stager* st, mi_batch* b, u64 arena_off, int fd, u64 file_off, u64 len, const char* path) {
    auto f = std::make_shared<StageFile>();
    f->fd = fd;
    f->path = path ? path : "";
    if (len) enqueue(st, b, arena_off, len, nullptr, f, file_off, nullptr);
    return MI_OK;
}

// n whole files that the reader threads open themselves: one lock, one wake-up for all of them
int stager_put_paths(Stager* st, mi_batch* b, u64 n, const char* const* paths, const u64* arena_off,
                     const u64* len) {
    std::vector<StageItem> items;
    items.reserve(n);
    for (u64 i = 0; i < n; ++i) {
        if (len[i] == 0) continue;
        auto f = std::make_shared<StageFile>();
        f->fd = -1;
        f->path = paths[i];
        for (u64 done = 0; done < len[i];) {
            const u64 take = len[i] - done < st->slab_bytes ? len[i] - done : st->slab_bytes;
            items.push_back({b, arena_off[i] + done, take, nullptr, f, done, nullptr});
            done += take;
        }
    }
    if (items.empty()) return MI_OK;
    {
        std::lock_guard<std::mutex> g(st->mu);
        b->stage_pending += items.size();
        for (auto& it : items) st->queue.push_back(std::move(it));
    }
    st->cv_work.notify_all();
    return MI_OK;
}

// Blocks until every byte queued for the batch has landed in HBM; returns the first staging error.
int stager_drain(Stager* st, mi_batch* b) {
    {
        std::unique_lock<std::mutex> lk(st->mu);
        st->cv_done.wait(lk, [&] { return b->stage_pending == 0; });
    }
    if (!b->stage_err.empty()) {
        const std::string msg = b->stage_err;
        b->stage_err.clear();
        return fail(b->ctx, MI_ERR_IO, "%s", msg.c_str());
    }
    return MI_OK;
}

}  // namespace mi
