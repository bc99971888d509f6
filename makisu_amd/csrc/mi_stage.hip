// mi_stage.hip -- host->HBM staging for host-fed batches: a pool of pinned slabs filled by
// reader threads (pread from the page cache, or memcpy from caller memory), one hipMemcpyAsync
// per filled slab, copies of different threads' slabs in flight on different streams.
//
// What it replaces: the byte loop of tario.WriteEntry (lib/tario/write.go:28-52: open, then
// io.CopyN(w, f, h.Size) in 32 KiB pieces on one goroutine).  Here the files of a batch are read
// by several threads at once, consecutive small files share one slab (one PCIe transfer per
// ~8 MiB, not per file), and the transfer of one thread's slab overlaps the other threads' reads.
// MI_FLAG_VERIFY_STAGING: every copy is summed on the host (in the slab) and on the GPU (where it
// landed) -- stage_sum_kernel below, the one kernel of this file --, copied again if the sums
// differ, and summed once more when staging ends (stage_verify_final).
// A file that is handed over as a path is opened ONCE, by the reader that takes its first piece; the other pieces read
// through the same descriptor, the last one closes it.  A batch's pieces are queued in arena order, so "everything below
// arena offset X has landed" is one number (stager_wait_landed): what lets the commit's tar writer read a file out of HBM
// while files behind it are still on their way.
#include "mi_internal.h"
#include "mi_hostpath.h"      // mi_io: what was read of file content
#include "host_sha256.h"      // strings too long for a GPU lane: a SHA-NI stream per reader thread

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

using namespace mi;

namespace mi {

struct StageFile {                       // a source file shared by its pieces
    int fd;                              // open (mi_batch_add_path: checked at the call), or -1:
    std::string path;                    // ... the reader thread that takes the file's FIRST piece opens `path`, the others
    std::mutex mu;                       // read through the same descriptor: one open per file, and every piece of a file
    u64 pieces_left = 0;                 // that is renamed over under way comes from the same inode.  Closed with the LAST
    ~StageFile() { if (fd >= 0) close(fd); }   // piece read (under mu), not when the last item lets go: a run of small files
    void piece_read() {                  // must not hold a descriptor per file until its copy is done
        std::lock_guard<std::mutex> g(mu);
        if (pieces_left && --pieces_left == 0 && fd >= 0) { close(fd); fd = -1; }
    }
};

struct StageLatch {                      // completion of one blocking mi_batch_add_bytes call
    std::mutex mu;
    std::condition_variable cv;
    u64 left = 0;
};

struct StageItem {
    mi_batch* batch;
    u64 arena_off, len;
    const u8* src;                       // caller memory (valid until its latch opens), memory `keep` owns, or nullptr
    std::shared_ptr<StageFile> file;     // ... or a file range
    u64 file_off;
    StageLatch* latch;
    std::shared_ptr<void> keep;          // owner of `src` for blocks handed over for good (stager_put_block)
    mi_sum::FileSum* sums = nullptr;     // the file row's chunk sums (mi_filesum.h), or nullptr; this piece is bytes
    u64 row_off = 0;                     // [row_off, row_off + len) of that row
};

// Files per run: a slab of tiny files would otherwise be ONE thread's work for milliseconds (2 048
// 4 KiB files at ~4 us of open + pread + close each) while the others idle at the end of a batch.
constexpr size_t kMaxRunItems = 256;

struct HashLatch {                       // completion of one stager_hash_ranges call
    std::mutex mu;
    std::condition_variable cv;
    u64 left = 0;
    std::string err;
};
struct HashJob { mi_batch* batch; u64 arena_off, len; u8* out; HashLatch* latch; };

struct Stager {
    mi_ctx* ctx;
    u64 slab_bytes;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<StageItem> queue;
    std::deque<HashJob> hash_queue;      // whole strings to hash out of HBM (stager_hash_ranges): taken when no copy is waiting
    bool stop = false;
    u32 init_left = 0, n_ok = 0;         // reader threads still setting up / that got slab + stream
    std::atomic<long long> spans_copied{0};   // fault injection (MI_STAGE_FAULT) counts spans with it
    std::condition_variable cv_init;
    int pausers = 0;                     // threads in stager_pause (an adder waiting for descriptors to come back)
};

// ---- span sums (MI_FLAG_VERIFY_STAGING) --------------------------------------------------------
void stage_sum_host(const void* p, u64 len, u64* s1_out, u64* s2_out) {
    const u8* b = (const u8*)p;
    const u64 n = len / 8;
    u64 s1 = 0, s2 = 0;
    for (u64 i = 0; i < n; ++i) {
        u64 w;
        memcpy(&w, b + 8 * i, 8);
        s1 += w;
        s2 += s1;
    }
    if (len & 7) {
        u64 w = 0;
        memcpy(&w, b + 8 * n, len & 7);
        s1 += w;
        s2 += s1;
    }
    *s1_out = s1;
    *s2_out = s2;
}

constexpr u32 kSumTile = 65536;          // bytes one workgroup sums

// grid (spans, tiles): workgroup (i, t) sums words [8192 t, +8192) of span i (striding over t when a
// span has more tiles than the grid is high); 16-byte loads, a wave touches 1 KiB per instruction
__global__ __launch_bounds__(256)
void stage_sum_kernel(const u8* __restrict__ base, const u64* __restrict__ off, const u64* __restrict__ len,
                      u64 n_spans, u64* __restrict__ out) {
    const u64 i = blockIdx.x;
    if (i >= n_spans) return;
    const u64 L = len[i];
    const u64 n_words = (L + 7) / 8;                     // the tail word is zero-padded
    const u64 full = L / 8;
    const u8* p = base + off[i];
    u64 s1 = 0, s2 = 0;
    for (u64 t = blockIdx.y; t * (kSumTile / 8) < n_words; t += gridDim.y) {
        const u64 w0 = t * (kSumTile / 8);
#pragma unroll 4
        for (u32 k = 0; k < kSumTile / 16 / 256; ++k) {
            const u64 idx = w0 + 2ull * (k * 256u + threadIdx.x);   // my two words
            if (idx >= n_words) break;
            u64 a = 0, b = 0;
            if (idx + 1 < full) {
                const ulonglong2 v = *(const ulonglong2*)(p + 8 * idx);
                a = v.x;
                b = v.y;
            } else {                                       // the span's last words, byte-wise past `full`
                for (u32 j = 0; j < 16; ++j) {
                    const u64 at = 8 * idx + j;
                    if (at < L) {
                        const u64 v = p[at];
                        if (j < 8) a |= v << (8 * j); else b |= v << (8 * (j - 8));
                    }
                }
            }
            s1 += a;
            s2 += (n_words - idx) * a;
            if (idx + 1 < n_words) { s1 += b; s2 += (n_words - idx - 1) * b; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s1 += __shfl_xor(s1, d);
        s2 += __shfl_xor(s2, d);
    }
    __shared__ u64 part[2][4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][wave] = s1; part[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 a = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        const u64 b = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        if (a) atomicAdd((unsigned long long*)&out[2 * i], (unsigned long long)a);
        if (b) atomicAdd((unsigned long long*)&out[2 * i + 1], (unsigned long long)b);
    }
}

void launch_stage_sums(const u8* base, const u64* d_off, const u64* d_len, u64 n, u64 max_len, u64* d_out,
                       hipStream_t s) {
    if (n == 0) return;
    (void)hipMemsetAsync(d_out, 0, n * 16, s);
    u64 tiles = (max_len + kSumTile - 1) / kSumTile;
    if (tiles == 0) tiles = 1;
    if (tiles > 1024) tiles = 1024;
    hipLaunchKernelGGL(stage_sum_kernel, dim3((u32)n, (u32)tiles), dim3(256), 0, s, base, d_off, d_len, n, d_out);
}

namespace {

// the batch's own record of what it has queued (under the stager's mutex): pieces leave the queue in the order they came
void f_batch_push(const StageItem& it) {
    mi_batch* b = it.batch;
    if (!b->stage_queued.empty() && it.arena_off < b->stage_queued.back()) b->stage_unordered = true;
    b->stage_queued.push_back(it.arena_off);
}
void f_batch_pop(const StageItem& it) {
    mi_batch* b = it.batch;
    if (!b->stage_queued.empty()) b->stage_queued.pop_front();
}

// the item's source bytes sit in a pinned slab: a blocking adder may return (cgo pointer rule)
void item_consumed(const StageItem& it) {
    if (!it.latch) return;
    std::lock_guard<std::mutex> g(it.latch->mu);
    if (--it.latch->left == 0) it.latch->cv.notify_all();
}

// what the GPU holds in [dev, dev+len) next to what it should hold: one line for the error message
static std::string describe_span(const u8* dev, const u8* want, u64 len, hipStream_t stream) {
    std::vector<u8> got(len);
    if (hipMemcpyAsync(got.data(), dev, len, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return "(read-back failed)";
    u64 first = ~0ull, last = 0, n = 0;
    bool zeros = true, poison = true;
    for (u64 i = 0; i < len; ++i) {
        const bool differs = want ? got[i] != want[i] : true;
        if (!differs) continue;
        if (first == ~0ull) first = i;
        last = i;
        ++n;
        zeros = zeros && got[i] == 0;
        poison = poison && got[i] == 0xA5;
    }
    char buf[256];
    if (want)
        snprintf(buf, sizeof buf, "%llu byte(s) differ in [+%llu, +%llu], the GPU holds %s there",
                 (unsigned long long)n, (unsigned long long)first, (unsigned long long)last,
                 n == 0 ? "the right bytes (the sums raced a later write?)" : zeros ? "zeros" :
                 poison ? "the 0xA5 fill of new arena memory (the copy never landed)" : "other data");
    else
        snprintf(buf, sizeof buf, "the GPU holds %s there",
                 zeros ? "zeros" : poison ? "the 0xA5 fill of new arena memory" : "other data");
    return buf;
}

// One string out of HBM through a reader's slab, in two halves: the next piece is on its way while this one is hashed
// (a piece of 4 MiB: 0.15 ms of PCIe against 1.8 ms of SHA-NI)
void hash_range(const HashJob& job, u8* slab, u64 slab_bytes, hipStream_t stream) {
    const u64 half = (slab_bytes / 2) & ~(u64)4095;
    const u8* src = (const u8*)job.batch->arena.p + job.arena_off;
    mi_host::Sha256 sha;
    std::string err;
    auto fetch = [&](u64 at, int h) -> u64 {
        const u64 n = job.len - at < half ? job.len - at : half;
        if (n && err.empty()) {
            const hipError_t e = hipMemcpyAsync(slab + h * half, src + at, n, hipMemcpyDeviceToHost, stream);
            if (e != hipSuccess) err = std::string("device-to-host copy (hashing a long string): ") + hipGetErrorString(e);
        }
        return n;
    };
    u64 at = 0;
    int h = 0;
    u64 n = fetch(0, 0);
    while (n && err.empty()) {
        const hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { err = std::string("device-to-host copy (hashing a long string): ") + hipGetErrorString(e); break; }
        const u64 next = fetch(at + n, h ^ 1);
        sha.update(slab + h * half, (size_t)n);
        at += n;
        n = next;
        h ^= 1;
    }
    if (err.empty()) sha.final(job.out);
    std::lock_guard<std::mutex> g(job.latch->mu);
    if (!err.empty() && job.latch->err.empty()) job.latch->err = err;
    if (--job.latch->left == 0) job.latch->cv.notify_all();
}

// One reader thread: pops a run of queued items whose arena span fits its slab, fills the slab
// (slab offset = arena offset - span start, so alignment gaps between files travel as they are),
// issues ONE H2D copy for the span on its own stream and waits for it; the other threads read and
// copy meanwhile, so PCIe stays busy without any cross-thread event hand-over.  A blocking
// mi_batch_add_bytes returns as soon as its pieces sit in slabs, before their transfers finish.
void worker(Stager* st, u32 tid) {
    mi_ctx* c = st->ctx;
    (void)hipSetDevice(c->device);
    hipStream_t stream = nullptr;
    void* slab = nullptr;
    u64* d_sums = nullptr;                                   // {off, len, s1, s2} on the device
    u64* h_sums = nullptr;                                   // ... and pinned: [0..1] in, [2..3] out
    bool ok = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipHostMalloc(&slab, st->slab_bytes, hipHostMallocDefault) == hipSuccess;
    if (c->verify_staging) {
        ok = ok && hipMalloc((void**)&d_sums, 32) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&h_sums, 32, hipHostMallocDefault) == hipSuccess;
    }
    {
        std::lock_guard<std::mutex> g(st->mu);
        if (ok) ++st->n_ok;
        --st->init_left;
    }
    st->cv_init.notify_all();
    std::vector<StageItem> run;
    while (ok) {                                             // a thread without slab or stream takes no work
        run.clear();
        {
            std::unique_lock<std::mutex> lk(st->mu);
            st->cv_work.wait(lk, [&] { return st->stop || !st->queue.empty() || !st->hash_queue.empty(); });
            if (st->queue.empty() && !st->hash_queue.empty()) {
                const HashJob job = st->hash_queue.front();
                st->hash_queue.pop_front();
                lk.unlock();
                hash_range(job, (u8*)slab, st->slab_bytes, stream);
                continue;
            }
            if (st->queue.empty()) break;                      // stop requested and nothing left
            const mi_batch* b = st->queue.front().batch;
            const u64 start = st->queue.front().arena_off;
            u64 prev_end = start;
            while (!st->queue.empty() && run.size() < kMaxRunItems) {
                const StageItem& f = st->queue.front();
                if (f.batch != b || f.arena_off < start || f.arena_off + f.len - start > st->slab_bytes) break;
                // only alignment padding may lie between two items of a run: the span travels as ONE
                // copy, and a larger hole is somebody else's bytes (the batch's inline window, a file
                // another thread is staging) that this copy must not overwrite
                if (f.arena_off < prev_end || f.arena_off - prev_end >= kFileAlign) break;
                prev_end = f.arena_off + f.len;
                run.push_back(f);
                st->queue.pop_front();
                f_batch_pop(run.back());
            }
            run.front().batch->stage_inflight.insert(start);    // (taken back when the run has landed: stager_wait_landed)
        }
        mi_batch* b = run.front().batch;
        const u64 start = run.front().arena_off;
        std::string err;
        u64 end = start;
        for (const StageItem& it : run) {
            if (!err.empty()) break;
            u8* dst = (u8*)slab + (it.arena_off - start);
            if (it.arena_off > end) memset((u8*)slab + (end - start), 0, it.arena_off - end);   // alignment gap
            if (it.src) {
                memcpy(dst, it.src, it.len);
                if (it.sums) mi_sum::row_add(dst, it.len, it.row_off, it.sums);
            } else {
                // a deferred file (bulk adds, tree walks) is opened HERE, by one of several threads
                int fd;
                {
                    std::lock_guard<std::mutex> g(it.file->mu);
                    if (it.file->fd < 0) {
                        it.file->fd = open(it.file->path.c_str(), O_RDONLY | O_CLOEXEC);   // closed with the file's last piece
                        if (it.file->fd >= 0) mi_io::content_opens.fetch_add(1, std::memory_order_relaxed);
                    }
                    fd = it.file->fd;
                }
                if (fd < 0) { err = "open " + it.file->path + ": " + strerror(errno); break; }
                mi_io::content_bytes.fetch_add(it.len, std::memory_order_relaxed);
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
                it.file->piece_read();
                // the sums of the bytes as read() delivered them, taken in the pinned slab before the DMA sees it: what the
                // layer writer will hold against the bytes that come back from HBM
                if (it.sums && err.empty()) mi_sum::row_add(dst, it.len, it.row_off, it.sums);
            }
            end = it.arena_off + it.len;
        }
        for (const StageItem& it : run) item_consumed(it);
        const u64 span = end - start;
        StageSpan sp{start, span, 0, 0, tid};
        double ms_verify = 0;
        u64 mism = 0, repaired = 0;
        std::string note;
        if (err.empty() && span) {                               // the arena's memory behind this span: mapped by now, nearly always
            std::string m;
            if (arena_wait_mapped(c, &b->arena, start + span, &m) != MI_OK) err = "device memory for the staged bytes: " + m;
        }
        if (err.empty() && span) {
            u8* dev = (u8*)b->arena.p + start;
            auto copy = [&]() -> hipError_t {
                hipError_t e = hipMemcpyAsync(dev, slab, span, hipMemcpyHostToDevice, stream);
                return e == hipSuccess ? hipStreamSynchronize(stream) : e;
            };
            // device sums of the span as it lies in HBM now
            auto device_sums = [&](u64* s1, u64* s2) -> hipError_t {
                h_sums[0] = start;
                h_sums[1] = span;
                hipError_t e = hipMemcpyAsync(d_sums, h_sums, 16, hipMemcpyHostToDevice, stream);
                if (e != hipSuccess) return e;
                launch_stage_sums((const u8*)b->arena.p, d_sums, d_sums + 1, 1, span, d_sums + 2, stream);
                e = hipMemcpyAsync(h_sums + 2, d_sums + 2, 16, hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
                *s1 = h_sums[2];
                *s2 = h_sums[3];
                return e;
            };
            hipError_t e = hipSuccess;
            if (c->verify_staging) {
                const auto t0 = std::chrono::steady_clock::now();
                stage_sum_host(slab, span, &sp.s1, &sp.s2);     // before the copy: while it runs the slab is the DMA's
                ms_verify += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            e = copy();
            if (e == hipSuccess && st->spans_copied.fetch_add(1) == c->fault_copy) {      // tests only
                (void)hipMemsetAsync(dev + span / 2, 0, span / 2 < 4096 ? span / 2 : 4096, stream);
                (void)hipStreamSynchronize(stream);
            }
            if (e == hipSuccess && c->verify_staging) {
                const auto t0 = std::chrono::steady_clock::now();
                u64 d1 = 0, d2 = 0;
                e = device_sums(&d1, &d2);
                if (e == hipSuccess && (d1 != sp.s1 || d2 != sp.s2)) {
                    mism = 1;
                    char head[256];
                    snprintf(head, sizeof head, "staging verify: arena [%llu, +%llu) of batch %p, reader thread %u: "
                             "host sums %016llx/%016llx, device %016llx/%016llx; ", (unsigned long long)start,
                             (unsigned long long)span, (void*)b, tid, (unsigned long long)sp.s1,
                             (unsigned long long)sp.s2, (unsigned long long)d1, (unsigned long long)d2);
                    note = head + describe_span(dev, (const u8*)slab, span, stream);
                    e = copy();                                 // the slab still holds the bytes: once more
                    if (e == hipSuccess) e = device_sums(&d1, &d2);
                    if (e == hipSuccess) {
                        if (d1 == sp.s1 && d2 == sp.s2) { repaired = 1; note += "; a second copy from the slab matched"; }
                        else err = note + "; a second copy from the slab did not match either";
                    }
                }
                ms_verify += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            if (e != hipSuccess) err = std::string("host-to-device copy (staging): ") + hipGetErrorString(e);
        }
        if (err.empty() && span) {
            std::lock_guard<std::mutex> g(b->span_mu);
            mi_stage_stats& ss = b->stage_stats;
            ++ss.spans;
            ss.bytes += span;
            if (c->verify_staging) {
                ++ss.verified_spans;
                ss.mismatches += mism;
                ss.repaired += repaired;
                ss.ms_verify += ms_verify;
                b->stage_spans.push_back(sp);
                if (mism && b->stage_note.empty()) b->stage_note = note;
                if (mism) fprintf(stderr, "makisu_mi: %s\n", note.c_str());    // rare and worth a line even when repaired
            }
        }
        if (!err.empty()) {
            std::lock_guard<std::mutex> g(st->mu);
            if (b->stage_err.empty()) b->stage_err = err;
        }
        {                                                      // the run's bytes are in HBM (or its batch is marked failed)
            bool wake;
            {
                std::lock_guard<std::mutex> g(st->mu);
                b->stage_inflight.erase(b->stage_inflight.find(start));
                b->stage_pending -= run.size();
                wake = b->stage_pending == 0 || b->stage_waiters > 0 || st->pausers > 0;
            }
            if (wake) st->cv_done.notify_all();
        }
    }
    if (slab) (void)hipHostFree(slab);
    if (d_sums) (void)hipFree(d_sums);
    if (h_sums) (void)hipHostFree(h_sums);
    if (stream) (void)hipStreamDestroy(stream);
}

}  // namespace

// The reader threads start setting up (a pinned slab and a copy stream each: milliseconds) and the call returns;
// stager_ready waits for them -- whoever knows early that host-fed bytes are coming (the tree walk) starts the
// readers first and meets them when the first bytes are there.
Stager* stager_create(mi_ctx* c, u32 n_threads, u64 slab_bytes) {
    Stager* st = new Stager();
    st->ctx = c;
    st->slab_bytes = slab_bytes;
    st->init_left = n_threads;
    for (u32 i = 0; i < n_threads; ++i) st->threads.emplace_back(worker, st, i);
    return st;
}
// true when at least one reader thread got its slab and stream (false: nobody can take work -- say so now, not per batch).
// Returns as soon as the FIRST reader is up: the others join in as they come (a pinned slab takes ~5 ms to allocate and the
// driver hands them out one after the other -- waiting for all eight kept a cold ctx's first walk 40 ms from its first byte).
bool stager_ready(Stager* st) {
    std::unique_lock<std::mutex> lk(st->mu);
    st->cv_init.wait(lk, [&] { return st->n_ok != 0 || st->init_left == 0; });
    return st->n_ok != 0;
}

// ... and when EVERY reader thread is up or has given up (mi_ctx_warm: nothing of it is left for the first walk)
bool stager_ready_all(Stager* st) {
    std::unique_lock<std::mutex> lk(st->mu);
    st->cv_init.wait(lk, [&] { return st->init_left == 0; });
    return st->n_ok != 0;
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
                    const std::shared_ptr<StageFile>& file, u64 file_off, StageLatch* latch, mi_sum::FileSum* sums, u64 row_off0 = 0) {
    std::vector<StageItem> items;
    for (u64 done = 0; done < len;) {
        const u64 take = len - done < st->slab_bytes ? len - done : st->slab_bytes;
        items.push_back({b, arena_off + done, take, src ? src + done : nullptr, file, file_off + done, latch, nullptr, sums, row_off0 + done});
        done += take;
    }
    if (latch) latch->left = items.size();
    if (file) file->pieces_left = items.size();
    {
        std::lock_guard<std::mutex> g(st->mu);
        b->stage_pending += items.size();
        for (auto& it : items) { f_batch_push(it); st->queue.push_back(std::move(it)); }
    }
    if (items.size() == 1) st->cv_work.notify_one();      // one small file: one reader, not the whole pool
    else st->cv_work.notify_all();
}

int stager_put_bytes(Stager* st, mi_batch* b, u64 arena_off, const void* src, u64 len, mi_sum::FileSum* sums) {
    if (len == 0) return MI_OK;
    StageLatch latch;
    enqueue(st, b, arena_off, len, (const u8*)src, nullptr, 0, &latch, sums);
    std::unique_lock<std::mutex> lk(latch.mu);
    latch.cv.wait(lk, [&] { return latch.left == 0; });          // the engine never retains caller memory
    return MI_OK;
}

int stager_put_file(Stager* st, mi_batch* b, u64 arena_off, int fd, u64 file_off, u64 len, const char* path, mi_sum::FileSum* sums,
                    u64 row_off0) {
    auto f = std::make_shared<StageFile>();
    f->fd = fd;
    f->path = path ? path : "";
    if (len) enqueue(st, b, arena_off, len, nullptr, f, file_off, nullptr, sums, row_off0);
    return MI_OK;
}

// A block of host memory that already holds the bytes of [arena_off, +len) -- the files a directory reader of the tree
// walk read where it listed them (mi_tree.hip), laid out as they lie in the arena.  `keep` owns the block; the call
// returns at once, the block is let go of when its last piece sits in a slab.
int stager_put_block(Stager* st, mi_batch* b, u64 arena_off, const void* src, u64 len, std::shared_ptr<void> keep) {
    if (len == 0) return MI_OK;
    std::vector<StageItem> items;
    for (u64 done = 0; done < len;) {
        const u64 take = len - done < st->slab_bytes ? len - done : st->slab_bytes;
        items.push_back({b, arena_off + done, take, (const u8*)src + done, nullptr, 0, nullptr, keep});
        done += take;
    }
    {
        std::lock_guard<std::mutex> g(st->mu);
        b->stage_pending += items.size();
        for (auto& it : items) { f_batch_push(it); st->queue.push_back(std::move(it)); }
    }
    if (len > st->slab_bytes) st->cv_work.notify_all();        // several pieces: several readers
    else st->cv_work.notify_one();
    return MI_OK;
}

// n whole files that the reader threads open themselves: one lock, one wake-up for all of them
int stager_put_paths(Stager* st, mi_batch* b, u64 n, const char* const* paths, const u64* arena_off,
                     const u64* len, mi_sum::FileSum* const* sums) {
    std::vector<StageItem> items;
    items.reserve(n);
    for (u64 i = 0; i < n; ++i) {
        if (len[i] == 0) continue;
        auto f = std::make_shared<StageFile>();
        f->fd = -1;
        f->path = paths[i];
        for (u64 done = 0; done < len[i];) {
            const u64 take = len[i] - done < st->slab_bytes ? len[i] - done : st->slab_bytes;
            items.push_back({b, arena_off[i] + done, take, nullptr, f, done, nullptr, nullptr, sums ? sums[i] : nullptr, done});
            done += take;
            ++f->pieces_left;
        }
    }
    if (items.empty()) return MI_OK;
    {
        std::lock_guard<std::mutex> g(st->mu);
        b->stage_pending += items.size();
        for (auto& it : items) { f_batch_push(it); st->queue.push_back(std::move(it)); }
    }
    st->cv_work.notify_all();
    return MI_OK;
}

// Blocks until every byte queued for the batch has landed in HBM; returns the batch's first staging
// error -- STICKY: the message stays with the batch (every later run / submit / add fails the same
// way) until mi_batch_reset; a failed batch must never be scanned over half-staged bytes.
int stager_drain(Stager* st, mi_batch* b) {
    std::string msg;
    {
        std::unique_lock<std::mutex> lk(st->mu);
        st->cv_done.wait(lk, [&] { return b->stage_pending == 0; });
        msg = b->stage_err;
    }
    if (!msg.empty()) return fail(b->ctx, MI_ERR_IO, "%s", msg.c_str());
    return MI_OK;
}

void stager_pause(Stager* st, int ms) {
    std::unique_lock<std::mutex> lk(st->mu);
    ++st->pausers;
    st->cv_done.wait_for(lk, std::chrono::milliseconds(ms));
    --st->pausers;
}

// everything of the batch below this arena offset has landed (~0: everything queued so far); under the stager's mutex.
// The batch keeps the offsets of its queued pieces itself (stage_queued): no walk over the ctx's queue, in which another
// batch may have thousands of pieces, under the lock every reader thread needs.
static u64 landed_upto(Stager* st, mi_batch* b) {
    u64 m = b->stage_inflight.empty() ? ~0ull : *b->stage_inflight.begin();
    if (b->stage_unordered) {                                  // (an adder that did not enqueue in arena order: the slow, safe way)
        for (const StageItem& it : st->queue) if (it.batch == b && it.arena_off < m) m = it.arena_off;
    } else if (!b->stage_queued.empty() && b->stage_queued.front() < m) {
        m = b->stage_queued.front();
    }
    return m;
}

int stager_wait_landed(Stager* st, mi_batch* b, u64 upto, u64* landed_out) {
    std::string msg;
    {
        std::unique_lock<std::mutex> lk(st->mu);
        ++b->stage_waiters;
        st->cv_done.wait(lk, [&] { return !b->stage_err.empty() || landed_upto(st, b) >= upto; });
        --b->stage_waiters;
        msg = b->stage_err;
        if (landed_out) *landed_out = landed_upto(st, b);
    }
    if (!msg.empty()) return fail(b->ctx, MI_ERR_IO, "%s", msg.c_str());
    return MI_OK;
}

u64 stager_landed(Stager* st, mi_batch* b) {
    std::lock_guard<std::mutex> g(st->mu);
    if (!b->stage_err.empty()) return 0;                       // a batch whose staging failed has NOTHING a prefetch may read: its readers
    return landed_upto(st, b);                                 // gave their runs back without landing them (found by failing the mapper
}                                                              // under the tar writer on the double: a read of arena memory never mapped)

HashLatch* stager_hash_ranges(Stager* st, mi_batch* b, u64 n, const u64* arena_off, const u64* len, u8* out32) {
    HashLatch* latch = new HashLatch();
    latch->left = n;
    if (n == 0) return latch;
    {
        std::lock_guard<std::mutex> g(st->mu);
        for (u64 i = 0; i < n; ++i) st->hash_queue.push_back({b, arena_off[i], len[i], out32 + 32 * i, latch});
    }
    st->cv_work.notify_all();
    return latch;
}

int stager_hash_wait(mi_ctx* c, HashLatch* latch) {
    if (!latch) return MI_OK;
    std::string err;
    {
        std::unique_lock<std::mutex> lk(latch->mu);
        latch->cv.wait(lk, [&] { return latch->left == 0; });
        err = latch->err;
    }
    delete latch;
    if (!err.empty()) return fail(c, MI_ERR_HIP, "%s", err.c_str());
    return MI_OK;
}

void route_long_strings(const u64* lens, u64 n, u32 host_threads, bool h2d, std::vector<u32>* to_host) {
    to_host->clear();
    static const bool on_gpu = [] { const char* e = getenv("MI_SHA_LONG_ON_GPU"); return e && *e == '1'; }();   // (tests, A/B: every
    if (n == 0 || host_threads == 0 || on_gpu) return;                                                          //  string on a lane)
    constexpr double kLaneBps = 13.5e6, kRoofBps = 1.77e12, kPcieBps = 50e9;
    const double host_bps = mi_host::Sha256::have_shani() ? 2.2e9 : 0.35e9;
    u64 lmax = 0, total = 0;
    for (u64 i = 0; i < n; ++i) { total += lens[i]; if (lens[i] > lmax) lmax = lens[i]; }
    auto gpu_time = [&](u64 longest, u64 bytes) {                     // a pass over `bytes` whose longest string is `longest`
        const double lane = longest / kLaneBps, roof = bytes / kRoofBps;
        return (lane > roof ? lane : roof) + (h2d ? bytes / kPcieBps : 0.0);
    };
    // the common case, without a sort: the pass is not held up by its longest string (by more than a launch's own few ms)
    if (lmax / kLaneBps <= 2.0 * (total / kRoofBps) || lmax / kLaneBps < 5e-3) return;
    std::vector<u32> idx(n);
    for (u64 i = 0; i < n; ++i) idx[i] = (u32)i;
    std::sort(idx.begin(), idx.end(), [&](u32 x, u32 y) { return lens[x] != lens[y] ? lens[x] > lens[y] : x < y; });
    double best = gpu_time(lmax, total);
    u64 best_k = 0, moved = 0;
    for (u64 k = 0; k < n;) {
        u64 j = k;
        while (j < n && lens[idx[j]] == lens[idx[k]]) moved += lens[idx[j++]];      // the whole class of this length
        const u32 streams = host_threads < j ? host_threads : (u32)j;
        double host = moved / (host_bps * streams);
        if (lmax / host_bps > host) host = lmax / host_bps;                         // (one stream is one core)
        const double gpu = j < n ? gpu_time(lens[idx[j]], total - moved) : 0.0;
        const double t = host > gpu ? host : gpu;
        if (t < best) { best = t; best_k = j; }
        if (host > best) break;                                                      // the host side alone is already longer
        k = j;
    }
    to_host->assign(idx.begin(), idx.begin() + best_k);
}

// MI_FLAG_VERIFY_STAGING, when staging ends: every span this batch ever copied is summed again
// where it lies NOW -- after all readers drained and after any arena growth moved it -- in one
// launch, and compared with the sums taken in the pinned slab.
int stage_verify_final(mi_batch* b) {
    mi_ctx* c = b->ctx;
    std::vector<StageSpan> spans;
    {
        std::lock_guard<std::mutex> g(b->span_mu);
        spans = b->stage_spans;
    }
    const u64 n = spans.size();
    if (n == 0) return MI_OK;
    if (c->fault_final >= 0 && (u64)c->fault_final < n) {              // tests only
        const StageSpan& v = spans[c->fault_final];
        HIPCHK(c, hipMemsetAsync(b->arena.as<u8>() + v.off + v.len / 2, 0, v.len / 2 < 4096 ? v.len / 2 : 4096, c->stream));
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<u64> off(n), len(n), sums(2 * n);
    u64 max_len = 0;
    for (u64 i = 0; i < n; ++i) {
        off[i] = spans[i].off;
        len[i] = spans[i].len;
        max_len = len[i] > max_len ? len[i] : max_len;
    }
    HIPCHK(c, b->span_off.ensure(n * 8));
    HIPCHK(c, b->span_len.ensure(n * 8));
    HIPCHK(c, b->span_sums.ensure(n * 16));
    HIPCHK(c, hipMemcpyAsync(b->span_off.p, off.data(), n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b->span_len.p, len.data(), n * 8, hipMemcpyHostToDevice, c->stream));
    launch_stage_sums(b->arena.as<u8>(), b->span_off.as<u64>(), b->span_len.as<u64>(), n, max_len,
                      b->span_sums.as<u64>(), c->stream);
    HIPCHK(c, hipMemcpyAsync(sums.data(), b->span_sums.p, n * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    u64 bad = 0, first_bad = 0;
    for (u64 i = 0; i < n; ++i)
        if (sums[2 * i] != spans[i].s1 || sums[2 * i + 1] != spans[i].s2) {
            if (!bad) first_bad = i;
            ++bad;
        }
    {
        std::lock_guard<std::mutex> g(b->span_mu);
        b->stage_stats.final_spans += n;
        b->stage_stats.final_mismatches += bad;
        b->stage_stats.ms_verify += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (!bad) return MI_OK;
    const StageSpan& sp = spans[first_bad];
    char head[320];
    snprintf(head, sizeof head, "staging verify (end of staging): %llu of %llu span(s) no longer hold what was copied; "
             "first: arena [%llu, +%llu) of batch %p, %s %u, matched right after its copy: host sums %016llx/%016llx, "
             "device now %016llx/%016llx; ", (unsigned long long)bad, (unsigned long long)n,
             (unsigned long long)sp.off, (unsigned long long)sp.len, (void*)b,
             sp.thread == kStageInlineThread ? "inline window" : "reader thread",
             sp.thread == kStageInlineThread ? 0u : sp.thread, (unsigned long long)sp.s1, (unsigned long long)sp.s2,
             (unsigned long long)sums[2 * first_bad], (unsigned long long)sums[2 * first_bad + 1]);
    const std::string msg = head + describe_span(b->arena.as<u8>() + sp.off, nullptr, sp.len, c->stream);
    {
        std::lock_guard<std::mutex> g(b->span_mu);
        if (b->stage_note.empty()) b->stage_note = msg;
    }
    fprintf(stderr, "makisu_mi: %s\n", msg.c_str());
    b->stage_err = msg;                                        // sticky (stager_drain / stage_batch)
    return fail(c, MI_ERR_IO, "%s", msg.c_str());
}

}  // namespace mi
