// mi_tree.hip -- host side of the two callers of the hot path (no device code here):
// the directory walks that decide WHICH files reach the GPU batch and in what order.
//
// Restates, for a C++ host (the reference is Go; `go` is absent in this environment):
//   * filepath.Walk order as Go's path/filepath defines it: lstat the root, call the
//     visitor, and for a directory visit its names in sort.Strings (bytewise) order,
//     never following symlinks;
//   * MI_TREE_CONTEXT -- checksumPathContents (lib/builder/step/add_copy_step.go:194-238):
//     special files are skipped (a special directory would SkipDir), every other path
//     contributes filepath.Rel(contextDir, path), symlinks their target, files their bytes;
//   * MI_TREE_SCAN -- the snapshot walk (lib/snapshot/utils.go:37-75 walk/shouldSkip):
//     skip names starting with the AUFS whiteout-meta prefix ".wh..wh."
//     (lib/snapshot/const.go:17-22), descendants of the blacklist
//     (pathutils.IsDescendantOfAny, lib/pathutils/path.go:24-35), special files
//     (utils.IsSpecialFile, lib/utils/utils.go:161-163) and mountpoints
//     (mountutils.IsMountpoint, lib/mountutils/mountutils.go:54-93: targets of /proc/mounts
//     except "/"); a skipped directory is not descended into.
// Regular files are registered with the batch (mi_batch_add_paths, stat-time size as in
// tario.WriteEntry's io.CopyN, lib/tario/write.go:43-45) in visit order; results come back
// in the same order.
//
// The reference walks on ONE goroutine (filepath.Walk: an lstat per path, 2.4 us each here -- a
// layer of 100 000 small files spends 0.24 s on it, more than reading and scanning them takes).
// Round 3: the ENUMERATION is parallel -- directories are a work queue, N threads readdir + sort +
// fstatat (+ readlinkat) them and apply the skip rules -- and the ORDER is restored afterwards: one
// thread walks the finished directory records depth-first, names in sort.Strings order, exactly the
// sequence filepath.Walk produces, stopping at the first error in THAT order.  Same entries, same
// order, same errors (tests/test_host_walk.py compares the two on randomized trees); MI_WALK_THREADS
// (1 = the sequential walker below, the reference's shape) overrides the thread count.
//
// Also here, on entry lists and without state: the commit order, tario.IsSimilarHeader, the scan's layer diff
// (mi_snapshot_diff) and the layer merge (mi_entries_apply_layer).  MemFS itself, the copy ops and what surrounds a
// COPY step: mi_memfs.hip.
#include "mi_memtree.h"
#include "mi_filesum.h"     // the sums of a small file's bytes, taken where the directory reader read them

#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <ftw.h>
#include <grp.h>
#include <pwd.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <set>
#include <unordered_map>
#include <unordered_set>
#include <string>
#include <string_view>
#include <vector>


namespace mi_io {
std::atomic<uint64_t> content_opens{0}, content_bytes{0};
}

namespace mi_walk {

struct Walker {
    mi_batch* batch;
    std::string rel_base;
    std::string link_root;     // root that absolute symlink targets lose (empty = rel_base)
    std::vector<std::string> blacklist;
    std::vector<std::string> blacklist_clean;   // AbsPath of each, made when the first path asks (below_blacklist)
    std::atomic<bool> blacklist_cleaned{false};          // (asked by several directory readers at once)
    std::mutex blacklist_mu;
    bool below_blacklist(const std::string& path) {
        if (blacklist.empty()) return false;
        if (!blacklist_cleaned.load(std::memory_order_acquire)) {
            std::lock_guard<std::mutex> g(blacklist_mu);
            if (!blacklist_cleaned.load(std::memory_order_relaxed)) { blacklist_clean = cleaned_paths(blacklist); blacklist_cleaned.store(true, std::memory_order_release); }
        }
        return is_descendant_of_any_cleaned(path, blacklist_clean, blacklist);
    }
    static bool whiteout_meta(const std::string& path) {            // a base name that begins with ".wh..wh." (const.go:17-22)
        const size_t cut = path.find_last_of('/');
        return path.compare(cut == std::string::npos ? 0 : cut + 1, 8, ".wh..wh.") == 0;
    }
    uint32_t mode;
    Tree* tree;
    std::string err;
    int rc = MI_OK;
    int64_t n_regular = 0;
    // optional: "is this regular file's content known?" (path on disk, size, inode stamp) -- such a file gets its entry and
    // no row in the batch; asked by whoever stats the file (the directory readers: several threads)
    KnownFn content_known;
    static InodeStamp stamp_of(const struct stat& st) {
        InodeStamp s;
        s.dev = (uint64_t)st.st_dev;
        s.ino = (uint64_t)st.st_ino;
        s.mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
        s.ctime_ns = (int64_t)st.st_ctim.tv_sec * 1000000000ll + st.st_ctim.tv_nsec;
        return s;
    }
    // regular files wait here and go to the batch in bulk: the reader threads open them (several
    // at a time) while the walk goes on -- a per-file open + hand-over cost 9 us, the walk's lstat 2
    std::vector<std::string> pend_path;
    std::vector<uint64_t> pend_size, pend_tag;
    uint64_t batch_files = 0;                                // files the batch holds + pending ones
    bool counted = false;

    // what the (parallel) enumeration has seen so far, ahead of what was handed over: the arena is sized for it
    const std::atomic<uint64_t>* ahead_files = nullptr;
    const std::atomic<uint64_t>* ahead_bytes = nullptr;
    uint64_t handed_files = 0, handed_bytes = 0;

    // files whose bytes a directory reader already read (ParallelWalker::read_dir) and that lie in the arena as part of
    // their directory's block: table rows only
    std::vector<uint64_t> place_off, place_size, place_tag, place_sum;   // place_sum: 2 words per file (mi_filesum.h), or empty

    // the arena is sized for what the enumeration has seen, which runs ahead of what has been handed over
    bool reserve_ahead() {
        if (!ahead_files) return true;
        const uint64_t f = ahead_files->load(std::memory_order_relaxed), by = ahead_bytes->load(std::memory_order_relaxed);
        if (f > handed_files || by > handed_bytes) {
            const int r = mi_batch_reserve_ahead(batch, f > handed_files ? f - handed_files : 0, by > handed_bytes ? by - handed_bytes : 0);
            if (r && !rc) { rc = r; return false; }
        }
        return true;
    }
    void flush_placed() {
        if (place_off.empty() || !batch) return;
        const int r = mi_batch_add_placed(batch, place_off.size(), place_off.data(), place_size.data(), place_tag.data(),
                                          place_sum.size() == 2 * place_off.size() ? place_sum.data() : nullptr);
        if (r && !rc) rc = r;
        place_off.clear();
        place_size.clear();
        place_tag.clear();
        place_sum.clear();
    }
    // one directory's block (its small files, laid out as they are to lie in the arena) becomes a piece of the arena
    bool place_block(const std::shared_ptr<uint8_t[]>& blob, uint64_t len, uint64_t n_files, uint64_t* at) {
        if (!reserve_ahead()) return false;
        handed_files += n_files;
        handed_bytes += len;
        auto* keep = new std::shared_ptr<uint8_t[]>(blob);
        const int r = mi_batch_add_block(batch, blob.get(), len, [](void* p) { delete (std::shared_ptr<uint8_t[]>*)p; }, keep, at);
        if (r && !rc) rc = r;
        return r == MI_OK;
    }

    void flush_pending() {
        flush_placed();
        if (pend_path.empty() || !batch) return;
        if (ahead_files) {
            if (!reserve_ahead()) return;
            handed_files += pend_path.size();
            for (uint64_t sz : pend_size) handed_bytes += sz;
        }
        std::vector<const char*> ptrs(pend_path.size());
        for (size_t i = 0; i < ptrs.size(); ++i) ptrs[i] = pend_path[i].c_str();
        const int r = mi_batch_add_paths(batch, ptrs.size(), ptrs.data(), pend_size.data(), pend_tag.data());
        if (r && !rc) rc = r;                                // message already on the ctx
        pend_path.clear();
        pend_size.clear();
        pend_tag.clear();
    }

    bool should_skip(const std::string& path, const struct stat& st) {
        const bool special = S_ISCHR(st.st_mode) || S_ISBLK(st.st_mode) || S_ISFIFO(st.st_mode) || S_ISSOCK(st.st_mode);
        if (mode == MI_TREE_CONTEXT) return special;
        if (whiteout_meta(path)) return true;
        if (below_blacklist(path) || special) return true;
        const MountTable& mt = mountpoints();
        if (!mt.error.empty()) { err = "ismount: mountmanager initialize: " + mt.error; rc = MI_ERR_IO; return true; }
        return mt.targets.count(path) != 0;
    }

    // One visited, not skipped path -> its entry (+ its registration with the batch); link = the raw
    // readlink result of a symlink.  A directory's children are the caller's business.
    // rel (optional): the path's relpath when the caller already knows it (a child's is its parent's
    // plus its name; filepath.Rel per entry costs more than the lstat it follows)
    // placed (optional): the file's bytes are in the arena already, at this offset (its directory's block)
    // known (optional): the answer content_known already gave for this file (the parallel walk asks where it stats)
    // sum (with placed, optional): the file's (a, b) of mi_filesum.h as its reader computed them
    void emit(const std::string& path, const struct stat& st, const std::string* link, const std::string* rel = nullptr,
              const uint64_t* placed = nullptr, const uint8_t* known = nullptr, const uint64_t* sum = nullptr) {
        Entry e;
        e.relpath = rel ? *rel : rel_to(rel_base, path);
        if (e.relpath.empty()) {
            err = "path is outside of the base dir (" + rel_base + "," + path + ")";
            rc = MI_ERR_INVALID;
            return;
        }
        e.mode = (uint32_t)st.st_mode;
        e.mtime = (int64_t)st.st_mtime;
        e.uid = (uint32_t)st.st_uid;
        e.gid = (uint32_t)st.st_gid;
        if (S_ISDIR(st.st_mode)) {
            e.kind = 0;
            tree->push(std::move(e));
        } else if (S_ISLNK(st.st_mode)) {
            e.kind = 2;
            e.link = *link;
            e.has_link = true;
            if (mode == MI_TREE_SCAN && !e.link.empty() && e.link[0] == '/') {
                // memLayer.createHeader (lib/snapshot/mem_layer.go:171-185): an absolute target
                // loses the root prefix -- pathutils.TrimRoot (lib/pathutils/path.go:63-68):
                // plain string prefix, then AbsPath; a target outside the root fails the scan
                const std::string& lr = link_root.empty() ? rel_base : link_root;
                if (!has_prefix(e.link, lr)) {
                    err = "trim symlink root: failed to trim root prefix " + lr + " from path " + e.link;
                    rc = MI_ERR_INVALID;
                    return;
                }
                e.link = abs_path(e.link.substr(lr.size()));
            }
            tree->push(std::move(e));
        } else {
            e.kind = 1;
            e.size = (uint64_t)st.st_size;
            const InodeStamp stamp = stamp_of(st);
            const uint8_t kf = !batch ? 0 : known ? *known : content_known ? content_known(path, st, stamp) : 0;
            if (kf & kContentKnown) {
                e.file_index = -1;                               // no row: nothing of it is read
                tree->push(std::move(e), &stamp, kf);
                return;
            }
            if (batch) {
                if (!counted) { mi_batch_counts(batch, &batch_files, nullptr, nullptr); counted = true; }
                e.file_index = (int64_t)batch_files++;
                if (placed) {                               // (the batch's file table keeps the walk's order: one kind of
                    if (!pend_path.empty()) flush_pending();   // pending row at a time)
                    place_off.push_back(*placed);
                    place_size.push_back(e.size);
                    place_tag.push_back(tree->entries.size());
                    if (sum) { place_sum.push_back(sum[0]); place_sum.push_back(sum[1]); }
                    if (place_off.size() >= 4096) flush_placed();
                } else {
                    flush_placed();
                    pend_path.push_back(path);
                    pend_size.push_back(e.size);
                    pend_tag.push_back(tree->entries.size());
                    if (pend_path.size() >= 1024) flush_pending();
                }
                if (rc) return;
            } else {
                e.file_index = n_regular;                   // listing only: running file ordinal
            }
            ++n_regular;
            tree->push(std::move(e), &stamp);
        }
    }

    // the sequential walk: filepath.Walk as the reference runs it
    void visit(const std::string& path) {
        if (rc) return;
        struct stat st;
        if (lstat(path.c_str(), &st) != 0) {
            err = "lstat " + path + ": " + strerror(errno);
            rc = MI_ERR_IO;
            return;
        }
        if (should_skip(path, st)) return;                  // a skipped directory is not entered
        if (S_ISDIR(st.st_mode)) {
            emit(path, st, nullptr);
            if (rc) return;
            std::vector<std::string> names;
            DIR* d = opendir(path.c_str());
            if (!d) { err = "open " + path + ": " + strerror(errno); rc = MI_ERR_IO; return; }
            while (struct dirent* de = readdir(d)) {
                if (!strcmp(de->d_name, ".") || !strcmp(de->d_name, "..")) continue;
                names.push_back(de->d_name);
            }
            closedir(d);
            std::sort(names.begin(), names.end());          // sort.Strings: bytewise
            for (const std::string& n : names) visit(path == "/" ? "/" + n : path + "/" + n);
        } else if (S_ISLNK(st.st_mode)) {
            std::vector<char> buf(4096);
            ssize_t n = readlink(path.c_str(), buf.data(), buf.size() - 1);
            if (n < 0) { err = "read link " + path + ": " + strerror(errno); rc = MI_ERR_IO; return; }
            const std::string link(buf.data(), (size_t)n);
            emit(path, st, &link);
        } else {
            emit(path, st, nullptr);
        }
    }
};

// ---- parallel enumeration, sequential order -------------------------------------------------------
struct DirRec;
struct Child {
    std::string name, link;
    uint32_t mode = 0, uid = 0, gid = 0;
    uint64_t size = 0;
    int64_t mtime = 0;
    InodeStamp stamp;                    // regular files
    uint8_t known = 0;                   // ... whose content the caller knows (Walker::content_known's flags): not read, not staged
    bool skip = false;
    int rc = MI_OK;                      // this path's own failure (lstat / readlink / skip-rule / read error)
    std::string err;
    uint64_t blob_off = ~0ull;           // a regular file that was read where it was listed: its place in the directory's block
    uint64_t sum[2] = {0, 0};            // ... and the sums of its bytes as read() delivered them (mi_filesum.h; a batch that keeps sums)
    std::unique_ptr<DirRec> sub;         // a directory that is entered
};
struct DirRec {
    std::string path;
    std::vector<Child> kids;             // sort.Strings order
    int rc = MI_OK;                      // opening / reading the directory failed
    std::string err;
    bool done = false;                   // read completely (guarded by ParallelWalker::mu)
    // the directory's small regular files, read by the thread that listed them, laid out as they will lie in the arena
    // (each on a 256-byte boundary): ONE piece of the arena, copied by the reader threads of mi_stage.hip
    std::shared_ptr<uint8_t[]> blob;
    uint64_t blob_len = 0, blob_files = 0;
};

// Small files are read WHERE THEY ARE LISTED (round 4).  Until then every regular file went to the reader threads of
// mi_stage.hip as a path: a second path walk, an open, a pread and a close per file there -- and open / close take the
// lock of the file-descriptor table, which all threads of a process share: on the GPU box 100 000 4 KiB files took
// 95-135 ms with 4, 8, 16 or 32 threads alike (profiles/r04_ubench_small_file_reads.txt).  The directory readers of the
// walk are threads of this library that touch nothing but files, so they LEAVE the shared table -- unshare(CLONE_FILES),
// then close_range over their private copy -- and open / read / close against a table of their own: the same files in
// 56 / 42 / 29 ms with 8 / 16 / 32 threads.  (A thread that may not unshare -- a seccomp profile without it -- reads
// through the shared table: correct, slower.)  What a reader thread of mi_stage.hip is left with is a memcpy into its
// pinned slab and the host-to-device copy.  Bounded: files up to MI_WALK_INLINE_MAX_KIB (default 16) KiB -- the block
// costs a second copy of the bytes, which pays while the per-file system calls dominate: 100 000 x 4 KiB 4.5-5.3 ->
// 9-12.5 GB/s end to end; at 16 KiB both ways give 22 GB/s, at 32 KiB paths 27 against 24, at 64 KiB 42 against 32
// (profiles/r04_many_small_files.txt) -- and at most MI_WALK_INLINE_MB (default 1024) MiB of blocks alive at a time;
// beyond either a file goes the old way, as a path.  MI_WALK_UNSHARE=0 keeps the shared table (ThreadSanitizer models
// descriptors per process and takes two threads' private "fd 4" for one).
static uint64_t inline_file_max() {
    static const uint64_t v = [] {
        const char* e = getenv("MI_WALK_INLINE_MAX_KIB");
        const long kib = e && *e ? atol(e) : 16;
        const uint64_t v = kib <= 0 ? 0ull : (uint64_t)kib << 10;
        return v > mi_sum::kChunk ? mi_sum::kChunk : v;          // (a file read in place carries ONE chunk's sums)
    }();
    return v;
}
// The blocks' memory: recycled, not returned.  A 2 MB block from the C library is its own mapping -- page faults for
// every one of its pages, by 16 threads in one address space, and an unmap when the reader threads let go of it: 100 000
// 4 KiB files then spend more time faulting than reading.  Freed blocks wait here by size class (powers of two from
// 64 KiB; at most 512 MiB kept resident while a walk runs, 64 MiB between walks: MI_WALK_POOL_MB / MI_WALK_POOL_IDLE_MB), and since a block lives only until a reader thread has copied it, a walk
// cycles through a few tens of megabytes.
// New blocks are CARVED from 32 MiB slabs (one mmap on a 2 MiB boundary, advised to use huge pages, never unmapped):
// a mapping of its own per block cost three address-space calls (mmap, the munmap that trims it to the boundary,
// madvise), each under the address-space lock in write mode with 16 directory readers faulting beside it -- a
// process's first walk spent 100-180 ms of its readers' time there (profiles/r04_many_small_files.txt).  With slabs
// it is one such call per 16 blocks of 2 MiB, and a block's first touch is one huge-page fault.  A block beyond the
// resident limit gives its pages back (MADV_DONTNEED) and keeps its address range for the next taker.
struct BlockPool {
    static constexpr uint64_t kSlab = 32ull << 20, kHuge = 2ull << 20;
    std::mutex mu;
    std::vector<uint8_t*> free_[16];                          // class k: 64 KiB << k; resident blocks
    std::vector<uint8_t*> cold_[16];                          // carved blocks whose pages were given back
    uint64_t kept = 0;                                        // bytes in free_
    // what may stay resident: MI_WALK_POOL_MB (default 512) while a walk runs, MI_WALK_POOL_IDLE_MB (default 64) between
    // walks -- a long-lived host (the Go builder) does not pay half a gigabyte of RSS for ever after its first walk; the
    // address ranges of carved blocks stay (their pages are given back: MADV_DONTNEED), mappings of their own are unmapped
    int walks = 0;                                            // walks under way (any thread, any ctx: the pool is the process's)
    static uint64_t env_mb(const char* name, long dflt) {
        const char* e = getenv(name);
        const long mb = e && *e ? atol(e) : dflt;
        return mb <= 0 ? 0ull : (uint64_t)mb << 20;
    }
    uint64_t limit() const {                                  // mu held
        static const uint64_t busy = env_mb("MI_WALK_POOL_MB", 512), idle = env_mb("MI_WALK_POOL_IDLE_MB", 64);
        return walks > 0 ? busy : idle;
    }
    void walk_begins() { std::lock_guard<std::mutex> g(mu); ++walks; }
    void walk_ends() {
        std::vector<std::pair<uint8_t*, uint64_t>> go;
        {
            std::lock_guard<std::mutex> g(mu);
            if (--walks > 0) return;
            for (int k = 15; k >= 0 && kept > limit(); --k)   // the largest blocks first
                while (!free_[k].empty() && kept > limit()) {
                    go.emplace_back(free_[k].back(), cap(k));
                    free_[k].pop_back();
                    kept -= cap(k);
                }
        }
        if (const char* e = getenv("MI_WALK_TIMING"); e && *e && *e != '0')
            fprintf(stderr, "mi_walk: block pool between walks: %.1f MB resident (limit %.0f), %zu blocks given back\n", resident() / 1e6,
                    (double)(env_mb("MI_WALK_POOL_IDLE_MB", 64) >> 20), go.size());
        for (auto& b : go) {
            if (carved(b.second)) {
                (void)madvise(b.first, b.second, MADV_DONTNEED);
                std::lock_guard<std::mutex> g(mu);
                cold_[cls(b.second)].push_back(b.first);
            } else {
                munmap(b.first, b.second);
            }
        }
    }
    uint64_t resident() { std::lock_guard<std::mutex> g(mu); return kept; }
    uint8_t* slab_at = nullptr;                               // the current slab's unused tail
    uint64_t slab_left = 0;
    uint64_t n_slabs = 0, n_carved = 0, n_own = 0;            // MI_WALK_TIMING
    static int cls(uint64_t n) { int k = 0; while ((65536ull << k) < n && k < 15) ++k; return k; }
    static uint64_t cap(int k) { return 65536ull << k; }
    static bool carved(uint64_t capacity) { return capacity <= kSlab / 2; }   // larger blocks are mappings of their own
    static uint8_t* map_aligned(uint64_t bytes) {
        void* p = mmap(nullptr, bytes + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) return nullptr;
        uint8_t* a = (uint8_t*)(((uintptr_t)p + kHuge - 1) & ~((uintptr_t)kHuge - 1));
        if (a > (uint8_t*)p) munmap(p, (size_t)(a - (uint8_t*)p));
        const uint64_t tail = (uint64_t)((uint8_t*)p + bytes + kHuge - (a + bytes));
        if (tail) munmap(a + bytes, tail);
        (void)madvise(a, bytes, MADV_HUGEPAGE);
        return a;
    }
    uint8_t* carve(uint64_t capacity) {                       // mu held
        const uint64_t align = capacity < kHuge ? capacity : kHuge;   // capacities are powers of two
        uint64_t skip = slab_at ? (align - ((uintptr_t)slab_at & (align - 1))) & (align - 1) : 0;
        if (!slab_at || skip + capacity > slab_left) {
            uint8_t* s = map_aligned(kSlab);                  // what is left of the old slab is lost (< one block)
            if (!s) return nullptr;
            slab_at = s;
            slab_left = kSlab;
            skip = 0;
            ++n_slabs;
        }
        uint8_t* p = slab_at + skip;
        slab_at = p + capacity;
        slab_left -= skip + capacity;
        ++n_carved;
        return p;
    }
    uint8_t* get(uint64_t n, uint64_t* cap_out) {
        const int k = cls(n);
        if (cap(k) < n) {                                                             // beyond the largest class
            *cap_out = (n + kHuge - 1) & ~(kHuge - 1);
            std::lock_guard<std::mutex> g(mu);
            ++n_own;
            return map_aligned(*cap_out);
        }
        *cap_out = cap(k);
        std::lock_guard<std::mutex> g(mu);
        if (!free_[k].empty()) {
            uint8_t* p = free_[k].back();
            free_[k].pop_back();
            kept -= cap(k);
            return p;
        }
        if (!cold_[k].empty()) {
            uint8_t* p = cold_[k].back();
            cold_[k].pop_back();
            return p;
        }
        if (carved(cap(k))) return carve(cap(k));
        ++n_own;
        return map_aligned(cap(k));
    }
    void put(uint8_t* p, uint64_t capacity) {
        const int k = cls(capacity);
        if (cap(k) == capacity) {
            std::unique_lock<std::mutex> g(mu);
            if (kept + capacity <= limit()) { free_[k].push_back(p); kept += capacity; return; }
            if (carved(capacity)) {                           // part of a slab: the pages go, the range stays
                g.unlock();
                (void)madvise(p, capacity, MADV_DONTNEED);
                g.lock();
                cold_[k].push_back(p);
                return;
            }
        }
        munmap(p, capacity);
    }
};
static BlockPool& block_pool() { static BlockPool* p = new BlockPool(); return *p; }   // never destroyed: blocks may outlive exit handlers

// MI_WALK_TIMING=1: one line per walk on stderr -- where the directory readers' time went (summed over the threads) and
// when the assembly and the readers were done
static bool walk_timing() {
    static const bool on = [] { const char* e = getenv("MI_WALK_TIMING"); return e && *e && *e != '0'; }();
    return on;
}
static std::atomic<uint64_t> g_ns_list{0}, g_ns_stat{0}, g_ns_block{0}, g_ns_read{0}, g_ns_unshare{0};
static inline uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool walk_unshare() {
    static const bool on = [] { const char* e = getenv("MI_WALK_UNSHARE"); return !(e && *e == '0'); }();
    return on;
}
static bool walk_close_range() {                                     // MI_WALK_CLOSE_RANGE=0: as if the kernel had no close_range
    static const bool on = [] { const char* e = getenv("MI_WALK_CLOSE_RANGE"); return !(e && *e == '0'); }();
    return on;
}
#define MI_CLOSE_RANGE_UNSHARE (1u << 1)                              // linux/close_range.h (absent from older kernel headers)
static std::atomic<uint64_t> g_inline_bytes{0};                    // bytes in blocks alive now
static std::atomic<uint64_t> g_inline_peak{0}, g_inline_files{0};  // MI_WALK_TIMING: the most there were; files read into blocks
static uint64_t inline_budget() {
    static const uint64_t v = [] {
        const char* e = getenv("MI_WALK_INLINE_MB");
        const long mb = e && *e ? atol(e) : 1024;
        return mb <= 0 ? 0ull : (uint64_t)mb << 20;
    }();
    return v;
}

struct ParallelWalker {
    std::atomic<uint64_t> seen_files{0}, seen_bytes{0};        // regular files enumerated so far (the batch reserves for them)
    Walker* w;                           // rules, rel_base, blacklist, mode; receives the entries
    std::mutex mu;
    std::condition_variable cv, cv_done; // work for the readers / a finished directory for the assembly
    std::vector<DirRec*> stack;          // LIFO: close to the depth-first order the assembly wants
    size_t outstanding = 0;              // directories queued or being read
    bool abort = false;                  // the assembly stopped (an error): the readers only drain
    bool inline_reads = false;           // a batch is attached: small files are read where they are listed
    bool want_sums = false;              // ... and it keeps its files' source sums (mi_batch_keeps_sums)
    static thread_local bool own_table;  // this directory reader left the process's descriptor table (worker())
    static uint64_t fd_budget() {        // descriptors a thread may hold at once (its table's soft limit)
        static const uint64_t v = [] {
            struct rlimit rl;
            return getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur != RLIM_INFINITY ? (uint64_t)rl.rlim_cur : 1024ull;
        }();
        return v;
    }
    std::vector<std::thread> pool;

    // Walker::should_skip without side effects on the shared walker: a broken mounts table is the
    // child's own error here
    bool skip_rule(const std::string& path, const struct stat& st, int* rc, std::string* err) const {
        const bool special = S_ISCHR(st.st_mode) || S_ISBLK(st.st_mode) || S_ISFIFO(st.st_mode) || S_ISSOCK(st.st_mode);
        if (w->mode == MI_TREE_CONTEXT) return special;
        if (Walker::whiteout_meta(path)) return true;
        if (w->below_blacklist(path) || special) return true;
        const MountTable& mt = mountpoints();
        if (!mt.error.empty()) { *err = "ismount: mountmanager initialize: " + mt.error; *rc = MI_ERR_IO; return true; }
        return mt.targets.count(path) != 0;
    }

    // one directory: names, sorted; every child's lstat, skip decision, link target; the
    // subdirectories that are entered go back on the stack
    void read_dir(DirRec* d) {
        const int fd = open(d->path.c_str(), O_RDONLY | O_DIRECTORY | O_CLOEXEC);
        DIR* dir = fd >= 0 ? fdopendir(fd) : nullptr;
        if (!dir) {
            d->rc = MI_ERR_IO;
            d->err = "open " + d->path + ": " + strerror(errno);
            if (fd >= 0) close(fd);
            return;
        }
        const uint64_t tl0 = walk_timing() ? now_ns() : 0;
        // name, d_type -- scratch of this thread that keeps its capacity from directory to directory (a process's first
        // walk grows 16 fresh heaps at once; every growth step is an address-space call beside the others' page faults)
        static thread_local std::vector<std::pair<std::string, unsigned char>> names;
        static thread_local std::string path;
        names.clear();
        while (struct dirent* de = readdir(dir)) {
            if (!strcmp(de->d_name, ".") || !strcmp(de->d_name, "..")) continue;
            names.emplace_back(de->d_name, de->d_type);
        }
        std::sort(names.begin(), names.end(),               // sort.Strings: bytewise
                  [](const std::pair<std::string, unsigned char>& a, const std::pair<std::string, unsigned char>& b) { return a.first < b.first; });
        d->kids.resize(names.size());
        std::vector<DirRec*> subs;
        // A regular file that will be read here anyway is OPENED first and its header taken from the descriptor (fstat): one
        // path lookup per file instead of two.  Only on a descriptor table of this thread's own (every file of the directory is
        // open at once until the block is laid out) and only while the directory fits the table.
        std::vector<int> fds;
        // (Not when the caller may know a file's content -- MI_MEMFS_TRUST_CTIME: most files are then not read at all, and an
        // open + fstat + close each is three system calls where one fstatat does.)
        const bool open_first = inline_reads && own_table && !w->content_known && names.size() + 64 <= fd_budget();
        if (open_first) fds.assign(names.size(), -1);
        const uint64_t ts0 = tl0 ? now_ns() : 0;
        if (tl0) g_ns_list += ts0 - tl0;
        for (size_t i = 0; i < names.size(); ++i) {
            Child& c = d->kids[i];
            c.name.swap(names[i].first);
            path.assign(d->path == "/" ? "" : d->path);
            path += '/';
            path += c.name;
            struct stat st;
            bool have = false;
            if (open_first && names[i].second == DT_REG) {
                const int f = openat(fd, c.name.c_str(), O_RDONLY | O_CLOEXEC | O_NOFOLLOW | O_NONBLOCK);
                if (f >= 0) {
                    if (fstat(f, &st) == 0 && S_ISREG(st.st_mode)) { fds[i] = f; have = true; }
                    else close(f);                           // no longer what readdir said: the lstat below decides
                }                                            // (unreadable, gone, a link by now: likewise)
            }
            if (!have && fstatat(fd, c.name.c_str(), &st, AT_SYMLINK_NOFOLLOW) != 0) {
                c.rc = MI_ERR_IO;
                c.err = "lstat " + path + ": " + strerror(errno);
                continue;
            }
            c.skip = skip_rule(path, st, &c.rc, &c.err);
            if (c.rc || c.skip) continue;
            c.mode = (uint32_t)st.st_mode;
            c.mtime = (int64_t)st.st_mtime;
            c.uid = (uint32_t)st.st_uid;
            c.gid = (uint32_t)st.st_gid;
            c.size = (uint64_t)st.st_size;
            if (S_ISREG(st.st_mode)) {
                c.stamp = Walker::stamp_of(st);
                c.known = w->batch && w->content_known ? w->content_known(path, st, c.stamp) : 0;
                if (!c.known) {
                    seen_files.fetch_add(1, std::memory_order_relaxed);
                    seen_bytes.fetch_add((uint64_t)st.st_size, std::memory_order_relaxed);
                }
            }
            if (S_ISDIR(st.st_mode)) {
                c.sub.reset(new DirRec());
                c.sub->path = path;
                subs.push_back(c.sub.get());
            } else if (S_ISLNK(st.st_mode)) {
                char buf[4096];
                const ssize_t n = readlinkat(fd, c.name.c_str(), buf, sizeof buf - 1);
                if (n < 0) { c.rc = MI_ERR_IO; c.err = "read link " + path + ": " + strerror(errno); continue; }
                c.link.assign(buf, (size_t)n);
            }
        }
        if (ts0) g_ns_stat += now_ns() - ts0;
        if (inline_reads) read_small_files(d, fd, fds);
        for (int f : fds) if (f >= 0) close(f);              // skipped, too large for a block, or no block to be had
        closedir(dir);                                       // closes fd
        if (!subs.empty()) {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = subs.size(); i-- > 0;) stack.push_back(subs[i]);   // first name on top
            outstanding += subs.size();
            cv.notify_all();
        }
    }

    // the directory's small regular files into one block; a file that cannot be read is that child's error (reported
    // by the assembly in walk order, with the words the reader threads use)
    // fds (may be empty): descriptors read_dir already holds, by child index; taken (closed, set to -1) as they are read
    void read_small_files(DirRec* d, int dfd, std::vector<int>& fds) {
        uint64_t total = 0, n = 0;
        for (Child& c : d->kids) {
            if (c.rc || c.skip || c.known || !S_ISREG(c.mode) || c.size > inline_file_max()) continue;
            c.blob_off = (total + 255) & ~255ull;
            total = c.blob_off + c.size;
            ++n;
        }
        if (n == 0) return;
        const uint64_t held = g_inline_bytes.fetch_add(total) + total;
        if (walk_timing()) { uint64_t pk = g_inline_peak.load(); while (held > pk && !g_inline_peak.compare_exchange_weak(pk, held)) {} }
        if (held > inline_budget()) {                        // too much host memory in blocks already -- or this one directory
                                                             // alone is more than the budget: these files go as paths
            g_inline_bytes.fetch_sub(total);
            for (Child& c : d->kids) c.blob_off = ~0ull;
            return;
        }
        uint64_t capacity = 0;
        const uint64_t tb0 = walk_timing() ? now_ns() : 0;
        uint8_t* buf = total ? block_pool().get(total, &capacity) : nullptr;
        if (tb0 && buf) { memset(buf, 0, 1); g_ns_block += now_ns() - tb0; }
        if (total && !buf) {
            g_inline_bytes.fetch_sub(total);
            for (Child& c : d->kids) c.blob_off = ~0ull;
            return;
        }
        d->blob = std::shared_ptr<uint8_t[]>(buf, [total, capacity](uint8_t* p) {
            if (p) block_pool().put(p, capacity);
            g_inline_bytes.fetch_sub(total);
        });
        d->blob_len = total;
        d->blob_files = n;
        if (walk_timing()) g_inline_files += n;
        uint64_t end = 0;
        const uint64_t tr0 = walk_timing() ? now_ns() : 0;
        struct AddUp { uint64_t t0; ~AddUp() { if (t0) g_ns_read += now_ns() - t0; } } add_up{tr0};
        for (size_t ci = 0; ci < d->kids.size(); ++ci) {
            Child& c = d->kids[ci];
            if (c.blob_off == ~0ull) continue;
            if (c.blob_off > end) memset(buf + end, 0, c.blob_off - end);       // alignment gap
            end = c.blob_off + c.size;
            if (c.size == 0) continue;
            int fd = -1;
            mi_io::content_opens.fetch_add(1, std::memory_order_relaxed);
            mi_io::content_bytes.fetch_add(c.size, std::memory_order_relaxed);
            if (!fds.empty() && fds[ci] >= 0) { fd = fds[ci]; fds[ci] = -1; }
            else fd = openat(dfd, c.name.c_str(), O_RDONLY | O_CLOEXEC | O_NOFOLLOW);
            if (fd < 0) {
                c.rc = MI_ERR_IO;
                c.err = "open " + (d->path == "/" ? "/" + c.name : d->path + "/" + c.name) + ": " + strerror(errno);
                continue;
            }
            uint64_t got = 0;
            while (got < c.size) {
                const ssize_t r = pread(fd, buf + c.blob_off + got, c.size - got, (off_t)got);
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) {
                    c.rc = MI_ERR_IO;
                    c.err = "read " + (d->path == "/" ? "/" + c.name : d->path + "/" + c.name) + ": " +
                            (r == 0 ? std::string("file shorter than the size given") : std::string(strerror(errno)));
                    break;
                }
                got += (uint64_t)r;
            }
            close(fd);
            if (want_sums && !c.rc) mi_sum::chunk_add(buf + c.blob_off, (size_t)c.size, 0, &c.sum[0], &c.sum[1]);
        }
    }

    void worker() {
        const uint64_t tu0 = walk_timing() ? now_ns() : 0;
        own_table = false;
        // A descriptor table of this thread's own (see above), and an EMPTY one: close_range with CLOSE_RANGE_UNSHARE does both
        // in one call -- the table is unshared and the copies of the process's descriptors go.  Where that call does not
        // exist (kernels before 5.9: ENOSYS) or is refused (a seccomp profile), the thread STAYS on the shared table: a
        // private table still full of the process's descriptors would hold the host's sockets and pipes open for the whole
        // walk (a close on the host's side would send no FIN, reach no EOF) and would not have room for open_first's
        // descriptors.  MI_WALK_CLOSE_RANGE=0 takes that way on purpose (tests).
        // (Also for a walk that reads nothing: sixteen readers opening and closing directories on ONE table contend for its lock.)
        if (walk_unshare() && walk_close_range() &&
            syscall(SYS_close_range, 3u, ~0u, (unsigned)MI_CLOSE_RANGE_UNSHARE) == 0)
            own_table = true;
        if (tu0) g_ns_unshare += now_ns() - tu0;
        for (;;) {
            DirRec* d = nullptr;
            bool skip_read = false;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !stack.empty() || outstanding == 0; });
                if (stack.empty()) return;
                d = stack.back();
                stack.pop_back();
                skip_read = abort;
            }
            if (!skip_read) read_dir(d);
            {
                std::lock_guard<std::mutex> g(mu);
                d->done = true;
                if (--outstanding == 0) cv.notify_all();
            }
            cv_done.notify_all();
        }
    }

    // the readers run on their own threads; the caller assembles behind them
    void start(DirRec* root, unsigned n_threads) {
        stack.push_back(root);
        outstanding = 1;
        for (unsigned i = 0; i < n_threads; ++i) pool.emplace_back([this] { worker(); });
    }
    void finish() {
        {
            std::lock_guard<std::mutex> g(mu);
            abort = true;                                    // whatever is still queued is only drained
        }
        cv.notify_all();
        for (auto& t : pool) t.join();
        pool.clear();
    }
    void wait_done(const DirRec* d) {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return d->done; });
    }

    // filepath.Walk's sequence over the finished records (the directory's own entry was emitted by
    // the caller): children in order, a directory child followed by its subtree; the first failure
    // in THAT order is the walk's failure
    // rel = the directory's own relpath ("." for the base itself)
    void assemble(DirRec* d, const std::string& rel) {
        if (w->rc) return;
        wait_done(d);                                        // usually is: the readers go depth-first too
        if (d->rc) { w->rc = d->rc; w->err = d->err; return; }
        uint64_t block_at = 0;
        if (d->blob_len || d->blob_files) {
            if (d->blob_len) {
                const bool placed = w->place_block(d->blob, d->blob_len, d->blob_files, &block_at);
                d->blob.reset();                             // the stager's pieces own the block now: it goes back to the pool when
                if (!placed) return;                         // the last of them sits in a slab, not when the walk ends
            }
            else { w->handed_files += d->blob_files; }           // empty files only: rows without bytes
        }
        std::string path, crel;
        for (const Child& c : d->kids) {
            if (c.rc) { w->rc = c.rc; w->err = c.err; return; }
            if (c.skip) continue;
            path.assign(d->path == "/" ? "" : d->path).append("/").append(c.name);
            if (rel == ".") crel = c.name; else crel.assign(rel).append("/").append(c.name);
            struct stat st;
            memset(&st, 0, sizeof st);
            st.st_mode = c.mode;
            st.st_mtime = c.mtime;
            st.st_uid = c.uid;
            st.st_gid = c.gid;
            st.st_size = (off_t)c.size;
            if (S_ISREG(c.mode)) {                               // the inode stamp as the reader's stat gave it (stamp_of's inverse)
                st.st_dev = (dev_t)c.stamp.dev;
                st.st_ino = (ino_t)c.stamp.ino;
                st.st_mtim.tv_nsec = (long)(c.stamp.mtime_ns - (int64_t)c.mtime * 1000000000ll);
                const int64_t cs = c.stamp.ctime_ns >= 0 ? c.stamp.ctime_ns / 1000000000ll : -((-c.stamp.ctime_ns + 999999999ll) / 1000000000ll);
                st.st_ctim.tv_sec = (time_t)cs;
                st.st_ctim.tv_nsec = (long)(c.stamp.ctime_ns - cs * 1000000000ll);
            }
            const uint64_t placed = block_at + (c.blob_off == ~0ull ? 0 : c.blob_off);
            w->emit(path, st, S_ISLNK(c.mode) ? &c.link : nullptr, &crel, c.blob_off == ~0ull ? nullptr : &placed,
                    S_ISREG(c.mode) ? &c.known : nullptr, c.blob_off != ~0ull && want_sums ? c.sum : nullptr);
            if (w->rc) return;
            if (c.sub) { assemble(c.sub.get(), crel); if (w->rc) return; }
        }
    }
};

thread_local bool ParallelWalker::own_table = false;

static unsigned walk_threads() {
    if (const char* e = getenv("MI_WALK_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) return (unsigned)v;
    }
    cpu_set_t set;
    unsigned n = sched_getaffinity(0, sizeof set, &set) == 0 ? (unsigned)CPU_COUNT(&set) : 1u;
    return n > 16 ? 16 : (n < 1 ? 1 : n);
}

// the walk of `root` into w->tree (and w->batch): the sequential Walker::visit with one thread, else
// the parallel enumeration + ordered assembly
static void walk_root(Walker* w, const std::string& root) {
    const unsigned nt = walk_threads();
    if (nt <= 1) { w->visit(root); return; }
    struct stat st;
    if (lstat(root.c_str(), &st) != 0) { w->err = "lstat " + root + ": " + strerror(errno); w->rc = MI_ERR_IO; return; }
    if (w->should_skip(root, st) || w->rc) return;
    std::string link;
    if (S_ISLNK(st.st_mode)) {
        char buf[4096];
        const ssize_t n = readlink(root.c_str(), buf, sizeof buf - 1);
        if (n < 0) { w->err = "read link " + root + ": " + strerror(errno); w->rc = MI_ERR_IO; return; }
        link.assign(buf, (size_t)n);
    }
    w->emit(root, st, S_ISLNK(st.st_mode) ? &link : nullptr);
    if (w->rc || !S_ISDIR(st.st_mode)) return;
    ParallelWalker pw;
    pw.w = w;
    if (w->batch && inline_budget()) {
        const char* e = getenv("MI_WALK_INLINE");
        pw.inline_reads = !(e && *e == '0');
        pw.want_sums = mi_batch_keeps_sums(w->batch) != 0;
    }
    struct PoolUse { bool on; PoolUse(bool o) : on(o) { if (on) block_pool().walk_begins(); } ~PoolUse() { if (on) block_pool().walk_ends(); } }
        pool_use(pw.inline_reads);                                  // (blocks still with the reader threads come back under the idle limit)
    w->ahead_files = &pw.seen_files;
    w->ahead_bytes = &pw.seen_bytes;
    struct Detach { Walker* w; ~Detach() { w->ahead_files = w->ahead_bytes = nullptr; } } detach{w};
    DirRec top;
    top.path = root;
    const std::string root_rel = w->tree->entries.back().relpath;   // the root's own entry was just emitted (a copy:
    const uint64_t tw0 = walk_timing() ? now_ns() : 0;
    if (tw0) { g_ns_list = 0; g_ns_stat = 0; g_ns_block = 0; g_ns_read = 0; g_ns_unshare = 0; g_inline_peak = 0; g_inline_files = 0; }
    pw.start(&top, nt);                                               // the vector grows under the assembly)
    pw.assemble(&top, root_rel);       // behind the readers: files reach the batch while the walk goes on
    const uint64_t tw1 = tw0 ? now_ns() : 0;
    pw.finish();
    if (tw0)
        fprintf(stderr, "mi_walk: %u threads; assembly done after %.1f ms, threads joined after %.1f ms; summed over the threads: "
                "unshare %.1f ms, readdir + sort %.1f, fstatat + rules %.1f, block from the pool (first byte touched) %.1f, "
                "open + pread + close %.1f; %llu files seen; block pool so far: %llu slabs, %llu blocks carved, %llu mappings of "
                "their own, at most %.1f MB of blocks alive, %llu files read into blocks\n", nt, (tw1 - tw0) / 1e6, (now_ns() - tw0) / 1e6,
                g_ns_unshare.load() / 1e6, g_ns_list.load() / 1e6, g_ns_stat.load() / 1e6, g_ns_block.load() / 1e6,
                g_ns_read.load() / 1e6, (unsigned long long)pw.seen_files.load(), (unsigned long long)block_pool().n_slabs,
                (unsigned long long)block_pool().n_carved, (unsigned long long)block_pool().n_own, g_inline_peak.load() / 1e6, (unsigned long long)g_inline_files.load());
}

// the walk the copy ops need (mi_memfs.hip): one source, scan rules, no blacklist
int scan_walk_collect(const std::string& src, const std::string& link_root, Tree* out, std::string* err) {
    Walker w;
    w.batch = nullptr;
    w.rel_base = src;
    w.mode = MI_TREE_SCAN;
    w.tree = out;
    w.link_root = link_root;
    w.visit(src);
    if (w.rc && err) *err = w.err;
    return w.rc;
}

// the same walk with a batch attached: the source's regular files are staged while it is listed (parallel enumeration,
// small files read where they are listed), an entry's file_index = its row in the batch
int scan_walk_collect_batch(const std::string& src, const std::string& link_root, Tree* out, std::string* err, mi_batch* b) {
    Walker w;
    w.batch = b;
    w.rel_base = src;
    w.mode = MI_TREE_SCAN;
    w.tree = out;
    w.link_root = link_root;
    mi_batch_expect_host_bytes(b);
    walk_root(&w, src);
    w.flush_pending();
    if (w.rc && err) *err = w.err;
    return w.rc;
}

int scan_walk_listing(const std::string& root, const std::vector<std::string>& blacklist, Tree* out, std::string* err) {
    out->want_stamps = true;
    Walker w;
    w.batch = nullptr;
    w.rel_base = root;
    w.mode = MI_TREE_SCAN;
    w.tree = out;
    w.blacklist = blacklist;
    std::string r = root;
    while (r.size() > 1 && r.back() == '/') r.pop_back();
    walk_root(&w, r);
    if (w.rc && err) *err = w.err;
    return w.rc;
}

int scan_walk_batch_filtered(mi_batch* b, const std::string& root, const std::vector<std::string>& blacklist,
                             const KnownFn& known, Tree** tree_out, std::string* err) {
    void** slot = mi_batch_tree_slot(b);
    if (!*slot) *slot = new Tree();
    Tree* t = (Tree*)*slot;
    t->want_stamps = true;
    if (t->stamps.size() < t->entries.size()) { t->stamps.resize(t->entries.size()); t->known.resize(t->entries.size()); }
    Walker w;
    w.batch = b;
    w.rel_base = root;
    w.mode = MI_TREE_SCAN;
    w.tree = t;
    w.blacklist = blacklist;
    w.content_known = known;
    std::string r = root;
    while (r.size() > 1 && r.back() == '/') r.pop_back();
    mi_batch_expect_host_bytes(b);
    walk_root(&w, r);
    w.flush_pending();
    if (w.rc && err) *err = w.err;
    if (tree_out) *tree_out = t;
    return w.rc;
}

}  // namespace mi_walk


using mi_walk::Tree;

// the batch keeps its tree behind an opaque pointer (mi_api.hip owns the slot)

extern "C" {

int mi_batch_add_tree(mi_batch* b, const char* root, const char* rel_base, const char* const* blacklist,
                      uint64_t n_blacklist, uint32_t mode, uint64_t* n_entries) {
    if (!b || !root || (n_blacklist && !blacklist) || mode > MI_TREE_SCAN) return MI_ERR_INVALID;
    void** slot = mi_batch_tree_slot(b);
    if (!*slot) *slot = new Tree();
    Tree* t = (Tree*)*slot;
    t->want_stamps = true;                                       // (a commit records the inodes it hashed: mi_memfs.hip)
    if (t->stamps.size() < t->entries.size()) { t->stamps.resize(t->entries.size()); t->known.resize(t->entries.size()); }
    mi_walk::Walker w;
    w.batch = b;
    w.rel_base = rel_base ? rel_base : root;
    w.mode = mode;
    w.tree = t;
    for (uint64_t i = 0; i < n_blacklist; ++i) w.blacklist.push_back(blacklist[i]);
    std::string r = root;
    while (r.size() > 1 && r.back() == '/') r.pop_back();
    mi_batch_expect_host_bytes(b);                           // the reader threads set up while the walk begins
    mi_walk::walk_root(&w, r);
    w.flush_pending();
    if (w.rc) {
        if (!w.err.empty()) mi_set_error(b, w.err.c_str());
        return w.rc;
    }
    if (n_entries) *n_entries = t->entries.size();
    return MI_OK;
}

static void fill_entries(const Tree* t, mi_tree_entry* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        const mi_walk::Entry& e = t->entries[i];
        out[i].relpath = e.relpath.c_str();
        out[i].link_target = e.has_link ? e.link.c_str() : nullptr;
        out[i].file_index = e.file_index;
        out[i].size = e.size;
        out[i].mtime_sec = e.mtime;
        out[i].mode = e.mode;
        out[i].kind = e.kind;
        out[i].uid = e.uid;
        out[i].gid = e.gid;
    }
}

// ---- the walk on its own (no ctx, no GPU): what the engine WOULD add, in which order -------
int mi_tree_walk(const char* root, const char* rel_base, const char* const* blacklist,
                 uint64_t n_blacklist, uint32_t mode, mi_tree** out, uint64_t* n_entries) {
    if (!root || !out || (n_blacklist && !blacklist) || mode > MI_TREE_SCAN) return MI_ERR_INVALID;
    Tree* t = new Tree();
    mi_walk::Walker w;
    w.batch = nullptr;
    w.rel_base = rel_base ? rel_base : root;
    w.mode = mode;
    w.tree = t;
    for (uint64_t i = 0; i < n_blacklist; ++i) w.blacklist.push_back(blacklist[i]);
    std::string r = root;
    while (r.size() > 1 && r.back() == '/') r.pop_back();
    mi_walk::walk_root(&w, r);
    if (w.rc) { delete t; return w.rc; }
    *out = (mi_tree*)t;
    if (n_entries) *n_entries = t->entries.size();
    return MI_OK;
}

int mi_tree_entries(const mi_tree* tree, mi_tree_entry* out, uint64_t cap) {
    if (!tree || (!out && cap)) return MI_ERR_INVALID;
    const Tree* t = (const Tree*)tree;
    if (cap < t->entries.size()) return MI_ERR_CAPACITY;
    fill_entries(t, out, t->entries.size());
    return MI_OK;
}

void mi_tree_free(mi_tree* tree) { delete (Tree*)tree; }

int mi_batch_tree_entries(mi_batch* b, mi_tree_entry* out, uint64_t cap) {
    if (!b || (!out && cap)) return MI_ERR_INVALID;
    Tree* t = (Tree*)*mi_batch_tree_slot(b);
    const uint64_t n = t ? t->entries.size() : 0;
    if (cap < n) { mi_set_error(b, "tree entry buffer too small"); return MI_ERR_CAPACITY; }
    if (n) fill_entries(t, out, n);
    return MI_OK;
}

int mi_context_checksum_tree(mi_batch* b, const void* prefix, uint64_t prefix_len, uint32_t* crc_out) {
    if (!b || !crc_out) return MI_ERR_INVALID;
    Tree* t = (Tree*)*mi_batch_tree_slot(b);
    std::vector<mi_ctx_entry> es(t ? t->entries.size() : 0);
    for (size_t i = 0; i < es.size(); ++i) {
        const mi_walk::Entry& e = t->entries[i];
        es[i].relpath = e.relpath.c_str();
        es[i].link_target = e.has_link ? e.link.c_str() : nullptr;
        es[i].file_index = e.file_index;
    }
    return mi_context_checksum(b, prefix, prefix_len, es.data(), es.size(), crc_out);
}

void mi_batch_tree_free(void* tree) { delete (Tree*)tree; }

// The order MemFS.commitLayer writes a layer's entries in: memLayer.rangeFiles
// (lib/snapshot/mem_layer.go:232-244) sort.Strings the map keys, and addHeader (:190-211) keys
// an entry by its absolute dst path -- a whiteout marker ".wh.<name>" by the path of the file it
// deletes (dir + name without the prefix).  This is NOT the walk order ("a-b" sorts before
// "a/x": '-' < '/'), so the shim needs it to line the engine's per-file results up with the tar
// stream.  order_out[k] = index of the k-th entry to commit; equal keys keep their input order.
// How: the keys are laid out in contiguous arenas (no string per entry), and the order is a STABLE NATURAL MERGE SORT of
// the indices -- maximal non-decreasing runs are found first and merged pairwise.  A walk hands its entries over almost
// sorted (filepath.Walk order differs from sort.Strings only where a name holds a byte below '/': "a.txt" before "a/x"),
// so there are few runs and the cost is a few passes instead of log2(n): 100 000 walk-ordered entries 50 -> 11 ms, a
// million 600 -> 50-130 ms, ten million 0.6-1.4 s (more directories: more runs; 8 cores of this container); shuffled input
// 110 -> 50 ms per 100 000, 2.7 -> 0.19 s per million.  From 131 072 entries on, blocks of the input are
// keyed and sorted by up to 16 threads (MI_WALK_THREADS) and merged pairwise (C4 names 10 M entries).
namespace {
using mi_walk::KeyRef;
using mi_walk::key_less;
using mi_walk::natural_merge_sort;
// the key of one entry appended to `arena`: AbsPath(dst), a whiteout marker under the path it deletes
inline void append_key(const char* rp, std::string* arena) {
    rp = rp ? rp : "";
    if (strcmp(rp, ".") == 0) rp = "";
    *arena += *rp ? mi_walk::abs_path_of_rel(rp) : mi_walk::abs_path("");
    const size_t cut = arena->find_last_of('/');               // the key starts with '/': the last one lies in THIS key
    if (arena->compare(cut + 1, 4, ".wh.") == 0) arena->erase(cut + 1, 4);
}
}  // namespace

int mi_entries_commit_order(const mi_tree_entry* entries, uint64_t n, uint64_t* order_out) {
    if ((n && !entries) || (n && !order_out)) return MI_ERR_INVALID;
    if (n == 0) return MI_OK;
    const uint64_t kBlockMin = 65536;
    unsigned nt = (unsigned)std::min<uint64_t>(mi_walk::walk_threads(), n / kBlockMin);
    if (nt < 1) nt = 1;
    std::vector<uint64_t> bounds(nt + 1);
    for (unsigned t = 0; t <= nt; ++t) bounds[t] = n * t / nt;
    std::vector<KeyRef> keys((size_t)n);
    std::vector<std::string> arenas(nt);
    std::vector<uint64_t> tmp((size_t)n);
    auto block = [&](unsigned t) {                               // keys of the block, then the block sorted in place
        const uint64_t lo = bounds[t], hi = bounds[t + 1];
        std::string& arena = arenas[t];
        std::vector<uint64_t> at((size_t)(hi - lo) + 1);
        for (uint64_t i = lo; i < hi; ++i) { at[(size_t)(i - lo)] = arena.size(); append_key(entries[i].relpath, &arena); }
        at[(size_t)(hi - lo)] = arena.size();
        for (uint64_t i = lo; i < hi; ++i) {
            keys[(size_t)i].p = arena.data() + at[(size_t)(i - lo)];
            keys[(size_t)i].len = (uint32_t)(at[(size_t)(i - lo) + 1] - at[(size_t)(i - lo)]);
            order_out[i] = i;
        }
        natural_merge_sort(order_out + lo, tmp.data() + lo, (size_t)(hi - lo), keys.data());
    };
    if (nt == 1) { block(0); return MI_OK; }
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(block, t);
        block(0);
        for (auto& x : th) x.join();
    }
    auto less = [&keys](uint64_t a, uint64_t b) { return key_less(keys[(size_t)a], keys[(size_t)b]); };
    uint64_t *src = order_out, *dst = tmp.data();
    while (bounds.size() > 2) {                                  // sorted blocks merged pairwise, a thread per pair
        std::vector<uint64_t> next;
        next.push_back(0);
        std::vector<std::thread> th;
        size_t r = 0;
        for (; r + 2 < bounds.size(); r += 2) {
            const uint64_t a = bounds[r], b = bounds[r + 1], c = bounds[r + 2];
            th.emplace_back([=] { std::merge(src + a, src + b, src + b, src + c, dst + a, less); });
            next.push_back(c);
        }
        if (r + 1 < bounds.size()) {
            std::copy(src + bounds[r], src + bounds[r + 1], dst + bounds[r]);
            next.push_back(bounds[r + 1]);
        }
        for (auto& x : th) x.join();
        std::swap(src, dst);
        bounds.swap(next);
    }
    if (src != order_out) std::copy(src, src + n, order_out);
    return MI_OK;
}

extern "C" int mi_entry_similar(const mi_tree_entry* a, const mi_tree_entry* b, int ignore_time,
                                const uint8_t* root_a, const uint8_t* root_b, int* similar);

// The layer diff of a scan, on two walks: what MemFS.createLayerByScan + maybeAddToLayer
// (lib/snapshot/mem_fs.go:315-341, 440-480) decide entry by entry against the in-memory tree --
//   * an entry is added when its path is new or isUpdated says so (mem_fs.go:487-503; here
//     mi_entry_similar, content-aware when both sides carry chunk roots); the root itself never is;
//   * every ancestor directory of an added entry and of a whiteout is carried along
//     (addAncestors);
//   * a path that was there and is gone gets ONE whiteout at the top of the deleted subtree, and
//     only under a parent that still is a directory (the loop over n.children of a TypeDir header).
extern "C" int mi_snapshot_diff(const mi_snapshot_side* before, const mi_snapshot_side* after, int ignore_time,
                                uint8_t* after_flags, uint8_t* before_whiteout) {
    if (!before || !after || (before->n && !before->entries) || (after->n && !after->entries) ||
        (after->n && !after_flags) || (before->n && !before_whiteout))
        return MI_ERR_INVALID;
    auto root_of = [](const mi_snapshot_side* s, const mi_tree_entry& e) -> const uint8_t* {
        if (!s->roots || e.kind != 1 || e.file_index < 0) return nullptr;
        return (const uint8_t*)s->roots + (uint64_t)e.file_index * s->root_stride;
    };
    // path -> index, the LAST entry of a path winning.  The paths of a side lie in one arena; the table is flat -- a slot holds the
    // path's 64-bit hash and its entry index, equal hashes are settled on the arena's bytes -- so that an insert or a lookup is
    // one cache miss and no allocation (a node-based map: two misses and a malloc per path; at ten million entries a side the
    // diff spent three quarters of its time there: 0.87 -> see profiles/r05_host_scale.txt).
    struct PathTable {
        std::string arena;
        std::vector<uint64_t> off;                               // [i], [i + 1]: entry i's path in the arena
        std::vector<uint64_t> hash, index;                       // open addressing; index ~0 = free
        std::vector<uint8_t> superseded;                         // [i] = 1: entry i's path occurs again later (that one counts)
        uint64_t mask = 0;
        static uint64_t hash_of(std::string_view p) {
            uint64_t h = 0x9E3779B97F4A7C15ull ^ p.size();
            const char* d = p.data();
            size_t n = p.size();
            while (n >= 8) { uint64_t w; memcpy(&w, d, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; d += 8; n -= 8; }
            uint64_t w = 0;
            memcpy(&w, d, n);
            h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
            h ^= h >> 29; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 32;
            return h;
        }
        std::string_view path(uint64_t i) const { return std::string_view(arena.data() + off[(size_t)i], (size_t)(off[(size_t)i + 1] - off[(size_t)i])); }
        void build(const mi_snapshot_side* side) {
            off.resize((size_t)side->n + 1);
            std::string p;
            for (uint64_t i = 0; i < side->n; ++i) {
                mi_walk::abs_path_of_rel_into(side->entries[i].relpath ? side->entries[i].relpath : "", &p);
                off[(size_t)i] = arena.size();
                arena.append(p);
            }
            off[(size_t)side->n] = arena.size();
            uint64_t cap = 16;
            while (cap < side->n * 2) cap <<= 1;
            mask = cap - 1;
            hash.assign((size_t)cap, 0);
            index.assign((size_t)cap, ~0ull);
            superseded.assign((size_t)side->n, 0);
            for (uint64_t i = 0; i < side->n; ++i) {             // (the arena is complete: its views stay where they are)
                const std::string_view p_i = path(i);
                const uint64_t h = hash_of(p_i);
                uint64_t s = h & mask;
                for (; index[(size_t)s] != ~0ull; s = (s + 1) & mask)
                    if (hash[(size_t)s] == h && path(index[(size_t)s]) == p_i) {           // the same path again: the last one wins
                        superseded[(size_t)index[(size_t)s]] = 1;
                        break;
                    }
                hash[(size_t)s] = h;
                index[(size_t)s] = i;
            }
        }
        uint64_t find(std::string_view p) const {                // ~0: not there
            if (index.empty()) return ~0ull;
            const uint64_t h = hash_of(p);
            for (uint64_t s = h & mask; index[(size_t)s] != ~0ull; s = (s + 1) & mask)
                if (hash[(size_t)s] == h && path(index[(size_t)s]) == p) return index[(size_t)s];
            return ~0ull;
        }
    };
    PathTable old_at, new_at;
    old_at.build(before);
    new_at.build(after);
    for (uint64_t i = 0; i < after->n; ++i) after_flags[i] = MI_DIFF_SAME;
    for (uint64_t i = 0; i < before->n; ++i) before_whiteout[i] = 0;
    static const std::string_view kRoot("/");
    auto parent_of = [](std::string_view p) -> std::string_view {          // path.Dir of a clean absolute path; "" = none
        const size_t cut = p.rfind('/');
        if (cut == std::string_view::npos) return std::string_view();
        return cut == 0 ? kRoot : p.substr(0, cut);
    };
    auto carry_ancestors = [&](std::string_view p) {
        for (std::string_view d = parent_of(p); !d.empty() && d != kRoot; d = parent_of(d)) {
            const uint64_t at = new_at.find(d);
            if (at != ~0ull) {
                // an ancestor that is already carried (or changed) has had ITS ancestors carried then: the chain is done
                if (after_flags[at] != MI_DIFF_SAME) return;
                after_flags[at] = MI_DIFF_ANCESTOR;
            }
        }
    };
    // the entries in the order they came (the memory the caller handed over is walked once, front to back); an entry whose path
    // occurs again later is not the one the table holds and decides nothing (flags are only raised, SAME -> ANCESTOR -> CHANGED:
    // the order the paths are visited in decides nothing either)
    std::vector<uint8_t> still_there((size_t)before->n, 0);   // [j] = 1: the path of `before`'s entry j is one of `after`'s
    for (uint64_t i = 0; i < after->n; ++i) {
        const std::string_view p = new_at.path(i);
        if (p == kRoot) continue;                            // "Root itself is not added to layers"
        if (new_at.superseded[(size_t)i]) continue;
        const mi_tree_entry& e = after->entries[i];
        bool updated = true;
        const uint64_t j = old_at.find(p);
        if (j != ~0ull) {
            still_there[(size_t)j] = 1;
            const mi_tree_entry& o = before->entries[j];
            int similar = 0;
            int rc = mi_entry_similar(&o, &e, ignore_time, root_of(before, o), root_of(after, e), &similar);
            if (rc) return rc;                               // "unsupported type"
            updated = !similar;
        }
        if (updated) {
            after_flags[i] = MI_DIFF_CHANGED;
            carry_ancestors(p);
        }
    }
    for (uint64_t j = 0; j < before->n; ++j) {
        const std::string_view p = old_at.path(j);
        if (still_there[(size_t)j] || old_at.superseded[(size_t)j] || p == kRoot) continue;    // (almost every entry: no lookup)
        if (new_at.find(p) != ~0ull) continue;               // (the root; a path `after` lists only in a superseded entry)
        const uint64_t par = new_at.find(parent_of(p));
        if (par == ~0ull || after->entries[par].kind != 0) continue;                  // deeper in a deleted subtree,
                                                                                       // or its parent became a file
        if (after->disk_root) {
            // child.isOnDisk() (mem_fs.go:49-57, 466): a path the walk no longer lists because it is now
            // skipped (a new mountpoint, a blacklisted dir) is still on disk and gets NO whiteout
            const std::string on_disk = std::string(after->disk_root) + std::string(p);
            struct stat st;
            if (lstat(on_disk.c_str(), &st) == 0) continue;
            if (errno != ENOENT && errno != ENOTDIR) return MI_ERR_IO;           // "check on disk"
        }
        before_whiteout[j] = 1;
        carry_ancestors(p);
    }
    return MI_OK;
}

// tario.IsSimilarHeader (lib/tario/compare.go:24-117) on walk entries, plus the optional
// content roots (see the header)
int mi_entry_similar(const mi_tree_entry* a, const mi_tree_entry* b, int ignore_time,
                     const uint8_t* root_a, const uint8_t* root_b, int* similar) {
    if (!a || !b || !similar) return MI_ERR_INVALID;
    auto empty = [](const char* s) { return !s || !*s; };
    auto same_str = [&](const char* x, const char* y) { return strcmp(x ? x : "", y ? y : "") == 0; };
    *similar = 0;
    if (empty(a->relpath) && empty(b->relpath)) { *similar = 1; return MI_OK; }   // "/" is never modified
    if (a->kind > 3) return MI_ERR_INVALID;                                       // unsupported type
    if (a->kind != b->kind) return MI_OK;
    const bool time_eq = ignore_time || a->mtime_sec == b->mtime_sec;
    const bool owner_mode_eq = a->uid == b->uid && a->gid == b->gid && (a->mode & 07777u) == (b->mode & 07777u);
    switch (a->kind) {
    case 2: *similar = same_str(a->link_target, b->link_target); break;
    case 3:                                       // UpdateFromTarReader stores AbsPath(Linkname) (mem_fs.go:214-216)
        *similar = time_eq && owner_mode_eq &&
                   mi_walk::abs_path(a->link_target ? a->link_target : "") ==
                       mi_walk::abs_path(b->link_target ? b->link_target : "");
        break;
    case 0: *similar = time_eq && owner_mode_eq; break;
    default:
        *similar = time_eq && owner_mode_eq && a->size == b->size &&
                   (!(root_a && root_b) || memcmp(root_a, root_b, 32) == 0);
    }
    return MI_OK;
}

}  // extern "C"
