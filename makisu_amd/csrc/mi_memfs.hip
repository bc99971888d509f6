// mi_memfs.hip -- the reference's MemFS and what surrounds a COPY/ADD step, on the host (no device code here):
//   * MemFS as a handle -- the one implementation of the layer merge and the copy-op layer (MemFS.addToLayer):
//     UpdateFromTarReader with and without untar, AddLayerByScan, AddLayerByCopyOps, Checkpoint, Reset;
//   * step.commitLayer in ONE call, mi_memfs_commit_layer -- with a ctx THE SEAM of this library: the walk stages
//     every file into one batch, the GPU cuts and hashes them, the diff runs with chunk roots (a path is in the layer if its
//     header changed OR its content did), the layer tar is framed from the bytes in HBM; the scan runs on a thread of its own
//     beside the diff and the writer (ScanJob), and with MI_MEMFS_TRUST_CTIME only files whose inode changed are read at all;
//   * CopyOperation.Execute over fileio.Copier; MemFS.untarOneItem + tario.ApplyHeader;
//   * the caller's side of the step: --chown (utils.ResolveChown), source patterns (filepath.Match / Glob as
//     addCopyStep.resolveFromPaths uses them), NewCopyOperation's checks.
// Every function cites the Go it restates; the tree they share is mi_memtree.h, the walks are mi_tree.hip's.
#include "mi_memtree.h"
#include "../../include/makisu_mi_host.h"      // the optional helpers defined here (Execute, chown, glob, untar, checkpoint)

#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <ftw.h>
#include <grp.h>
#include <pwd.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory_resource>
#include <mutex>
#include <thread>
#include <new>
#include <chrono>
#include <atomic>
#include <unordered_map>
#include <unordered_set>



// ---- MemFS.AddLayerByCopyOps: the layer a COPY / ADD step creates, on entry lists -----------------
// addToLayer + maybeAddToLayer(createWhiteout = false) + addAncestors + isUpdated + createHeader
// (lib/snapshot/mem_fs.go:276-289, 343-421, 440-503, 505-566; mem_layer.go:152-190) and
// CopyOperation's source resolution (copy_op.go, utils.go:249-327), restated on a path-keyed tree:
//   * a single non-directory source copies onto dst (or into dst + "/" + base when dst ends with "/");
//     otherwise dst is ensured to exist -- every EXISTING ancestor is carried into the layer, a
//     symlink on the way is followed (its target joined with the rest of the path, the reference's
//     own rule, depth-limited), missing directories are created with the last existing ancestor's
//     mode, the op's uid/gid and mtime = now -- and the sources' CONTENTS are copied below it;
//   * sources are resolved through symlinks inside src_root (a link leaving the root is an error)
//     and walked like every snapshot walk (".wh..wh." names, special files and mountpoints skipped;
//     no blacklist: the reference passes nil here);
//   * every walked path gets createHeader's header with the op's uid/gid and is added iff isUpdated
//     says so (tario.IsSimilarHeader against what the tree holds); adding a path first carries its
//     existing ancestors, then replaces the node: a directory keeps the old node's children, anything
//     else drops them.
// Result: the layer's entries in commit order (sorted by dst), each with the path its content is
// read from ("/" for directories the op created: memLayer.addHeader("", ...) -> AbsPath("")).  The caller's tree is not modified; to continue,
// apply the layer to it with mi_entries_apply_layer.
namespace mi_copy {

struct Node {
    mi_walk::Entry e;          // relpath = dst without the leading "/"
    std::string src;           // where the content is read from; what memFSNode.isOnDisk looks at
    bool has_root = false;     // chunk root of the content, when the caller scanned it (content-aware isUpdated)
    uint8_t root[32];
    int64_t batch_file = -1;   // a content-aware commit under way: the file's row in the commit's batch -- its bytes lie in HBM
    uint32_t batch_gen = 0;    // ... of WHICH commit (Fs::commit_gen): the tree keeps its nodes, a later commit that puts one back into
                               // its layer "as it is" (addAncestors over a path the tree holds as a file) must not take the old row
                               // for a row of its own batch -- it reads that file from its source path, as the reference does
    bool root_pending = false; // ... and its root is still being computed (a pipelined commit: ScanJob); root[] is not valid yet
};
// MI_MEMFS_TRUST_CTIME, per node whose root a commit of this handle computed: the inode as it was when the content was read, and
// when that was (CLOCK_REALTIME at the start of that commit's walk).  Beside the nodes (Fs::hashed, by node index), not in them:
// a tree merged from base layers and never scanned with a ctx pays nothing for it.
struct HashedAs { mi_walk::InodeStamp stamp; int64_t at_ns = 0; };
// a scan's mark ("this scan's walk lists the path", mi_memtree::Node::seen): never 0, never given twice
static uint32_t next_scan_mark() {
    static std::atomic<uint32_t> g{0};
    uint32_t m = ++g;
    if (m == 0) m = ++g;
    return m;
}

// The GPU scan of a pipelined commit, on a thread of its own: mi_batch_run (the end of staging, the kernels) and the roots'
// way back, while the committing thread computes the layer and frames the tar from the bytes that have already landed.
// Whoever needs a root before the scan is through waits for it (Fs::root_now); everybody else is handed the roots at the end.
struct ScanJob {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int rc = MI_OK;
    std::string err;
    std::vector<uint8_t> roots;
    uint64_t n_chunks = 0;
    double seconds = 0;
    void start(mi_batch* b, uint64_t n_files) {
        roots.resize(n_files * 32);
        th = std::thread([this, b, n_files] {
            const auto t0 = std::chrono::steady_clock::now();
            int r = mi_batch_run(b);
            if (!r) r = mi_batch_roots(b, roots.data(), n_files);
            uint64_t nc = 0;
            if (!r) mi_batch_counts(b, nullptr, &nc, nullptr);
            std::string e = r ? mi_last_error_of_batch(b) : "";          // (a locked copy: the committing thread may be failing too)
            {
                std::lock_guard<std::mutex> g(mu);
                rc = r;
                err = std::move(e);
                n_chunks = nc;
                seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                done = true;
            }
            cv.notify_all();
        });
    }
    const uint8_t* wait_roots() {                              // nullptr: the scan failed (rc, err)
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return rc ? nullptr : roots.data();
    }
    void join() { if (th.joinable()) th.join(); }
    ~ScanJob() { join(); }
};
// memLayer.files: path -> header.  A merged base image puts a million keys in and throws them away when the merge is done
// (only their number is reported): from the C library's heap that is a million small allocations, a million frees, and --
// some unrelated allocation later -- a sweep over a million free fragments (measured: 0.3 s added to the next scan that
// has more than a handful of changes).  So keys and table live in a pool that grows in large blocks and is given back whole.
// A MERGE (UpdateFromTarReader) never looks at the map again: "Merged %d headers from tar to memfs" is all that is left of
// it.  In count-only mode the map is therefore a set of 128-bit fingerprints of its keys (two independent 64-bit hashes,
// open addressing, 16 bytes a key instead of ~130 with two allocations): the number of distinct keys, exact unless two
// of a layer's paths collide in 128 bits (10^7 keys: 10^-25).  At C4's ten million entries the full map was 1.3 GB of
// first-touched memory and a microsecond per header (profiles/r05_host_scale.txt).
struct KeyCounter {
    struct Fp { uint64_t a, b; };
    std::vector<Fp> tab;                                                            // a == 0 && b == 0: free
    size_t n = 0;
    static Fp fp(const char* p, size_t len) {
        uint64_t a = 0x9E3779B97F4A7C15ull ^ len, b = 0xC2B2AE3D27D4EB4Full + len;
        while (len >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            a = (a ^ w) * 0xFF51AFD7ED558CCDull; a ^= a >> 32;
            b = (b + w) * 0xC4CEB9FE1A85EC53ull; b = (b << 29) | (b >> 35);
            p += 8; len -= 8;
        }
        uint64_t w = 0;
        memcpy(&w, p, len);
        a = (a ^ w) * 0xFF51AFD7ED558CCDull; a ^= a >> 33; a *= 0xC4CEB9FE1A85EC53ull; a ^= a >> 29;
        b = (b + w) * 0x9FB21C651E98DF25ull; b ^= b >> 31; b *= 0xFF51AFD7ED558CCDull; b ^= b >> 32;
        if (!(a | b)) a = 1;
        return {a, b};
    }
    void reserve(size_t keys) {
        size_t want = 16;
        while (want < keys * 2) want <<= 1;
        if (want > tab.size()) rehash(want);
    }
    void rehash(size_t cap) {
        std::vector<Fp> old;
        old.swap(tab);
        tab.assign(cap, Fp{0, 0});
        for (const Fp& f : old) if (f.a | f.b) place(f);
    }
    void place(const Fp& f) {
        size_t i = (size_t)f.a & (tab.size() - 1);
        while (tab[i].a | tab[i].b) i = (i + 1) & (tab.size() - 1);
        tab[i] = f;
    }
    void add(const char* p, size_t len) {
        if ((n + 1) * 2 > tab.size()) rehash(tab.empty() ? 1024 : tab.size() * 2);
        const Fp f = fp(p, len);
        size_t i = (size_t)f.a & (tab.size() - 1);
        for (; tab[i].a | tab[i].b; i = (i + 1) & (tab.size() - 1))
            if (tab[i].a == f.a && tab[i].b == f.b) return;
        tab[i] = f;
        ++n;
    }
    void clear() { std::vector<Fp>().swap(tab); n = 0; }
};

struct LayerMap {
    using Map = std::pmr::unordered_map<std::pmr::string, int64_t>;
    std::pmr::monotonic_buffer_resource pool;
    alignas(Map) unsigned char store[sizeof(Map)];
    Map* m;
    bool touched = false;
    bool count_only = false;                                                        // a merge under way: see KeyCounter
    KeyCounter counter;
    LayerMap() : m(new (store) Map(&pool)) {}
    ~LayerMap() { m->~Map(); }
    LayerMap(const LayerMap&) = delete;
    LayerMap& operator=(const LayerMap&) = delete;
    void set(const std::string& key, int64_t ref) {                                // l.files[key] = ...
        if (count_only) { counter.add(key.data(), key.size()); return; }
        touched = true;
        auto it = m->find(std::pmr::string(key.data(), key.size(), &scratch));
        scratch.release();
        if (it != m->end()) { it->second = ref; return; }
        auto at = m->emplace(std::piecewise_construct, std::forward_as_tuple(key.data(), key.size()), std::forward_as_tuple(ref)).first;
        seq.push_back(&*at);                                                       // (a table's elements do not move)
    }
    std::vector<const Map::value_type*> seq;                                       // the keys in the order they came
    size_t size() const { return count_only ? counter.n : m->size(); }
    void reserve(size_t n) { if (count_only) { counter.reserve(n); return; } touched = true; m->reserve(n); }
    void clear() {
        counter.clear();
        count_only = false;
        if (!touched) return;
        m->~Map();
        pool.release();
        m = new (store) Map(&pool);
        seq.clear();
        touched = false;
    }
private:
    unsigned char scratch_buf[512];
    std::pmr::monotonic_buffer_resource scratch{scratch_buf, sizeof scratch_buf};  // the lookup's key: never the pool's
};

struct Fs {
    mi_memtree::Tree t;                      // fs.tree; a node's ref indexes `nodes`
    std::vector<Node> nodes;
    LayerMap layer;                          // memLayer.files: keyed by dst -- by the DELETED path for a ".wh." name;
                                             // sorted when the layer is taken (rangeFiles, mem_layer.go:232-244)
    std::vector<Node> sorted_layer() const {                                        // sort.Strings on the keys: they came in
        const size_t n = layer.seq.size();                                          // walk order, which is nearly sorted
        std::vector<mi_walk::KeyRef> keys(n);
        std::vector<uint64_t> idx(n), tmp(n);
        for (size_t i = 0; i < n; ++i) {
            keys[i].p = layer.seq[i]->first.data();
            keys[i].len = (uint32_t)layer.seq[i]->first.size();
            idx[i] = i;
        }
        mi_walk::natural_merge_sort(idx.data(), tmp.data(), n, keys.data());
        std::vector<Node> out;
        out.reserve(n);
        for (size_t i = 0; i < n; ++i) out.push_back(nodes[layer.seq[idx[i]]->second]);
        return out;
    }
    std::string root;                      // fs.tree.src
    // the scan under way: "its walk lists this path".  Nodes the tree held when the scan began carry the answer as a
    // mark (Node::seen == scan_mark, set in one pass over the walk); for a node made DURING the scan the set of the walk's
    // paths is asked -- built the first time that happens, which a scan of an unchanged tree never sees
    uint32_t scan_mark = 0;                                          // 0 = no scan under way
    std::function<bool(const std::string&)> listed_by_walk;          // the fallback
    int64_t now = 0;
    std::string err;
    int rc = MI_OK;

    bool fail(int code, const std::string& m) { if (!rc) { rc = code; err = m; } return false; }
    int64_t keep(Node n) { nodes.push_back(std::move(n)); return (int64_t)nodes.size() - 1; }

    Fs() {
        t.on_add = [this](const std::string& dst, int64_t ref) {               // memLayer.addHeader (mem_layer.go:197-212)
            if (ref < 0) {                                                      // a parent the caller's list left out
                Node d;
                d.e.mode = (uint32_t)(S_IFDIR | 0755); d.e.kind = 0; d.e.relpath = dst.substr(1); d.e.mtime = now;
                d.src = "/";
                ref = keep(d);
                if (mi_memtree::Node* n = t.find(dst)) n->ref = ref;
            }
            const size_t cut = dst.find_last_of('/');
            if (dst.compare(cut == std::string::npos ? 0 : cut + 1, 4, ".wh.") == 0) {
                const std::string name = mi_walk::base_of(dst);
                const std::string dir = mi_walk::dir_of(dst);
                layer.set((dir == "/" ? "" : dir) + "/" + name.substr(4), ref);
            } else {
                layer.set(dst, ref);
            }
        };
        t.make_dir = [this](const std::string& dst, const mi_memtree::Node& last_ancestor, uint32_t uid, uint32_t gid) {
            Node d;                                                             // mem_fs.go:551-559
            d.e.mode = last_ancestor.ref >= 0 ? nodes[last_ancestor.ref].e.mode : (uint32_t)(S_IFDIR | 0755);
            d.e.kind = 0;
            d.e.relpath = dst.substr(1);
            d.e.mtime = now;
            d.e.uid = uid;
            d.e.gid = gid;
            d.src = "/";                       // l.addHeader("", curr, hdr): src = AbsPath("") -- isOnDisk says yes, always
            return keep(d);
        };
    }
    // addAncestors (mem_fs.go:505-566); returns the resolved dst
    // Entries arrive directory by directory, and addAncestors of a second path below the same parent repeats the first
    // one's work to the letter when nothing of the chain has changed in between: the same existing directories go
    // through addHeader again (layer[path] = the same header), nothing is created, nothing is cleared.  So the parent of
    // the last call is remembered -- while its chain held directories only (no symlink to follow, no file in the way),
    // the tree has not changed shape since (a leaf put below that parent keeps the memo: it is nobody's ancestor) and
    // the layer map has not been emptied -- and such a call returns at once (a merge of 10^6 entries: 3.4 -> 1.6 us each, with the kept parent node of mi_memtree.h).
    struct { bool valid = false; std::string parent; uint64_t gen = 0; } anc_memo;
    uint64_t n_anc_calls = 0, n_anc_memo = 0;                                   // MI_MEMFS_TIMING
    // content-aware isUpdated, counted per layer (mi_commit_stats): files whose header tario.IsSimilarHeader calls similar
    // and whose chunk roots differ; unchanged files whose node had no root yet and took the scan's
    uint64_t n_content_changed = 0, n_roots_learned = 0;
    const mi_walk::InodeStamp* stamp_for_next_keep = nullptr;                   // memfs_scan -> maybe_add: the new node's inode stamp
    std::vector<HashedAs> hashed;                                               // [node index]; shorter than `nodes`: no record beyond it
    void record_hashed(int64_t ref, const mi_walk::InodeStamp& st) {
        if ((size_t)ref >= hashed.size()) hashed.resize(nodes.size());
        hashed[(size_t)ref].stamp = st;
        hashed[(size_t)ref].at_ns = commit_started_ns;
    }
    const uint64_t id = [] { static std::atomic<uint64_t> n{0}; return ++n; }();    // (a handle's number: never given twice)
    uint32_t reserved_mark = 0;                                                 // the coming scan's mark, taken before its walk (0: none)
    uint32_t commit_gen = 0;                                                    // which mi_memfs_commit_layer call of this handle is under way (Node::batch_gen)
    bool trust_ctime = false;                                                   // mi_memfs_set_options(MI_MEMFS_TRUST_CTIME)
    int64_t commit_started_ns = 0;                                              // CLOCK_REALTIME when the commit under way began its walk
    // "is the content of the regular file at `disk_path` known?" -- asked by the walk's directory readers, several at a time,
    // while nothing changes the tree (the diff runs after the walk).  Known = the tree holds the path with a root that was
    // computed from THIS inode in THIS state: same device, inode, size, mtime and ctime to the nanosecond -- and the ctime lies
    // safely before the moment the content was read (a write in the same clock tick as the recorded ctime would leave no
    // trace: git's "racily clean" rule; MI_TRUST_CTIME_SLACK_MS, default 20, covers the kernel's coarse clock).
    // Returns mi_walk::kContentKnown, and with it kEntryHeld when the file's header is also what the node holds (tario.IsSimilarHeader
    // on a regular file: mode bits, owner, size, whole-second mtime) -- such an entry is nothing to the diff: the node is marked
    // "listed by this scan's walk" HERE (`mark`: the scan's mark, reserved before the walk), by the reader that stat'ed the file,
    // and memfs_scan passes it over.  A million unchanged files are then compared by sixteen readers beside their fstatat, not by
    // the committing thread after the walk.
    uint8_t content_is_known(const std::string& disk_path, const struct stat& sb, const mi_walk::InodeStamp& st, uint32_t mark) {
        const uint64_t size = (uint64_t)sb.st_size;
        static const int64_t slack_ns = [] { const char* e = getenv("MI_TRUST_CTIME_SLACK_MS"); return (int64_t)(e && *e ? atol(e) : 20) * 1000000ll; }();
        const size_t root_len = root == "/" ? 0 : root.size();
        if (disk_path.size() <= root_len || memcmp(disk_path.data(), root.data(), root_len) != 0) return 0;
        // A directory reader asks about ONE directory's files, in name order -- the order of the node's children map.  So the
        // directory's node and a position among its children are kept PER THREAD (the tree's own kept parent belongs to the
        // committing thread): the next file is the next child, or a few steps on -- not a walk from the root and a search among
        // thousands of siblings whose map nodes no cache holds (2.4 us per file, summed over the readers, before; the walk of a
        // million trusted files cost 0.12 s more than the plain walk).  Nothing changes the tree while the walk runs.
        using Kids = decltype(mi_memtree::Node::children);
        struct DirMemo { uint64_t fs_id = 0, gen = 0; uint32_t mark = 0; std::string dir; const mi_memtree::Node* node = nullptr; Kids::const_iterator at; };   // (mark: a memo never outlives the walk it was made in)
        static thread_local DirMemo memo;
        const size_t cut = disk_path.find_last_of('/');
        const mi_memtree::Node* nd = nullptr;
        if (cut != std::string::npos && cut >= root_len && cut + 1 < disk_path.size()) {
            const size_t dir_len = cut - root_len;                                // the directory below the root ("" = the root itself)
            if (!(memo.fs_id == id && memo.gen == t.gen && memo.mark == mark && memo.dir.size() == dir_len &&
                  memcmp(memo.dir.data(), disk_path.data() + root_len, dir_len) == 0)) {
                memo.dir.assign(disk_path, root_len, dir_len);
                memo.node = t.find_walk(memo.dir);                                // (find_walk keeps no cache: safe from many threads)
                memo.fs_id = id;
                memo.gen = t.gen;
                memo.mark = mark;
                if (memo.node) memo.at = memo.node->children.begin();
            }
            if (memo.node) {
                const std::string_view name(disk_path.data() + cut + 1, disk_path.size() - cut - 1);
                const Kids& kids = memo.node->children;
                int steps = 0;
                while (memo.at != kids.end() && steps < 8 && std::string_view(memo.at->first) < name) { ++memo.at; ++steps; }
                if (memo.at == kids.end() || std::string_view(memo.at->first) != name) memo.at = kids.lower_bound(name);
                if (memo.at != kids.end() && std::string_view(memo.at->first) == name) { nd = memo.at->second.get(); ++memo.at; }
            }
        } else {
            nd = t.find_walk(disk_path.substr(root_len));
        }
        if (!nd || nd->ref < 0) return 0;
        const Node& x = nodes[(size_t)nd->ref];
        if (x.e.kind != 1 || !x.has_root || x.root_pending || x.e.size != size || (size_t)nd->ref >= hashed.size()) return 0;
        const HashedAs& h = hashed[(size_t)nd->ref];
        // a file system that keeps whole seconds (ext3, FAT, some network mounts) shows as a ctime without a sub-second part:
        // its racy window is the second (two for FAT), not the kernel's clock tick
        const int64_t window_ns = st.ctime_ns % 1000000000ll == 0 ? std::max<int64_t>(slack_ns, 2000000000ll) : slack_ns;
        if (!h.at_ns || !(h.stamp == st)) return 0;                             // (with a clock that only moves forward the next line
        if (st.ctime_ns + window_ns >= h.at_ns) return 0;                       //  alone catches every later change: its ctime is
                                                                                 //  newer than our read.  The equality is what holds
                                                                                 //  when the clock was set back in between.)  Racily
                                                                                 //  clean: read it again
        // (counted after the walk, from its record: a shared counter here is a cache line sixteen directory readers fight over)
        if (mark && x.e.mtime == (int64_t)sb.st_mtime && x.e.uid == (uint32_t)sb.st_uid && x.e.gid == (uint32_t)sb.st_gid &&
            (x.e.mode & 07777u) == ((uint32_t)sb.st_mode & 07777u)) {
            const_cast<mi_memtree::Node*>(nd)->seen = mark;                      // (this reader's directory, this node: nobody else's)
            return mi_walk::kContentKnown | mi_walk::kEntryHeld;
        }
        return mi_walk::kContentKnown;
    }
    ScanJob* job = nullptr;                                                     // a pipelined commit's scan (else roots come ready)
    std::vector<int64_t> pending_refs;                                          // nodes whose root[] is filled in when it ends
    // the root a DECISION needs, now: waits for the scan when the node's root is still on its way (nullptr + rc: it failed)
    const uint8_t* root_now(Node& x) {
        if (!x.has_root) return nullptr;
        if (x.root_pending) {
            const uint8_t* r = job ? job->wait_roots() : nullptr;
            if (!r) { fail(MI_ERR_IO, "gpu scan: " + (job ? job->err : std::string("no scan under way"))); return nullptr; }
            memcpy(x.root, r + 32 * (uint64_t)x.batch_file, 32);
            x.root_pending = false;
        }
        return x.root;
    }
    // the scan is through (roots) or has failed (nullptr): every node that was promised a root gets it, or loses the promise
    void settle_pending(const uint8_t* roots, std::vector<Node>* layer_nodes) {
        auto settle = [&](Node& x) {
            if (!x.root_pending) return;
            x.root_pending = false;
            if (roots) memcpy(x.root, roots + 32 * (uint64_t)x.batch_file, 32);
            else x.has_root = false;
        };
        for (int64_t ref : pending_refs) settle(nodes[(size_t)ref]);
        pending_refs.clear();
        if (layer_nodes) for (Node& x : *layer_nodes) settle(x);
    }
    void clear_layer() { layer.clear(); anc_memo.valid = false; }
    // where dst splits into parent and name; npos = do not memo.  (dst is AbsPath's result in every caller -- the merge,
    // the scan, the copy ops: "/", then clean elements; so its parent IS the chain addAncestors walks.)
    static size_t parent_len(const std::string& dst) {
        const size_t cut = dst.find_last_of('/');
        return (dst.size() < 2 || dst[0] != '/' || cut + 1 >= dst.size()) ? std::string::npos : cut;
    }
    std::string add_ancestors(const std::string& dst, bool inclusive, uint32_t uid, uint32_t gid) {
        const size_t plen = inclusive ? std::string::npos : parent_len(dst);
        ++n_anc_calls;
        if (plen != std::string::npos && anc_memo.valid && anc_memo.gen == t.gen && anc_memo.parent.size() == plen &&
            memcmp(anc_memo.parent.data(), dst.data(), plen) == 0) {
            ++n_anc_memo;
            return dst;
        }
        anc_memo.valid = false;
        std::string resolved = dst;
        if (!t.add_ancestors(dst, inclusive, 0, uid, gid, &resolved)) { fail(MI_ERR_INVALID, "add ancestors of " + dst + ": " + t.err); return resolved; }
        if (plen != std::string::npos && t.chain_plain && !rc) {
            anc_memo.valid = true;
            anc_memo.parent.assign(dst, 0, plen);
            anc_memo.gen = t.gen;
        }
        return resolved;
    }
    // a leaf has been put at dst (not a whiteout): below the memo's parent that changes no chain
    void leaf_put(const std::string& dst, uint64_t gen_before) {
        if (!anc_memo.valid || anc_memo.gen != gen_before) { anc_memo.valid = false; return; }
        const size_t plen = parent_len(dst);
        if (plen != std::string::npos && anc_memo.parent.size() == plen && memcmp(anc_memo.parent.data(), dst.data(), plen) == 0)
            anc_memo.gen = t.gen;
        else
            anc_memo.valid = false;
    }
    // memLayer.addWhiteout (mem_layer.go:214-228) + whiteoutMemFile.updateMemFS
    bool add_whiteout(const std::string& p) {
        const std::string name = mi_walk::base_of(p);
        if (mi_walk::has_prefix(name, ".wh.")) return fail(MI_ERR_INVALID, "add whiteout to layer " + p + ": base name contains whiteout prefix: " + p);
        const std::string dir = mi_walk::dir_of(p);
        Node w;
        w.e.kind = 1;
        w.e.mode = 0;
        w.e.relpath = ((dir == "/" ? "" : dir) + "/.wh." + name).substr(1);
        layer.set(p, keep(w));
        if (!t.wipe(p)) return fail(MI_ERR_INVALID, "update memfs with whiteout " + p + ": " + t.err);
        return true;
    }
    // maybeAddToLayer(l, src, dst, hdr, createWhiteout) (mem_fs.go:440-483)
    void maybe_add(const std::string& src, const std::string& dst, Node n, bool create_whiteout = false) {
        if (rc) return;
        bool updated = true;
        mi_memtree::Node* cur = t.find(dst);                                      // isUpdated (:487-503)
        const bool had_node = cur != nullptr;
        if (cur && cur->ref >= 0) {
            mi_tree_entry a, b;
            auto fill = [](const Node& x, mi_tree_entry* o) {
                memset(o, 0, sizeof *o);
                o->relpath = x.e.relpath.empty() ? "" : x.e.relpath.c_str();
                o->link_target = x.e.has_link ? x.e.link.c_str() : nullptr;
                o->size = x.e.size; o->mtime_sec = x.e.mtime; o->mode = x.e.mode; o->kind = x.e.kind;
                o->uid = x.e.uid; o->gid = x.e.gid; o->file_index = -1;
            };
            fill(nodes[cur->ref], &a);
            fill(n, &b);
            int similar = 0;
            Node& o = nodes[cur->ref];
            // the roots decide only when BOTH sides carry one (mi_entry_similar ignores a single one): only then may the
            // decision have to wait for a scan that is still running
            const bool both = o.has_root && n.has_root && a.kind == 1 && b.kind == 1;
            const uint8_t* ra = both ? root_now(o) : nullptr;
            const uint8_t* rb = both ? root_now(n) : nullptr;
            if (rc) return;
            if (a.kind <= 3 && b.kind <= 3 && mi_entry_similar(&a, &b, 0, ra, rb, &similar) != MI_OK) {
                fail(MI_ERR_INVALID, "check header " + dst + ": unsupported type");
                return;
            }
            updated = !similar;
            if (similar && n.has_root && !o.has_root && n.e.kind == 1 && o.e.kind == 1) {   // the first content scan of an unchanged
                Node& held = nodes[cur->ref];                                                // file: its root is known from now on
                held.has_root = true;
                if (n.root_pending) { held.root_pending = true; held.batch_file = n.batch_file; held.batch_gen = commit_gen; pending_refs.push_back(cur->ref); }
                else memcpy(held.root, n.root, 32);
                ++n_roots_learned;
            } else if (!similar && o.has_root && n.has_root && a.kind == 1 && b.kind == 1) {
                int meta = 0;
                if (mi_entry_similar(&a, &b, 0, nullptr, nullptr, &meta) == MI_OK && meta) ++n_content_changed;
            }
        }
        if (updated && dst != "/") {
            add_ancestors(dst, false, 0, 0);
            if (rc) return;
            n.src = src;
            const uint8_t kind = n.e.kind;
            const std::string link = n.e.has_link ? n.e.link : std::string();
            const int64_t kref = keep(std::move(n));
            if (nodes[(size_t)kref].root_pending) pending_refs.push_back(kref);
            if (stamp_for_next_keep) { record_hashed(kref, *stamp_for_next_keep); stamp_for_next_keep = nullptr; }
            else if ((size_t)kref < hashed.size()) hashed[(size_t)kref] = HashedAs();
            // updateMemFS walks the tree part by part (mem_layer.go:57-80): every part before the last has to be a
            // node -- of any type.  A destination spelled THROUGH a symlink therefore works one level below the
            // link (the link node takes the child) and fails deeper ("missing intermediate directory"): what
            // addAncestors created lies on the link's TARGET, and its resolved path is only used by the createDst branch
            const uint64_t gen_before = t.gen;
            if (!t.add(dst, kref, kind, link)) {
                fail(MI_ERR_INVALID, "update memfs with file " + dst + ": " + t.err);
                return;
            }
            if (dst.compare(dst.find_last_of('/') + 1, 4, ".wh.") != 0) leaf_put(dst, gen_before);   // (a ".wh." name wipes)
        }
        if (create_whiteout && n.e.kind == 0 && had_node) whiteout_missing_children(dst);
    }
    // "Handle deletions.  Note: Only one whiteout file is needed for a deleted subtree." (:460-480): the children
    // the tree holds for this directory that are no longer on disk
    void whiteout_missing_children(const std::string& dst) {
        mi_memtree::Node* dir = t.find(dst);
        if (!dir) return;
        std::vector<std::string> gone;                                      // (wiping changes the map: collect first)
        const std::string stem = dst == "/" ? "/" : dst + "/";
        const size_t root_len = root == "/" ? 0 : root.size();
        for (auto& kv : dir->children) {
            const int64_t ref = kv.second->ref;
            static const std::string no_src;
            const std::string& child_src = ref >= 0 ? nodes[ref].src : no_src;   // (two lvalues: no copy per child)
            // memFSNode.isOnDisk (:49-57) is an lstat of the node's source.  When that source is the node's own place
            // under the root and the walk of THIS scan lists it, the walk has just lstat'ed it: no second one (the
            // reference pays it for every node of the tree on every scan)
            if (scan_mark && child_src.size() == root_len + stem.size() + kv.first.size() &&
                memcmp(child_src.data(), root.data(), root_len) == 0 &&
                memcmp(child_src.data() + root_len, stem.data(), stem.size()) == 0 &&
                memcmp(child_src.data() + root_len + stem.size(), kv.first.data(), kv.first.size()) == 0 &&
                (kv.second->seen == scan_mark || (listed_by_walk && listed_by_walk(stem + kv.first))))
                continue;
            const std::string child = stem + kv.first;
            struct stat st;
            if (lstat(child_src.c_str(), &st) == 0) continue;
            if (errno != ENOENT && errno != ENOTDIR) {
                fail(MI_ERR_IO, "check on disk " + child + ": lstat " + child_src + ": " + strerror(errno));
                return;
            }
            gone.push_back(child);
        }
        for (const std::string& child : gone) {
            if (!add_whiteout(child)) return;
            add_ancestors(child, false, 0, 0);
            if (rc) return;
        }
    }
    // isUpdated (:487-503) on a walk entry as it comes, before any node is built for it: true = the tree holds this path
    // with a header tario.IsSimilarHeader calls similar (and, when both sides carry one, the same content root) --
    // maybeAddToLayer then adds nothing.  The scan's common case: most of a tree does not change between two steps.
    // lazy_file >= 0: the file's root is row lazy_file of a scan that may still be running (content_root is NULL then)
    // stamp (optional): the file's inode as the walk saw it -- recorded with the root whenever THIS commit hashed the content
    // own_root: the walk did not read the file because its content is KNOWN to be what the tree holds (MI_MEMFS_TRUST_CTIME):
    // its root is the node's own
    bool holds_similar(const std::string& dst, const mi_tree_entry& e, const uint8_t* content_root, int64_t lazy_file = -1,
                       const mi_walk::InodeStamp* stamp = nullptr, bool own_root = false) {
        mi_memtree::Node* cur = t.find(dst);
        if (!cur || cur->ref < 0 || e.kind > 3) return false;
        Node& o = nodes[cur->ref];
        if (o.e.kind > 3) return false;
        // (own_root: the content IS what the node's root describes -- nothing to compare, and the node's root stays untouched
        //  in whatever cache line it lies)
        const bool same_content = own_root && o.has_root && !o.root_pending;
        mi_tree_entry a, b = e;
        memset(&a, 0, sizeof a);
        a.relpath = o.e.relpath.empty() ? "" : o.e.relpath.c_str();
        a.link_target = o.e.has_link ? o.e.link.c_str() : nullptr;
        a.size = o.e.size; a.mtime_sec = o.e.mtime; a.mode = o.e.mode; a.kind = o.e.kind;
        a.uid = o.e.uid; a.gid = o.e.gid; a.file_index = -1;
        b.relpath = dst.c_str() + 1;                                             // dst without its leading "/"
        b.file_index = -1;
        const bool file_has_root = content_root != nullptr || lazy_file >= 0;
        const uint8_t *ra = nullptr, *rb = nullptr;
        if (!same_content && o.has_root && file_has_root && a.kind == 1 && b.kind == 1) {   // both carry a root: it decides -- now
            ra = root_now(o);
            rb = content_root;
            if (!rb && !rc) {
                const uint8_t* r = job ? job->wait_roots() : nullptr;
                if (!r) fail(MI_ERR_IO, "gpu scan: " + (job ? job->err : std::string("no scan under way")));
                else rb = r + 32 * (uint64_t)lazy_file;
            }
            if (rc) return false;
        }
        int similar = 0;
        if (mi_entry_similar(&a, &b, 0, ra, rb, &similar) != MI_OK) return false;
        if (similar && file_has_root && !o.has_root && o.e.kind == 1 && b.kind == 1) {   // (as in maybe_add)
            o.has_root = true;
            if (content_root) memcpy(o.root, content_root, 32);
            else { o.root_pending = true; o.batch_file = lazy_file; o.batch_gen = commit_gen; pending_refs.push_back(cur->ref); }
            ++n_roots_learned;
        }                                                                                // (a content-only change is counted
        if (similar && stamp && file_has_root && o.has_root && b.kind == 1)              //  where it is added: maybe_add)
            record_hashed(cur->ref, *stamp);                                             // hashed now, in this state
        return similar != 0;
    }
};

// evalSymlinks (utils.go:249-327): resolves the symlinks of p inside root; a link that leaves the
// root is an error.  Returns the path relative to root ("/"-rooted).
static bool eval_symlinks(const std::string& p, const std::string& root, std::string* out, std::string* err) {
    if (p.empty()) { *out = p; return true; }
    std::string cur = p;
    for (int walked = 0; walked <= 255;) {
        // resolve the first symlink found walking the components of cur
        const std::vector<std::string> parts = mi_memtree::Tree::parts(cur);
        std::string acc;
        bool replaced = false;
        for (size_t i = 0; i < parts.size(); ++i) {
            const std::string here = acc + "/" + parts[i];
            struct stat st;
            if (lstat((root + here).c_str(), &st) != 0) { *err = "walk link: lstat: " + here + ": " + strerror(errno); return false; }
            if (S_ISLNK(st.st_mode)) {
                std::vector<char> buf(4096);
                const ssize_t n = readlink((root + here).c_str(), buf.data(), buf.size() - 1);
                if (n < 0) { *err = "readlink " + here + ": " + strerror(errno); return false; }
                std::string target(buf.data(), (size_t)n);
                if (!target.empty() && target[0] == '/') {
                    if (!mi_walk::has_prefix(target, root)) {
                        *err = "link points outside of root: " + root + here + " -> " + target;
                        return false;
                    }
                    target = target.substr(root.size());
                    if (target.empty() || target[0] != '/') target = "/" + target;
                } else {
                    target = acc + "/" + target;                         // relative to the link's directory
                }
                for (size_t k = i + 1; k < parts.size(); ++k) target += "/" + parts[k];
                cur = mi_walk::clean_rooted(target);
                ++walked;
                replaced = true;
                break;
            }
            acc = here;
        }
        if (!replaced) { *out = mi_walk::abs_path(cur); return true; }
    }
    *err = "eval symlinks: too many links";
    return false;
}

}  // namespace mi_copy

struct mi_copy_layer {
    std::vector<mi_copy::Node> nodes;                       // commit order
};

// isDirFormat / checkCopyParams / resolveDestination (lib/snapshot/copy_op.go:149-180)
static bool copy_dst_is_dir_format(const std::string& dst) {
    return (!dst.empty() && dst.back() == '/') || dst == "." || dst == "..";
}
static std::string copy_check_params(uint64_t n_srcs, const char* work_dir, const std::string& dst) {
    if (n_srcs == 0) return "srcs cannot be empty";
    if (n_srcs > 1 && !copy_dst_is_dir_format(dst)) return "tarring multiple sources, destination must end with \"/\"";
    if ((dst.empty() || dst[0] != '/') && !(work_dir && work_dir[0] == '/'))
        return "dst is not absolute path, must specify absolute working directory";
    return "";
}

// ---- the caller's side of a COPY/ADD step: --chown and the source patterns ------------------------------------------
//
// utils.ResolveChown (lib/utils/utils.go:186-228): "<user>[:<group>]", each a number (strconv.Atoi: an optional sign and
// decimal digits) or a name looked up in the user / group database; no group = the uid; more than one ':' is an error.
static bool go_atoi(const std::string& t, long long* v) {
    size_t i = 0;
    if (!t.empty() && (t[0] == '+' || t[0] == '-')) i = 1;
    if (i == t.size()) return false;
    long long x = 0;
    for (size_t k = i; k < t.size(); ++k) {
        if (t[k] < '0' || t[k] > '9') return false;
        const int d = t[k] - '0';
        if (x > (9223372036854775807LL - d) / 10) return false;                  // beyond int64: Atoi reports a range error
        x = x * 10 + d;
    }
    *v = t[0] == '-' ? -x : x;
    return true;
}
extern "C" int mi_resolve_chown(const char* chown, int preserve_owner, int64_t* uid, int64_t* gid, char* err,
                                uint64_t err_cap) {
    auto put_err = [&](const std::string& m) { if (err && err_cap) snprintf(err, (size_t)err_cap, "%s", m.c_str()); };
    if (!uid || !gid) return MI_ERR_INVALID;
    *uid = *gid = 0;
    const std::string c = chown ? chown : "";
    if (!c.empty() && preserve_owner) { put_err("both chown and archive are true"); return MI_ERR_INVALID; }   // copy_op.go:52-55
    if (c.empty()) return MI_OK;
    std::vector<std::string> split(1);
    for (char ch : c) { if (ch == ':') split.emplace_back(); else split.back() += ch; }
    if (split.size() > 2) { put_err("resolve chown str: failed to split on ':'"); return MI_ERR_INVALID; }
    long long u = 0, g = 0;
    if (!go_atoi(split[0], &u)) {
        struct passwd pw, *res = nullptr;
        std::vector<char> buf(1 << 16);
        if (split[0].empty() || getpwnam_r(split[0].c_str(), &pw, buf.data(), buf.size(), &res) != 0 || !res) {
            put_err("resolve chown str: failed to look up user '" + split[0] + "'");
            return MI_ERR_INVALID;
        }
        u = (long long)pw.pw_uid;
    }
    if (split.size() == 1) { *uid = *gid = u; return MI_OK; }
    if (!go_atoi(split[1], &g)) {
        struct group gr, *res = nullptr;
        std::vector<char> buf(1 << 16);
        if (split[1].empty() || getgrnam_r(split[1].c_str(), &gr, buf.data(), buf.size(), &res) != 0 || !res) {
            put_err("resolve chown str: failed to look up group '" + split[0] + "'");   // (the reference names the USER here, :221)
            return MI_ERR_INVALID;
        }
        g = (long long)gr.gr_gid;
    }
    *uid = u; *gid = g;
    return MI_OK;
}

// path/filepath.Match and Glob as the Go 1.14 toolchain the reference builds with defines them (Makefile:34) --
// resolveFromPaths (lib/builder/step/add_copy_step.go:171-185) runs every source of a COPY/ADD through Glob:
//   '*' any run of non-'/' characters, '?' one non-'/' character, '[' ['^'] ranges ']' a character class (not empty;
//   lo '-' hi; characters are runes), '\\' escapes the next character; the whole name has to match.  A malformed
//   pattern is ErrBadPattern -- but only where matching GETS to the bad part (1.14 stops at the end of the name).
namespace mi_glob {

static size_t rune_at(const std::string& s, size_t i, uint32_t* r) {           // utf8.DecodeRuneInString
    const unsigned char c = (unsigned char)s[i];
    auto cont = [&](size_t k) { return i + k < s.size() && ((unsigned char)s[i + k] & 0xC0) == 0x80; };
    if (c < 0x80) { *r = c; return 1; }
    if (c >= 0xC2 && c <= 0xDF && cont(1)) { *r = ((c & 0x1Fu) << 6) | ((unsigned char)s[i + 1] & 0x3Fu); return 2; }
    if (c >= 0xE0 && c <= 0xEF && cont(1) && cont(2)) {
        const uint32_t v = ((c & 0x0Fu) << 12) | (((unsigned char)s[i + 1] & 0x3Fu) << 6) | ((unsigned char)s[i + 2] & 0x3Fu);
        if (v >= 0x800 && !(v >= 0xD800 && v <= 0xDFFF)) { *r = v; return 3; }
    }
    if (c >= 0xF0 && c <= 0xF4 && cont(1) && cont(2) && cont(3)) {
        const uint32_t v = ((c & 0x07u) << 18) | (((unsigned char)s[i + 1] & 0x3Fu) << 12) |
                           (((unsigned char)s[i + 2] & 0x3Fu) << 6) | ((unsigned char)s[i + 3] & 0x3Fu);
        if (v >= 0x10000 && v <= 0x10FFFF) { *r = v; return 4; }
    }
    *r = 0xFFFD;                                                                // RuneError, width 1
    return 1;
}

// getEsc: one possibly escaped character of a class; false = ErrBadPattern
static bool get_esc(const std::string& chunk, size_t* at, uint32_t* r) {
    size_t i = *at;
    if (i >= chunk.size() || chunk[i] == '-' || chunk[i] == ']') return false;
    if (chunk[i] == '\\') { if (++i >= chunk.size()) return false; }
    const size_t n = rune_at(chunk, i, r);
    bool ok = !(*r == 0xFFFD && n == 1);
    i += n;
    if (i >= chunk.size()) ok = false;
    *at = i;
    return ok;
}

// matchChunk: does chunk (no '*') match a prefix of s[from:]?  rest = where the match ends
static bool match_chunk(const std::string& chunk, const std::string& s, size_t from, size_t* rest, bool* bad) {
    size_t c = 0, i = from;
    while (c < chunk.size()) {
        if (i >= s.size()) return false;
        switch (chunk[c]) {
            case '[': {
                uint32_t r;
                i += rune_at(s, i, &r);
                if (++c >= chunk.size()) { *bad = true; return false; }
                const bool negated = chunk[c] == '^';
                if (negated) ++c;
                bool match = false;
                for (int nrange = 0;; ++nrange) {
                    if (c < chunk.size() && chunk[c] == ']' && nrange > 0) { ++c; break; }
                    uint32_t lo, hi;
                    if (!get_esc(chunk, &c, &lo)) { *bad = true; return false; }
                    hi = lo;
                    if (chunk[c] == '-') {
                        ++c;
                        if (!get_esc(chunk, &c, &hi)) { *bad = true; return false; }
                    }
                    if (lo <= r && r <= hi) match = true;
                }
                if (match == negated) return false;
                break;
            }
            case '?': {
                if (s[i] == '/') return false;
                uint32_t r;
                i += rune_at(s, i, &r);
                ++c;
                break;
            }
            case '\\':
                if (++c >= chunk.size()) { *bad = true; return false; }
                /* fallthrough */
            default:
                if (chunk[c] != s[i]) return false;
                ++i; ++c;
        }
    }
    *rest = i;
    return true;
}

static bool match(const std::string& pattern, const std::string& name, bool* bad) {
    size_t p = 0, n = 0;
    *bad = false;
    while (p < pattern.size()) {
        bool star = false;                                                      // scanChunk
        while (p < pattern.size() && pattern[p] == '*') { ++p; star = true; }
        bool inrange = false;
        size_t e = p;
        for (; e < pattern.size(); ++e) {
            const char ch = pattern[e];
            if (ch == '\\') { if (e + 1 < pattern.size()) ++e; }
            else if (ch == '[') inrange = true;
            else if (ch == ']') inrange = false;
            else if (ch == '*' && !inrange) break;
        }
        const std::string chunk = pattern.substr(p, e - p);
        p = e;
        if (star && chunk.empty()) return name.find('/', n) == std::string::npos;   // a trailing * takes the rest
        size_t t = 0;
        const bool ok = match_chunk(chunk, name, n, &t, bad);
        if (ok && (t == name.size() || p < pattern.size())) { n = t; continue; }
        if (*bad) return false;
        if (star) {
            bool advanced = false;
            for (size_t i = n; i < name.size() && name[i] != '/'; ++i) {
                if (match_chunk(chunk, name, i + 1, &t, bad)) {
                    if (p >= pattern.size() && t < name.size()) continue;       // last chunk: the name has to end here
                    n = t;
                    advanced = true;
                    break;
                }
                if (*bad) return false;
            }
            if (advanced) continue;
        }
        return false;
    }
    return n == name.size();
}

static bool has_meta(const std::string& s) { return s.find_first_of("*?[\\") != std::string::npos; }

// glob(dir, pattern, matches): the names of dir that match, sorted, joined to dir; I/O errors are ignored
static bool glob_dir(const std::string& dir, const std::string& pattern, std::vector<std::string>* out) {
    struct stat st;
    if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return true;
    DIR* d = opendir(dir.c_str());
    if (!d) return true;
    std::vector<std::string> names;
    while (struct dirent* de = readdir(d)) {
        const std::string n = de->d_name;
        if (n != "." && n != "..") names.push_back(n);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    for (const std::string& n : names) {
        bool bad = false;
        if (match(pattern, n, &bad)) {
            std::string j = dir == "." ? n : (dir.back() == '/' ? dir + n : dir + "/" + n);   // filepath.Join(dir, n)
            out->push_back(dir == "." ? j : (dir[0] == '/' ? mi_walk::clean_rooted(j) : mi_walk::clean_any(j)));
        }
        if (bad) return false;
    }
    return true;
}

static bool glob(const std::string& pattern, std::vector<std::string>* out) {   // false = ErrBadPattern
    bool bad = false;
    match(pattern, "", &bad);
    if (bad) return false;
    if (!has_meta(pattern)) {
        struct stat st;
        if (lstat(pattern.c_str(), &st) == 0) out->push_back(pattern);
        return true;
    }
    const size_t cut = pattern.find_last_of('/');                               // filepath.Split
    std::string dir = cut == std::string::npos ? "" : pattern.substr(0, cut + 1);
    const std::string file = cut == std::string::npos ? pattern : pattern.substr(cut + 1);
    if (dir.empty()) dir = ".";                                                 // cleanGlobPath
    else if (dir != "/") dir.pop_back();
    if (!has_meta(dir)) return glob_dir(dir, file, out);
    if (dir == pattern) return false;                                           // "Prevent infinite recursion"
    std::vector<std::string> dirs;
    if (!glob(dir, &dirs)) return false;
    for (const std::string& d : dirs)
        if (!glob_dir(d, file, out)) return false;
    return true;
}

}  // namespace mi_glob

extern "C" int mi_path_match(const char* pattern, const char* name, int* matched) {
    if (!pattern || !name || !matched) return MI_ERR_INVALID;
    bool bad = false;
    *matched = mi_glob::match(pattern, name, &bad) ? 1 : 0;
    return bad ? MI_ERR_INVALID : MI_OK;                                        // ErrBadPattern
}

// resolveFromPaths: every source joined to the context root and globbed; no match (or a bad pattern) = the joined
// path itself.  out = the resolved paths, NUL-terminated, back to back.
extern "C" int mi_context_sources(const char* context_root, const char* const* from_paths, uint64_t n_paths,
                                  char* out, uint64_t cap, uint64_t* n_out, uint64_t* bytes_out) {
    if (!context_root || (n_paths && !from_paths) || !n_out || !bytes_out || (cap && !out)) return MI_ERR_INVALID;
    std::string all;
    uint64_t n = 0;
    for (uint64_t i = 0; i < n_paths; ++i) {
        const std::string joined0 = std::string(context_root) + "/" + (from_paths[i] ? from_paths[i] : "");
        const std::string source = joined0[0] == '/' ? mi_walk::clean_rooted(joined0) : mi_walk::clean_any(joined0);
        std::vector<std::string> m;
        if (!mi_glob::glob(source, &m) || m.empty()) m.assign(1, source);
        for (const std::string& x : m) { all += x; all.push_back('\0'); ++n; }
    }
    *n_out = n;
    *bytes_out = all.size();
    if (cap < all.size()) return MI_ERR_CAPACITY;
    if (!all.empty()) memcpy(out, all.data(), all.size());
    return MI_OK;
}

extern "C" int mi_copy_op_resolve(uint64_t n_srcs, const char* work_dir, const char* dst, char* dst_out,
                                  uint64_t cap, char* err, uint64_t err_cap) {
    if (!dst || !dst_out) return MI_ERR_INVALID;
    const std::string d = dst;
    const std::string bad = copy_check_params(n_srcs, work_dir, d);
    if (!bad.empty()) {
        if (err && err_cap) snprintf(err, (size_t)err_cap, "check copy param: %s", bad.c_str());
        return MI_ERR_INVALID;
    }
    std::string r = d;
    if (d[0] != '/') {                                      // filepath.Join cleans; the trailing "/" is put back (the reference
                                                            // appends it even to a joined "/", giving "//": the same path once cleaned)
        r = mi_walk::abs_path(std::string(work_dir) + "/" + d);
        if (copy_dst_is_dir_format(d) && r.back() != '/') r += "/";
    }
    if (cap < r.size() + 1) return MI_ERR_CAPACITY;
    memcpy(dst_out, r.c_str(), r.size() + 1);
    return MI_OK;
}

// addToLayer (mem_fs.go:343-421) for each op, against fs.t, into fs.layer; fs.rc / fs.err carry what maybeAddToLayer
// refuses, *err_out everything else.
// In two steps.  PLAN: what the ops read from the DISK -- the parameter check, the stat of a single source, evalSymlinks,
// the walk of every source -- for all ops, in order, stopping at the first failure.  None of it depends on the tree, so
// it can run ahead; with a batch attached the walks stage every regular file on the GPU while they list it (an entry's
// file_index = its row).  APPLY: the ops against the tree, in order, each failure raised where the interleaved loop of the
// reference raises it (an op-2 source that does not exist fails after op 1 has been applied, not before).  Between the
// two, a content-aware commit runs the batch: the apply step then sees a chunk root for every regular file.
struct CopySrcPlan { std::string src; mi_walk::Tree walked; };
struct CopyOpPlan {
    std::string src_root, dst;
    bool create_dst = true;
    std::vector<CopySrcPlan> srcs;
};
struct CopyPlan {
    std::vector<CopyOpPlan> ops;
    int err_rc = MI_OK;                 // the failure planning stopped at ...
    std::string err;
    bool err_before_dst = false;        // ... raised before the last planned op touches the tree / after its planned sources
    uint64_t n_walked = 0;
};
static void copy_ops_plan(const mi_copy::Fs& fs, const mi_copy_op* ops, uint64_t n_ops, mi_batch* batch, CopyPlan* plan) {
    auto stop = [&](int rc, const std::string& m, bool before_dst) { plan->err_rc = rc; plan->err = m; plan->err_before_dst = before_dst; };
    for (uint64_t k = 0; k < n_ops; ++k) {
        const mi_copy_op& c = ops[k];
        plan->ops.emplace_back();
        CopyOpPlan& op = plan->ops.back();
        if (!c.src_root || !c.dst || (c.n_srcs && !c.srcs)) return stop(MI_ERR_INVALID, "", true);
        {   // what NewCopyOperation refuses (copy_op.go:48-50): the dst here is the resolved one, so it is absolute
            const std::string bad = copy_check_params(c.n_srcs, nullptr, c.dst);
            if (!bad.empty()) return stop(MI_ERR_INVALID, "check copy param: " + bad, true);
        }
        op.src_root = mi_walk::abs_path(c.src_root);
        op.dst = c.dst;
        if (c.n_srcs == 1) {
            struct stat st;
            const std::string s0 = op.src_root + mi_walk::abs_path(c.srcs[0] ? c.srcs[0] : "");
            if (stat(s0.c_str(), &st) != 0) return stop(MI_ERR_IO, "stat src " + s0 + ": " + strerror(errno), true);
            if (!S_ISDIR(st.st_mode)) op.create_dst = false;          // case 1: file onto file
        }
        for (uint64_t si = 0; si < c.n_srcs; ++si) {
            std::string rel, e2;
            if (!mi_copy::eval_symlinks(mi_walk::abs_path(c.srcs[si] ? c.srcs[si] : ""), op.src_root, &rel, &e2))
                return stop(MI_ERR_IO, "eval symlinks for " + std::string(c.srcs[si] ? c.srcs[si] : "") + ": " + e2, false);
            CopySrcPlan sp;
            sp.src = op.src_root == "/" ? rel : op.src_root + (rel == "/" ? "" : rel);
            std::string werr;                                           // shouldSkip with a nil blacklist; createHeader
            const int wrc = batch ? mi_walk::scan_walk_collect_batch(sp.src, fs.root, &sp.walked, &werr, batch)   // trims link targets
                                  : mi_walk::scan_walk_collect(sp.src, fs.root, &sp.walked, &werr);               // by the MEMFS root
            if (wrc) return stop(wrc, "copy src " + sp.src + ": " + werr, false);
            plan->n_walked += sp.walked.entries.size();
            op.srcs.push_back(std::move(sp));
        }
    }
}
// roots: 32 bytes per batch row (NULL: the reference's metadata-only isUpdated)
static int copy_ops_apply(mi_copy::Fs& fs, const mi_copy_op* ops, const CopyPlan& plan, const uint8_t* roots, std::string* err_out) {
    auto put_err = [&](const std::string& m) { *err_out = m; };
    for (size_t k = 0; k < plan.ops.size() && !fs.rc; ++k) {
        const CopyOpPlan& op = plan.ops[k];
        const mi_copy_op& c = ops[k];
        const bool last = k + 1 == plan.ops.size();
        if (last && plan.err_rc && plan.err_before_dst) { put_err(plan.err); return plan.err_rc; }
        std::string dst = op.dst;
        if (op.create_dst) {
            std::string resolved = fs.add_ancestors(mi_walk::abs_path(dst), true, c.uid, c.gid);
            if (fs.rc) break;
            if (resolved.empty() || resolved.back() != '/') resolved += "/";
            dst = resolved;
        }
        const bool dst_is_dir = !dst.empty() && dst.back() == '/';
        for (size_t si = 0; si < op.srcs.size() && !fs.rc; ++si) {
            const std::string& src = op.srcs[si].src;
            for (const mi_walk::Entry& we : op.srcs[si].walked.entries) {
                const bool is_src = we.relpath == ".";
                std::string curr_dst;
                if (is_src) {
                    if (we.kind == 0) continue;                         // the directory itself: contents only
                    curr_dst = !dst_is_dir ? dst : mi_walk::clean_rooted(dst + "/" + mi_walk::base_of(src));
                } else {
                    curr_dst = mi_walk::clean_rooted(dst + "/" + we.relpath);
                }
                curr_dst = mi_walk::abs_path(curr_dst);
                mi_copy::Node n;
                n.e = we;
                n.e.relpath = curr_dst == "/" ? "" : curr_dst.substr(1);
                n.e.uid = c.uid;
                n.e.gid = c.gid;
                n.e.file_index = -1;
                if (roots && we.kind == 1 && we.file_index >= 0) {
                    n.batch_file = we.file_index;
                    n.batch_gen = fs.commit_gen;
                    n.has_root = true;
                    memcpy(n.root, roots + (uint64_t)we.file_index * 32, 32);
                } else if (fs.job && we.kind == 1 && we.file_index >= 0) {     // (a pipelined commit: the scan may still be running)
                    n.batch_file = we.file_index;
                    n.batch_gen = fs.commit_gen;
                    n.has_root = true;
                    n.root_pending = true;
                }
                const std::string curr_src = is_src ? src : src + "/" + we.relpath;
                fs.maybe_add(curr_src, curr_dst, n);
                if (fs.rc) break;
            }
        }
        if (!fs.rc && last && plan.err_rc) { put_err(plan.err); return plan.err_rc; }
    }
    if (fs.rc) { put_err(fs.err); return fs.rc; }
    return MI_OK;
}
static int copy_ops_into(mi_copy::Fs& fs, const mi_copy_op* ops, uint64_t n_ops, std::string* err_out) {
    CopyPlan plan;
    copy_ops_plan(fs, ops, n_ops, nullptr, &plan);
    return copy_ops_apply(fs, ops, plan, nullptr, err_out);
}

// ---- CopyOperation.Execute: the on-disk copy of a COPY/ADD step with --modifyfs (lib/snapshot/copy_op.go:83-147) over
// fileio.Copier (lib/fileio/copy.go:31-394).  Owners: --chown -> the op's uid/gid for the destination directory if it
// has to be created and, always, for everything copied; from the context without --chown -> the same with 0:0; --from
// --archive -> a created destination directory gets the source's owner, everything copied keeps its own; --from alone ->
// owners as they are (a created destination directory: root).  Permission bits travel with the files; mtimes do not.
namespace mi_copyexec {

struct Owner { bool set = false; uint32_t uid = 0, gid = 0; bool overwrite = false; };
struct Copier {
    std::vector<std::string> blacklist;
    Owner dst_dir, children;
    std::string err;

    bool fail(const std::string& m) { err = m; return false; }
    bool blacklisted(const std::string& p) const { return mi_walk::is_descendant_of_any(p, blacklist); }

    bool mkdir_all(const std::string& dst) {                                     // Copier.mkdirAll (:336-393)
        if (dst.empty()) return fail("empty dst directory");
        const std::string abs = mi_walk::abs_path(dst);                          // callers pass absolute paths
        std::string cur;
        const std::vector<std::string> ps = mi_memtree::Tree::parts(abs);
        for (size_t k = 0; k + 1 < ps.size(); ++k) {
            cur += "/" + ps[k];
            struct stat st;
            if (lstat(cur.c_str(), &st) == 0) continue;
            if (errno != ENOENT) return fail("stat " + cur + ": " + strerror(errno));
            if (mkdir(cur.c_str(), 0755) != 0) return fail("mkdir " + cur + " with default mode 0755: " + strerror(errno));
            if (chown(cur.c_str(), 0, 0) != 0) return fail("chown " + cur + " with default owner (0:0): " + strerror(errno));
        }
        struct stat st;
        if (lstat(abs.c_str(), &st) != 0) {
            if (errno != ENOENT) return fail("stat " + abs + ": " + strerror(errno));
            if (mkdir(abs.c_str(), 0755) != 0) return fail("mkdir " + abs + " with default mode 0755: " + strerror(errno));
            const uint32_t u = dst_dir.set ? dst_dir.uid : 0, g = dst_dir.set ? dst_dir.gid : 0;
            if (chown(abs.c_str(), u, g) != 0) return fail("chown " + abs + ": " + strerror(errno));
        } else if (dst_dir.set && dst_dir.overwrite) {
            if (chown(abs.c_str(), dst_dir.uid, dst_dir.gid) != 0) return fail("chown " + abs + ": " + strerror(errno));
        }
        return true;
    }
    bool copy_symlink(const std::string& src, const std::string& dst) {          // :232-247
        struct stat st;
        if (lstat(dst.c_str(), &st) == 0 && remove(dst.c_str()) != 0)
            return fail("remove existing file " + dst + ": " + strerror(errno));
        std::vector<char> buf(4096);
        const ssize_t n = readlink(src.c_str(), buf.data(), buf.size() - 1);
        if (n < 0) return fail("read link " + src + ": " + strerror(errno));
        const std::string target(buf.data(), (size_t)n);
        if (symlink(target.c_str(), dst.c_str()) != 0)
            return fail("write link " + dst + " with content " + target + ": " + strerror(errno));
        return true;
    }
    bool copy_file(const std::string& src, const std::string& dst) {             // copyFile + copyRegularFile (:160-230)
        struct stat fi;
        if (lstat(src.c_str(), &fi) != 0) return fail("lstat " + src + ": " + strerror(errno));
        // (a blacklisted SOURCE FILE is only logged here -- the reference's else-if chain goes on to copy it; blacklisted
        // entries below a copied directory never get this far.  The same chain would also skip the special-file test for
        // it and open a blacklisted FIFO for reading; that one corner is not followed: a special file is never opened)
        if (!S_ISREG(fi.st_mode) && !S_ISDIR(fi.st_mode) && !S_ISLNK(fi.st_mode)) return true;
        if (S_ISLNK(fi.st_mode)) return copy_symlink(src, dst);                  // never chown'ed: that would hit the target
        struct stat dt;
        if (lstat(dst.c_str(), &dt) == 0) {
            if (chmod(dst.c_str(), 0777) != 0) return fail("chmod " + dst + ": " + strerror(errno));
        } else if (errno != ENOENT) {
            return fail("lstat " + dst + ": " + strerror(errno));
        }
        const int r = open(src.c_str(), O_RDONLY | O_CLOEXEC);
        if (r < 0) return fail("open " + dst + ": " + strerror(errno));
        const int w = open(dst.c_str(), O_WRONLY | O_CREAT | O_CLOEXEC, 0777);
        if (w < 0) { const int e = errno; close(r); return fail("create " + dst + ": " + strerror(e)); }
        bool ok = ftruncate(w, 0) == 0;
        std::string e = ok ? "" : std::string("truncate ") + dst + ": " + strerror(errno);
        std::vector<char> buf(1 << 20);
        while (ok) {
            const ssize_t n = read(r, buf.data(), buf.size());
            if (n < 0 && errno == EINTR) continue;
            if (n < 0) { ok = false; e = "copy " + src + " to " + dst + ": " + strerror(errno); break; }
            if (n == 0) break;
            for (ssize_t done = 0; done < n;) {
                const ssize_t k = write(w, buf.data() + done, (size_t)(n - done));
                if (k < 0 && errno == EINTR) continue;
                if (k < 0) { ok = false; e = "copy " + src + " to " + dst + ": " + strerror(errno); break; }
                done += k;
            }
        }
        close(r);
        close(w);
        if (!ok) return fail(e);
        const uint32_t u = children.set && children.overwrite ? children.uid : fi.st_uid;
        const uint32_t g = children.set && children.overwrite ? children.gid : fi.st_gid;
        if (chown(dst.c_str(), u, g) != 0) return fail("chown " + dst + ": " + strerror(errno));
        if (chmod(dst.c_str(), fi.st_mode & 07777) != 0) return fail("chmod " + dst + ": " + strerror(errno));   // after chown
        return true;
    }
    bool copy_dir(const std::string& src, const std::string& dst) {              // copyDir (:289-330): one directory, no contents
        struct stat si;
        if (lstat(src.c_str(), &si) != 0) return fail("lstat " + src + ": " + strerror(errno));
        if (!S_ISDIR(si.st_mode)) return fail("source " + src + " is not a directory");
        if (blacklisted(src)) return true;
        struct stat di;
        if (lstat(dst.c_str(), &di) != 0) {
            if (errno != ENOENT) return fail("lstat " + dst + ": " + strerror(errno));
            if (mkdir(dst.c_str(), si.st_mode & 07777) != 0) return fail("mkdir " + dst + ": " + strerror(errno));
        } else if (!S_ISDIR(di.st_mode)) {
            return fail("dst is not a directory");
        }
        if (chmod(dst.c_str(), si.st_mode & 07777) != 0) return fail("chmod " + dst + ": " + strerror(errno));
        const uint32_t u = children.set && children.overwrite ? children.uid : si.st_uid;
        const uint32_t g = children.set && children.overwrite ? children.gid : si.st_gid;
        if (chown(dst.c_str(), u, g) != 0) return fail("chown " + dst + ": " + strerror(errno));
        return true;
    }
    bool copy_dir_contents(const std::string& src, const std::string& dst, const std::string& orig_dst) {   // :252-285
        DIR* d = opendir(src.c_str());
        if (!d) return fail("read dir " + src + ": " + strerror(errno));
        std::vector<std::string> names;
        while (struct dirent* de = readdir(d)) {
            const std::string n = de->d_name;
            if (n != "." && n != "..") names.push_back(n);
        }
        closedir(d);
        std::sort(names.begin(), names.end());                                   // ioutil.ReadDir sorts by name
        for (const std::string& n : names) {
            const std::string cs = (src == "/" ? "" : src) + "/" + n, cd = (dst == "/" ? "" : dst) + "/" + n;
            if (blacklisted(cs) || cs == orig_dst) continue;                     // "Silently break infinite loop"
            struct stat st;
            if (lstat(cs.c_str(), &st) != 0) return fail("lstat " + cs + ": " + strerror(errno));
            if (S_ISDIR(st.st_mode)) {
                if (!copy_dir(cs, cd)) return fail("copy dir " + cs + " to " + cd + ": " + err);
                if (!copy_dir_contents(cs, cd, orig_dst)) return fail("copy dir contents " + cs + " to " + cd + ": " + err);
            } else if (!copy_file(cs, cd)) {
                return fail("copy file " + cs + " to " + cd + ": " + err);
            }
        }
        return true;
    }
    bool CopyFile(const std::string& src, const std::string& dst) {              // :122-130
        const std::string dir = mi_walk::dir_of(dst);
        if (!mkdir_all(dir)) return fail("mkdir all " + dir + ": " + err);
        return copy_file(src, dst);
    }
    bool CopyDir(const std::string& src, const std::string& dst) {               // :142-156
        if (blacklisted(src)) return true;
        if (!mkdir_all(dst)) return fail("mkdir all " + dst + ": " + err);
        return copy_dir_contents(src, dst, mi_walk::abs_path(dst));
    }
};

}  // namespace mi_copyexec

extern "C" int mi_copy_op_execute(const mi_copy_op* op, uint32_t flags, const char* const* blacklist, uint64_t n_blacklist,
                                  char* err, uint64_t err_cap) {
    auto put_err = [&](const std::string& m) { if (err && err_cap) snprintf(err, (size_t)err_cap, "%s", m.c_str()); };
    if (!op || !op->src_root || !op->dst || (op->n_srcs && !op->srcs) || (n_blacklist && !blacklist)) return MI_ERR_INVALID;
    const bool chown_given = flags & MI_COPY_CHOWN, internal = flags & MI_COPY_INTERNAL, archive = flags & MI_COPY_PRESERVE_OWNER;
    if (chown_given && archive) { put_err("both chown and archive are true"); return MI_ERR_INVALID; }
    const std::string src_root = mi_walk::abs_path(op->src_root);
    const std::string dst = op->dst;
    for (uint64_t si = 0; si < op->n_srcs; ++si) {
        std::string rel, e2;
        const std::string given = op->srcs[si] ? op->srcs[si] : "";
        if (!mi_copy::eval_symlinks(mi_walk::abs_path(given), src_root, &rel, &e2)) {
            put_err("eval symlinks for " + given + ": " + e2);
            return MI_ERR_IO;
        }
        const std::string src = src_root == "/" ? rel : src_root + (rel == "/" ? "" : rel);
        struct stat fi;
        if (lstat(src.c_str(), &fi) != 0) { put_err("lstat " + src + ": " + strerror(errno)); return MI_ERR_IO; }
        mi_copyexec::Copier c;
        if (!internal)                                                           // "there is no need to blacklist any path" for a
            for (uint64_t k = 0; k < n_blacklist; ++k) c.blacklist.push_back(blacklist[k] ? blacklist[k] : "");   // checkpointed stage
        if (chown_given) {
            c.dst_dir = {true, op->uid, op->gid, false};
            c.children = {true, op->uid, op->gid, true};
        } else if (!internal) {
            c.dst_dir = {true, 0, 0, false};
            c.children = {true, 0, 0, true};
        } else if (archive) {
            c.dst_dir = {true, fi.st_uid, fi.st_gid, false};
        }
        bool ok;
        std::string what;
        if (S_ISDIR(fi.st_mode)) {
            ok = c.CopyDir(src, dst);
            what = "copy dir " + src + " to dir " + dst;
        } else if (copy_dst_is_dir_format(dst)) {
            const std::string target = mi_walk::abs_path(dst + "/" + mi_walk::base_of(src));
            ok = c.CopyFile(src, target);
            what = "copy file " + src + " to dir " + target;
        } else {
            ok = c.CopyFile(src, dst);
            what = "copy file " + src + " to file " + dst;
        }
        if (!ok) { put_err(what + ": " + c.err); return MI_ERR_IO; }
    }
    return MI_OK;
}

// ---- untar: MemFS.untarOneItem and tario.ApplyHeader (lib/snapshot/mem_fs.go:571-718, lib/tario/apply.go:23-47) --------
namespace mi_untar {

static int rm_cb(const char* p, const struct stat*, int, struct FTW*) { return remove(p); }
static bool remove_all(const std::string& p, std::string* err) {                 // os.RemoveAll
    struct stat st;
    if (lstat(p.c_str(), &st) != 0) {
        if (errno == ENOENT || errno == ENOTDIR) return true;
        *err = "lstat " + p + ": " + strerror(errno);
        return false;
    }
    if (!S_ISDIR(st.st_mode)) {
        if (unlink(p.c_str()) == 0 || errno == ENOENT) return true;
        *err = "unlinkat " + p + ": " + strerror(errno);
        return false;
    }
    if (nftw(p.c_str(), rm_cb, 64, FTW_DEPTH | FTW_PHYS) != 0) { *err = "unlinkat " + p + ": " + strerror(errno); return false; }
    return true;
}

// tario.ApplyHeader: owner, then permission bits (chmod after chown: setuid / setgid survive), then mtime; never on a
// symlink and never FOR a symlink header
static bool apply_header(const std::string& path, const mi_tree_entry& h, std::string* err) {
    struct stat st;
    if (lstat(path.c_str(), &st) != 0) { *err = "lstat " + path + ": " + strerror(errno); return false; }
    if (S_ISLNK(st.st_mode) || h.kind == 2) { *err = "update symlink instead of file: " + path; return false; }
    if (chown(path.c_str(), h.uid, h.gid) != 0) { *err = "chown " + path + ": " + strerror(errno); return false; }
    if (chmod(path.c_str(), h.mode & 07777) != 0) { *err = "chmod " + path + ": " + strerror(errno); return false; }
    struct timespec ts[2];
    ts[0].tv_sec = ts[1].tv_sec = (time_t)h.mtime_sec;
    ts[0].tv_nsec = ts[1].tv_nsec = 0;
    if (utimensat(AT_FDCWD, path.c_str(), ts, 0) != 0) { *err = "chtimes " + path + ": " + strerror(errno); return false; }
    return true;
}

static bool copy_range(int in_fd, uint64_t off, uint64_t len, int out_fd, std::string* err) {
    std::vector<char> buf(1 << 20);
    while (len) {
        const size_t want = len < buf.size() ? (size_t)len : buf.size();
        const ssize_t r = pread(in_fd, buf.data(), want, (off_t)off);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) { *err = r == 0 ? "unexpected EOF" : strerror(errno); return false; }
        size_t done = 0;
        while (done < (size_t)r) {
            const ssize_t w = write(out_fd, buf.data() + done, (size_t)r - done);
            if (w < 0 && errno == EINTR) continue;
            if (w < 0) { *err = strerror(errno); return false; }
            done += (size_t)w;
        }
        off += (uint64_t)r;
        len -= (uint64_t)r;
    }
    return true;
}

// untarOneItem(path, header, r): root = fs.tree.src; path = filepath.Join(root, hdr.Name); content of a regular file =
// [data_off, +size) of tar_fd
static bool one_item(const std::string& root, const std::string& path, const mi_tree_entry& h, int tar_fd, uint64_t data_off,
                     std::string* err) {
    const std::string base = mi_walk::base_of(path), dir = mi_walk::dir_of(path);
    std::string e;
    if (mi_walk::has_prefix(base, ".wh.")) {                                     // untarWhiteout
        if (!remove_all((dir == "/" ? "" : dir) + "/" + base.substr(4), &e)) { *err = "untar dir: untar whiteout: " + e; return false; }
        return true;
    }
    struct stat st;
    if (lstat(path.c_str(), &st) != 0) {
        if (errno != ENOENT && errno != ENOTDIR) { *err = "lstat " + path + ": " + strerror(errno); return false; }
    } else {
        // the header of what is there (tar.FileInfoHeader), link target trimmed of the root
        mi_tree_entry local;
        memset(&local, 0, sizeof local);
        std::string link;
        local.relpath = h.relpath && *h.relpath ? h.relpath : "x";              // (only "both names empty" matters to the predicate)
        local.mode = st.st_mode; local.mtime_sec = st.st_mtime; local.uid = st.st_uid; local.gid = st.st_gid;
        local.kind = S_ISDIR(st.st_mode) ? 0 : S_ISREG(st.st_mode) ? 1 : S_ISLNK(st.st_mode) ? 2 : 4;
        local.size = S_ISREG(st.st_mode) ? (uint64_t)st.st_size : 0;
        if (S_ISLNK(st.st_mode)) {
            std::vector<char> buf(4096);
            const ssize_t n = readlink(path.c_str(), buf.data(), buf.size() - 1);
            if (n < 0) { *err = "read link " + path + ": " + strerror(errno); return false; }
            link.assign(buf.data(), (size_t)n);
            if (!link.empty() && link[0] == '/') {
                if (!mi_walk::has_prefix(link, root)) { *err = "trim link " + link + ": failed to trim root prefix " + root + " from path " + link; return false; }
                link = mi_walk::abs_path(link.substr(root.size()));
            }
            local.link_target = link.c_str();
        }
        if (local.kind > 3) { *err = "compare headers " + path + ": unsupported type"; return false; }
        int similar = 0;
        mi_tree_entry hh = h;
        if (!hh.relpath || !*hh.relpath) hh.relpath = "x";
        if (mi_entry_similar(&local, &hh, 0, nullptr, nullptr, &similar) != MI_OK) { *err = "compare headers " + path + ": unsupported type"; return false; }
        if (similar) return true;                                                // "already on disk, nothing needs to be done"
        if (h.kind == 0 && S_ISDIR(st.st_mode)) {                                // existing directories are updated, not deleted
            if (!apply_header(path, h, &e)) { *err = "update fi " + path + ": " + e; return false; }
            return true;
        }
        if (!remove_all(path, &e)) { *err = "clear existing file " + path + ": " + e; return false; }
    }
    if (h.kind == 0) {
        if (mkdir(path.c_str(), h.mode & 07777) != 0) { *err = "untar dir: create dir " + path + ": " + strerror(errno); return false; }
        if (!apply_header(path, h, &e)) { *err = "untar dir: update fi " + path + ": " + e; return false; }
    } else if (h.kind == 2) {
        std::string target = h.link_target ? h.link_target : "";
        if (!target.empty() && target[0] == '/') target = mi_walk::abs_path(root + "/" + target);   // filepath.Join(root, target)
        if (symlink(target.c_str(), path.c_str()) != 0) { *err = "untar symlink: create symlink " + path + " => " + target + ": " + strerror(errno); return false; }
        if (lchown(path.c_str(), h.uid, h.gid) != 0) { *err = "untar symlink: lchown symlink: " + path; return false; }
    } else if (h.kind == 3) {
        const std::string target = mi_walk::abs_path(root + "/" + (h.link_target ? h.link_target : ""));
        if (link(target.c_str(), path.c_str()) != 0) { *err = "untar hard link: create link " + path + " => " + target + ": " + strerror(errno); return false; }
        if (!apply_header(path, h, &e)) { *err = "untar hard link: update hard link " + path + ": " + e; return false; }
    } else {
        const int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY | O_CLOEXEC, h.mode & 07777);
        if (fd < 0) { *err = "untar file: open file " + path + ": " + strerror(errno); return false; }
        const bool ok = h.size == 0 || (tar_fd >= 0 && copy_range(tar_fd, data_off, h.size, fd, &e));
        close(fd);
        if (!ok) { *err = "untar file: read from file " + path + ": " + (tar_fd < 0 ? "no archive to read from" : e); return false; }
        if (!apply_header(path, h, &e)) { *err = "untar file: update fi " + path + ": " + e; return false; }
    }
    return true;
}

}  // namespace mi_untar

// ---- MemFS as a handle: the reference's type (lib/snapshot/mem_fs.go:59-125) behind the ABI ---------------------------
// One tree for the life of a build, as in the reference: base layers are merged into it (UpdateFromTarReader), every
// step's layer is computed against it and folds into it (AddLayerByScan / AddLayerByCopyOps).  What the stateless calls
// above cannot keep between calls is kept here: the directories addAncestors created, and for every node the path its
// content came from (memFSNode.src -- what isOnDisk looks at: a copied file is "on disk" while its SOURCE is).
struct mi_memfs {
    mi_copy::Fs fs;
    std::vector<std::string> blacklist;
    std::string err;
    // the content-aware commit (mi_memfs_commit_layer with a ctx): ONE batch, kept between commits -- its arena and tables
    // are sized by the first commit and reused by the next (mi_batch_reset)
    mi_batch* batch = nullptr;
    mi_ctx* batch_ctx = nullptr;
    std::vector<mi_ctx*> batch_ctxs;     // ... or, behind the same handle, one batch per ctx (mi_memfs_commit_layer_n: a group)
    mi_index* index = nullptr;           // mi_memfs_set_index: every content-aware commit adds its batch's chunks
    mi_commit_stats last;                // of the last mi_memfs_commit_layer
    bool went_windowed = false;          // the scanned tree did not fit the device (the next full scan goes window by window at once)
    mi_memfs() { memset(&last, 0, sizeof last); }
    ~mi_memfs() { if (batch) mi_batch_free(batch); }
};

// MI_MEMFS_TIMING=1: one line per merge / scan on stderr
static bool memfs_timing() {
    static const bool on = [] { const char* e = getenv("MI_MEMFS_TIMING"); return e && *e && *e != '0'; }();
    return on;
}
struct MemfsTimer {
    const char* what; mi_copy::Fs& fs; uint64_t n; uint64_t calls0, memo0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    MemfsTimer(const char* w, mi_copy::Fs& f, uint64_t n_) : what(w), fs(f), n(n_), calls0(f.n_anc_calls), memo0(f.n_anc_memo) {}
    ~MemfsTimer() {
        if (!memfs_timing()) return;
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "mi_memfs %s: %llu entries in %.3f s (%.2f us each), addAncestors %llu calls, %llu answered by the memo, %zu nodes held\n",
                what, (unsigned long long)n, s, n ? s * 1e6 / (double)n : 0.0, (unsigned long long)(fs.n_anc_calls - calls0),
                (unsigned long long)(fs.n_anc_memo - memo0), fs.nodes.size());
    }
};
static mi_copy_layer* memfs_take_layer(mi_memfs* m) {
    const auto t0 = std::chrono::steady_clock::now();
    mi_copy_layer* l = new mi_copy_layer();
    l->nodes = m->fs.sorted_layer();
    m->fs.clear_layer();
    if (memfs_timing())
        fprintf(stderr, "mi_memfs: a layer of %zu entries put in commit order and handed over in %.3f s\n", l->nodes.size(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return l;
}
static int memfs_fail(mi_memfs* m) {
    m->err = m->fs.err;
    const int rc = m->fs.rc;
    m->fs.rc = MI_OK;                                                             // the handle stays usable, like the
    m->fs.err.clear();                                                            // reference's MemFS after an error
    m->fs.clear_layer();
    return rc;
}

extern "C" int mi_memfs_create(const char* root, const char* const* blacklist, uint64_t n_blacklist, int64_t now_sec,
                               mi_memfs** out) {
    if (!root || !out || (n_blacklist && !blacklist)) return MI_ERR_INVALID;
    struct stat st;
    if (lstat(root, &st) != 0) {
        mi_set_error(nullptr, (std::string("unable to stat root dir: ") + root).c_str());      // mi_last_error(NULL)
        return MI_ERR_IO;
    }
    mi_memfs* m = new mi_memfs();
    m->fs.root = mi_walk::abs_path(root);
    m->fs.now = now_sec;
    mi_copy::Node r;
    r.e.kind = 0;
    r.e.mode = st.st_mode; r.e.mtime = st.st_mtime; r.e.uid = st.st_uid; r.e.gid = st.st_gid;
    r.src = m->fs.root;
    m->fs.t.root.ref = m->fs.keep(r);
    for (uint64_t i = 0; i < n_blacklist; ++i) m->blacklist.push_back(blacklist[i] ? blacklist[i] : "");
    *out = m;
    return MI_OK;
}
extern "C" void mi_memfs_free(mi_memfs* m) { delete m; }
extern "C" const char* mi_memfs_error(const mi_memfs* m) { return m ? m->err.c_str() : ""; }
extern "C" int mi_memfs_set_clock(mi_memfs* m, int64_t now_sec) {
    if (!m) return MI_ERR_INVALID;
    m->fs.now = now_sec;
    return MI_OK;
}
extern "C" int mi_memfs_reset(mi_memfs* m) {                                      // MemFS.Reset (:127-130)
    if (!m) return MI_ERR_INVALID;
    m->fs.t.root.children.clear();
    m->fs.t.shape_reset();
    return MI_OK;
}

// MemFS.UpdateFromTarReader (:165-255) on a layer's entries (mi_tar_entries); tar_fd >= 0: untar = true, a regular file's
// bytes are [data_offsets[j], +size) of that descriptor
static int memfs_update(mi_memfs* m, const mi_tree_entry* layer, uint64_t n_layer, int tar_fd, const uint64_t* data_offsets,
                        bool untar, uint64_t* n_merged) {
    mi_copy::Fs& fs = m->fs;
    MemfsTimer timer(untar ? "untar + merge" : "merge", fs, n_layer);
    fs.clear_layer();
    std::map<std::string, struct timespec> modtimes;                              // parent directories, to be put back
    std::string uerr;
    const mi_walk::MountTable& mt = mi_walk::mountpoints();
    if (!mt.error.empty()) { m->err = "check if mounted: " + mt.error; return MI_ERR_IO; }
    std::vector<std::string> bl;
    for (const std::string& b : m->blacklist) bl.push_back(mi_walk::abs_path(b));
    std::vector<std::string> below_mount;                                         // "<target>/": isMounted's prefixes
    for (const std::string& t : mt.targets) below_mount.push_back(t.back() == '/' ? t : t + "/");
    // "is this directory at or below a mountpoint" is asked once per directory, not per entry (entries come
    // directory by directory): an entry is mounted iff it IS a target or its directory lies at or below one
    std::string mdir;
    bool mdir_set = false, mdir_below = false;
    std::string on_disk_buf;
    auto skipped = [&](const mi_tree_entry& e, const std::string& p) {            // shouldSkip + IsMounted (:190-199)
        const size_t cut = p.find_last_of('/');
        if (p.compare(cut == std::string::npos ? 0 : cut + 1, 8, ".wh..wh.") == 0) return true;
        if (e.kind > 3) return true;
        if (fs.root == "/") on_disk_buf = p; else { on_disk_buf = fs.root; if (p != "/") on_disk_buf += p; }
        const std::string& on_disk = on_disk_buf;
        if (!bl.empty() && mi_walk::is_descendant_of_any_cleaned(on_disk, bl, bl)) return true;   // (bl: AbsPath'ed above, once)
        if (mt.targets.empty()) return false;
        if (mt.targets.count(on_disk)) return true;
        const size_t dcut = on_disk.find_last_of('/');
        if (dcut == std::string::npos) return false;
        if (!mdir_set || mdir.size() != dcut + 1 || memcmp(mdir.data(), on_disk.data(), dcut + 1) != 0) {
            mdir.assign(on_disk, 0, dcut + 1);                                   // the directory with its slash
            mdir_set = true;
            mdir_below = false;
            for (const std::string& t : below_mount)
                if (mi_walk::has_prefix(mdir, t)) { mdir_below = true; break; }
        }
        return mdir_below;
    };
    auto disk_path = [&](const std::string& p) { return fs.root == "/" ? p : fs.root + (p == "/" ? "" : p); };
    std::string src_buf;
    auto one = [&](const mi_tree_entry& e, const std::string& p, uint64_t j) {
        if (untar && !mi_untar::one_item(fs.root, disk_path(p), e, tar_fd, data_offsets ? data_offsets[j] : 0, &uerr)) {
            fs.fail(MI_ERR_IO, "untar one item " + disk_path(p) + ": " + uerr);
            return;
        }
        mi_copy::Node n;
        n.e.relpath = p == "/" ? "" : p.substr(1);
        n.e.kind = e.kind; n.e.mode = e.mode; n.e.mtime = e.mtime_sec; n.e.uid = e.uid; n.e.gid = e.gid; n.e.size = e.size;
        if (e.link_target) {
            n.e.has_link = true;                                                  // "Docker hard link names are all absolute,
            n.e.link = e.kind == 3 ? mi_walk::abs_path(e.link_target) : e.link_target;   //  but don't have a leading slash"
        }
        // src: the reference passes AbsPath(hdr.Name) (:225) -- the path the entry is untarred to when the root is "/",
        // as in every real build; under another root that is filepath.Join(root, name), and isOnDisk must look THERE
        if (fs.root == "/") src_buf = p; else { src_buf = fs.root; if (p != "/") src_buf += p; }
        fs.maybe_add(src_buf, p, std::move(n), false);
    };
    std::map<std::string, uint64_t> hardlinks;
    // (with room to spare: the first header a later step adds must not be the one that moves a million nodes)
    fs.nodes.reserve(fs.nodes.size() + n_layer + n_layer / 4 + 1024);
    fs.layer.count_only = true;                                                   // (nothing reads this layer: its size is reported)
    fs.layer.reserve(n_layer + n_layer / 8 + 16);                                 // (its entries: the layer's paths and their ancestors)
    std::string p;                                                                // one buffer for every header's path
    for (uint64_t j = 0; j < n_layer && !fs.rc; ++j) {
        mi_walk::abs_path_of_rel_into(layer[j].relpath ? layer[j].relpath : "", &p);
        if (skipped(layer[j], p)) continue;
        if (untar) {                                                              // "Record the modtime of the parent directory to
            const std::string parent = mi_walk::dir_of(disk_path(p));             //  reset it after we deal with all of the other files"
            if (!modtimes.count(parent)) {
                struct stat st;
                if (lstat(parent.c_str(), &st) != 0) {
                    fs.fail(MI_ERR_IO, "stat parent dir of " + disk_path(p) + ": " + strerror(errno));
                    break;
                }
                modtimes[parent] = st.st_mtim;
            }
        }
        if (layer[j].kind == 3) { hardlinks[p] = j; continue; }
        one(layer[j], p, j);
    }
    for (auto& kv : hardlinks) {
        if (fs.rc) break;
        one(layer[kv.second], kv.first, kv.second);
    }
    const bool untar_failed = fs.rc == MI_ERR_IO;
    if (fs.rc) { if (!untar_failed) fs.err = "add hdr from tar to layer: " + fs.err; return memfs_fail(m); }
    for (auto& kv : modtimes) {                                                   // "Reset the mod times on all of the directory we changed"
        struct timespec ts[2] = {kv.second, kv.second};
        if (utimensat(AT_FDCWD, kv.first.c_str(), ts, 0) != 0) {
            m->err = "chtimes on parent directory " + kv.first + ": " + strerror(errno);
            fs.clear_layer();
            return MI_ERR_IO;
        }
    }
    if (n_merged) *n_merged = fs.layer.size();                                    // "Merged %d headers from tar to memfs"
    fs.clear_layer();
    return MI_OK;
}

extern "C" int mi_memfs_update_from_entries(mi_memfs* m, const mi_tree_entry* layer, uint64_t n_layer,
                                            uint64_t* n_merged) {
    if (!m || (n_layer && !layer)) return MI_ERR_INVALID;
    return memfs_update(m, layer, n_layer, -1, nullptr, false, n_merged);
}

// UpdateFromTarReader with untar = true: the entries of a PLAIN tar (mi_tar_entries of tar_path with their data offsets;
// a gzip blob goes through mi_tar_inflate first) are written below the root as untarOneItem does -- whiteouts delete,
// what is already there and similar stays, a directory on a directory is updated in place, anything else is replaced;
// hard links last; the parents' mtimes are put back -- and merged into the tree
extern "C" int mi_memfs_untar(mi_memfs* m, const char* tar_path, const mi_tree_entry* layer, const uint64_t* data_offsets,
                              uint64_t n_layer, uint64_t* n_merged) {
    if (!m || !tar_path || (n_layer && (!layer || !data_offsets))) return MI_ERR_INVALID;
    const int fd = open(tar_path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { m->err = std::string("open tar file ") + tar_path + ": " + strerror(errno); return MI_ERR_IO; }
    const int rc = memfs_update(m, layer, n_layer, fd, data_offsets, true, n_merged);
    close(fd);
    return rc;
}

// MemFS.createLayerByScan (:315-341) on a walk of the root (mi_tree_walk / mi_batch_add_tree with MI_TREE_SCAN,
// rel_base = root): every walked path through maybeAddToLayer with createWhiteout = true
// from_batch: the walk is a batch's (mi_batch_add_tree) -- an entry's file_index is its row there, and the layer's nodes keep it
// wt (optional): the walk's own record of the same entries (inode stamps, which files were not staged because their content is known)
static int memfs_scan(mi_memfs* m, const mi_tree_entry* walked, uint64_t n, const void* roots, uint64_t root_stride, bool from_batch,
                      mi_copy_layer** out, uint64_t* n_entries, const mi_walk::Tree* wt = nullptr) {
    if (!m || (n && !walked) || !out) return MI_ERR_INVALID;
    mi_copy::Fs& fs = m->fs;
    MemfsTimer timer("scan", fs, n);
    fs.clear_layer();
    // pass 1: the nodes the tree holds for the walk's paths are marked "listed by this scan's walk"
    // (a reserved mark: the readers of THIS walk -- wt -- have marked the entries they found held with it)
    const uint32_t mark = wt && wt->want_stamps && fs.reserved_mark ? fs.reserved_mark : mi_copy::next_scan_mark();
    fs.reserved_mark = 0;
    std::string p;                                                              // one buffer for every path of the walk
    const bool have_flags = wt && wt->want_stamps;
    auto held_by_walk = [&](uint64_t i) { return have_flags && i < wt->known.size() && (wt->known[i] & mi_walk::kEntryHeld); };
    for (uint64_t i = 0; i < n; ++i) {
        if (held_by_walk(i)) continue;                                           // (marked where it was stat'ed)
        mi_walk::abs_path_of_rel_into(walked[i].relpath ? walked[i].relpath : "", &p);
        if (mi_memtree::Node* nd = fs.t.find(p)) nd->seen = mark;
    }
    std::unordered_set<std::string> on_walk;                                    // the fallback, built when first asked
    bool on_walk_built = false;
    fs.scan_mark = mark;
    fs.listed_by_walk = [&](const std::string& q) {
        if (!on_walk_built) {
            if (memfs_timing()) fprintf(stderr, "mi_memfs scan: the set of the walk's paths is built (asked about %s)\n", q.c_str());
            on_walk.reserve(n * 2);
            for (uint64_t i = 0; i < n; ++i) on_walk.insert(mi_walk::abs_path_of_rel(walked[i].relpath ? walked[i].relpath : ""));
            on_walk_built = true;
        }
        return on_walk.count(q) != 0;
    };
    struct Unset { mi_copy::Fs& f; ~Unset() { f.scan_mark = 0; f.listed_by_walk = nullptr; } } unset{fs};
    for (uint64_t i = 0; i < n && !fs.rc; ++i) {
        if (held_by_walk(i)) continue;                                           // a regular file, header and content as the tree holds them
        const mi_tree_entry& e = walked[i];
        mi_walk::abs_path_of_rel_into(e.relpath ? e.relpath : "", &p);
        const bool lazy = !roots && fs.job && from_batch && e.kind == 1 && e.file_index >= 0;   // (the scan may still be running)
        const uint8_t* content_root =
            roots && e.kind == 1 && e.file_index >= 0 ? (const uint8_t*)roots + (uint64_t)e.file_index * root_stride : nullptr;
        const bool have_wt = wt && wt->want_stamps && i < wt->stamps.size();
        const bool hashed_now = (from_batch || roots) && e.kind == 1 && e.file_index >= 0 && have_wt;   // (a batch's row, or a window's)
        const bool held = fs.holds_similar(p, e, content_root, lazy ? e.file_index : -1, hashed_now ? &wt->stamps[i] : nullptr,
                                           have_wt && wt->known[i]);       // (not read again: the content the tree knows)
        if (fs.rc) break;
        if (held) {                                                               // nothing to add; a directory's deletions
            if (e.kind == 0) fs.whiteout_missing_children(p);                     // are still looked for (maybe_add's tail)
            continue;
        }
        mi_copy::Node nd;
        nd.e.relpath = p == "/" ? "" : p.substr(1);
        nd.e.kind = e.kind; nd.e.mode = e.mode; nd.e.mtime = e.mtime_sec; nd.e.uid = e.uid; nd.e.gid = e.gid; nd.e.size = e.size;
        if (e.link_target) { nd.e.has_link = true; nd.e.link = e.link_target; }
        if (roots && e.kind == 1 && e.file_index >= 0) {
            nd.has_root = true;
            memcpy(nd.root, (const uint8_t*)roots + (uint64_t)e.file_index * root_stride, 32);
        }
        if (from_batch && e.kind == 1 && e.file_index >= 0) { nd.batch_file = e.file_index; nd.batch_gen = fs.commit_gen; }
        if (lazy) { nd.has_root = true; nd.root_pending = true; }
        fs.stamp_for_next_keep = hashed_now && nd.has_root ? &wt->stamps[i] : nullptr;   // (recorded with the node maybe_add keeps)
        const std::string src = fs.root == "/" ? p : fs.root + (p == "/" ? "" : p);
        fs.maybe_add(src, p, std::move(nd), true);
        fs.stamp_for_next_keep = nullptr;
    }
    if (fs.rc) { fs.err = "add to layer: " + fs.err; return memfs_fail(m); }
    mi_copy_layer* l = memfs_take_layer(m);
    if (n_entries) *n_entries = l->nodes.size();
    *out = l;
    return MI_OK;
}

extern "C" int mi_memfs_add_layer_by_scan(mi_memfs* m, const mi_tree_entry* walked, uint64_t n, const void* roots,
                                          uint64_t root_stride, mi_copy_layer** out, uint64_t* n_entries) {
    return memfs_scan(m, walked, n, roots, root_stride, false, out, n_entries);
}

// MemFS.AddLayerByCopyOps (:276-289): the ops against THIS tree, which they update
extern "C" int mi_memfs_add_layer_by_copy_ops(mi_memfs* m, const mi_copy_op* ops, uint64_t n_ops, mi_copy_layer** out,
                                              uint64_t* n_entries) {
    if (!m || (n_ops && !ops) || !out) return MI_ERR_INVALID;
    m->fs.clear_layer();
    std::string e;
    const int rc = copy_ops_into(m->fs, ops, n_ops, &e);
    if (rc) { if (m->fs.rc) return memfs_fail(m); m->err = e; m->fs.clear_layer(); return rc; }
    mi_copy_layer* l = memfs_take_layer(m);
    if (n_entries) *n_entries = l->nodes.size();
    *out = l;
    return MI_OK;
}

// step.commitLayer (lib/builder/step/common.go:67-111) on the handle: the step's layer by scan (ctx.MustScan) or by its
// copy operations, through tarAndGzipDiffs' pipeline (the layer writer: tar framing, TarDigest, gzip leg, its digest and
// size), folded into the tree; nothing to do = *committed 0.  The walk of a scan happens here, with the handle's
// blacklist.  (MemFS.sync's one-second wait before either stays with the caller.)
//
// ctx == NULL: the reference's commit -- headers decide what changed (tario.IsSimilarHeader), the layer writer reads the
// changed files from disk.
// ctx != NULL: THE SEAM the GPU path exists for (mem_fs.go:315-341,487-503 + common.go:67-111 in one flow):
//     walk + stage   the root (must_scan) or the ops' sources are walked and every regular file they list is staged into
//                    ONE batch while the walk goes on (mi_batch_add_tree's way: small files read where they are listed,
//                    large ones by the reader threads) -- each file is opened and read ONCE;
//     scan           Gear CDC + SHA-256 per chunk + per-file chunk roots on the GPU (mi_batch_run);
//     diff           createLayerByScan / addToLayer with the roots: a path is in the layer if its header changed OR its
//                    content did (same size, same second, other bytes: invisible to the reference); unchanged files whose
//                    node had no root yet take theirs, so the next commit can tell;
//     write          the layer writer frames the tar; a regular file's bytes come from HBM -- the very bytes the root
//                    describes (mi_layer_add_batch_file) -- not from a second read of a file that may have moved on;
//     index          optionally (mi_memfs_set_index) the batch's chunk digests join the chunk index.
// MI_COMMIT_VERIFY=0 (measurements): no sums at the source, no check in the layer writer
static int memfs_verify_on() {
    static const int on = [] { const char* v = getenv("MI_COMMIT_VERIFY"); return !(v && *v == '0') ? 1 : 0; }();
    return on;
}
static double secs_since(const std::chrono::steady_clock::time_point& t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
static int memfs_commit_write(mi_memfs* m, mi_copy_layer* cl, uint64_t ne, const mi_layer_config* cfg, mi_layer_result* res,
                              mi_batch* batch, bool pipelined) {
    std::vector<mi_tree_entry> ents(ne ? ne : 1);
    std::vector<const char*> srcs(ne ? ne : 1);
    int rc = mi_copy_layer_entries(cl, ents.data(), srcs.data(), ne);
    mi_layer* lw = nullptr;
    if (rc) m->err = "failed to generate diff layer: layer entries";
    if (!rc && (rc = mi_layer_begin(cfg, &lw))) m->err = "failed to generate diff layer: the layer writer refused its configuration";
    if (lw && pipelined) mi_layer_set_pipelined(lw, 1);        // a file's bytes are waited for where they have not landed yet
    for (uint64_t i = 0; i < ne && !rc; ++i) {
        const mi_copy::Node& nd = cl->nodes[i];
        if (ents[i].kind == 1 && ents[i].file_index >= 0) { ++m->last.n_layer_files; m->last.layer_file_bytes += ents[i].size; }
        if (batch && ents[i].kind == 1 && ents[i].file_index >= 0 && nd.batch_file >= 0 && nd.batch_gen == m->fs.commit_gen)
            rc = mi_layer_add_batch_file(lw, &ents[i], batch, (uint64_t)nd.batch_file);
        else
            rc = mi_layer_add(lw, &ents[i], ents[i].kind == 1 && srcs[i] && srcs[i][0] ? srcs[i] : nullptr);
        if (rc) m->err = std::string("failed to generate diff layer: write diffs: commit layer: ") + mi_layer_error(lw);
    }
    if (!rc) {
        rc = mi_layer_finish(lw, res);
        if (rc) m->err = std::string("failed to generate diff layer: ") + mi_layer_error(lw);
    }
    if (lw) {
        uint64_t o = 0, by = 0;
        mi_layer_io_counts(lw, &o, &by);
        m->last.files_opened += o;
        m->last.file_bytes_read += by;
        mi_layer_verify_counts(lw, &m->last.n_verified_files, &m->last.verified_bytes, &m->last.n_refetched);
        mi_layer_free(lw);
    }
    return rc;
}

// A tree that does not fit the device: the chunk roots of its regular files, computed in WINDOWS -- runs of files (walk order) the
// batch has room for; each window is staged, scanned and forgotten.  roots: 32 bytes per regular file, by the file's ordinal
// (a scanned tree: mi_tree_walk's numbering; COPY sources: numbered across the ops' walks).  The window: MI_COMMIT_WINDOW_MB, or
// half of what the batch's arena could be grown to before it failed, at least 64 MiB; a single file larger than the window is
// tried alone.
struct WindowFile { std::string path; uint64_t size, ordinal; };
static int commit_roots_by_windows(mi_memfs* m, mi_batch* b, const std::vector<WindowFile>& files, std::vector<uint8_t>* roots) {
    uint64_t window = 0;
    if (const char* e = getenv("MI_COMMIT_WINDOW_MB")) window = (uint64_t)atol(e) << 20;
    if (!window) {
        uint64_t room = 0;
        mi_batch_arena_room(b, &room);
        window = room / 2;
    }
    if (window < (64ull << 20) && !getenv("MI_COMMIT_WINDOW_MB")) window = 64ull << 20;
    roots->assign(files.size() * 32, 0);
    std::vector<const char*> cpaths;
    std::vector<uint64_t> sizes;
    auto run = [&](size_t lo, size_t hi) -> int {
        if (lo == hi) return MI_OK;
        cpaths.clear(); sizes.clear();
        for (size_t k = lo; k < hi; ++k) { cpaths.push_back(files[k].path.c_str()); sizes.push_back(files[k].size); }
        int rc = mi_batch_reset(b);
        if (!rc) rc = mi_batch_add_paths(b, hi - lo, cpaths.data(), sizes.data(), nullptr);
        if (!rc) rc = mi_batch_run(b);
        std::vector<uint8_t> r((hi - lo) * 32);
        if (!rc) rc = mi_batch_roots(b, r.data(), hi - lo);
        if (!rc) {
            uint64_t nc = 0;
            mi_batch_counts(b, nullptr, &nc, nullptr);
            m->last.n_chunks += nc;
            for (size_t k = lo; k < hi; ++k) memcpy(roots->data() + files[k].ordinal * 32, r.data() + (k - lo) * 32, 32);
            ++m->last.n_windows;
        }
        return rc;
    };
    size_t lo = 0;
    uint64_t held = 0;
    for (size_t i = 0; i < files.size(); ++i) {
        if (held && held + files[i].size > window) { const int rc = run(lo, i); if (rc) return rc; lo = i; held = 0; }
        held += files[i].size;
        m->last.n_scanned_files += 1;
        m->last.scanned_bytes += files[i].size;
    }
    return run(lo, files.size());
}

static int memfs_commit(mi_memfs* m, mi_ctx* const* ctxs, uint32_t n_ctx, int must_scan, const mi_copy_op* ops, uint64_t n_ops,
                        const mi_layer_config* cfg, mi_layer_result* res, mi_copy_layer** layer_out, int* committed);
extern "C" int mi_memfs_commit_layer(mi_memfs* m, mi_ctx* ctx, int must_scan, const mi_copy_op* ops, uint64_t n_ops,
                                     const mi_layer_config* cfg, mi_layer_result* res, mi_copy_layer** layer_out,
                                     int* committed) {
    return memfs_commit(m, ctx ? &ctx : nullptr, ctx ? 1 : 0, must_scan, ops, n_ops, cfg, res, layer_out, committed);
}
// THE COMMIT OVER SEVERAL GPUs (north_star: "file batches shard across the 8 GPUs of one node").  One ctx per
// GPU; what the walk hands over -- a directory's block of small files, a large file -- goes to the GPU with the fewest bytes so
// far (the streaming form of longest-processing-time-first: a walk does not know its files in advance); every GPU stages through
// its own reader threads and PCIe link, scans its share, and the committing thread gets the roots back in the walk's order; the
// tar writer reads each file from the GPU that holds it; the chunk index (on one of the ctxs) takes the other GPUs' digests
// through the host (32 bytes per chunk).  Everything else -- the diff, the pipelining, the sums, TRUST_CTIME, the windows of a
// tree that does not fit -- is the one-GPU commit's: behind the handle the n batches look like one (mi_batch_group_begin).
// n_ctx = 1 is mi_memfs_commit_layer, n_ctx = 0 the reference's commit.  A file of 256 MiB and more is split over the GPUs as parts
// (mi_api.hip: the group's mi_batch_add_paths and group_resolve_parts).
// UNMEASURED on more than one physical GPU (no such box in this pool): tested with n ctxs on one device and on the HIP double.
extern "C" int mi_memfs_commit_layer_n(mi_memfs* m, mi_ctx* const* ctxs, uint32_t n_ctx, int must_scan, const mi_copy_op* ops,
                                       uint64_t n_ops, const mi_layer_config* cfg, mi_layer_result* res,
                                       mi_copy_layer** layer_out, int* committed) {
    if (n_ctx && !ctxs) return MI_ERR_INVALID;
    if (n_ctx > 64) return MI_ERR_INVALID;
    return memfs_commit(m, ctxs, n_ctx, must_scan, ops, n_ops, cfg, res, layer_out, committed);
}
static int memfs_commit(mi_memfs* m, mi_ctx* const* ctxs, uint32_t n_ctx, int must_scan, const mi_copy_op* ops, uint64_t n_ops,
                        const mi_layer_config* cfg, mi_layer_result* res, mi_copy_layer** layer_out, int* committed) {
    mi_ctx* const ctx = n_ctx ? ctxs[0] : nullptr;
    if (!m || !cfg || !res || !committed || (n_ops && !ops)) return MI_ERR_INVALID;
    *committed = 0;
    if (layer_out) *layer_out = nullptr;
    memset(&m->last, 0, sizeof m->last);
    if (!must_scan && n_ops == 0) return MI_OK;                                   // "Nothing to do, return."
    const auto t_all = std::chrono::steady_clock::now();
    const uint64_t opens0 = mi_io::content_opens.load(), bytes0 = mi_io::content_bytes.load();
    mi_copy::Fs& fs = m->fs;
    fs.n_content_changed = fs.n_roots_learned = 0;
    fs.reserved_mark = 0;
    if (++fs.commit_gen == 0) fs.commit_gen = 1;
    mi_copy_layer* cl = nullptr;
    uint64_t ne = 0;
    int rc;
    const char* how = must_scan ? "create layer by scan: " : "create layer by copy ops: ";
    auto fail_with = [&](int code, const std::string& what) {
        m->err = std::string("failed to generate diff layer: write diffs: ") + how + what;
        return code;
    };
    mi_batch* b = nullptr;
    uint64_t moves0 = 0;
    if (ctx) {
        const std::vector<mi_ctx*> want(ctxs, ctxs + n_ctx);
        if (m->batch && (m->batch_ctx != ctx || m->batch_ctxs != want)) { mi_batch_free(m->batch); m->batch = nullptr; }
        if (m->batch) mi_batch_arena_info(m->batch, nullptr, nullptr, &moves0);
        if (m->batch) rc = mi_batch_reset(m->batch);
        else {
            rc = n_ctx > 1 ? mi_batch_group_begin(ctxs, n_ctx, &m->batch) : mi_batch_begin(ctx, 0, 0, &m->batch);
            m->batch_ctx = ctx;
            m->batch_ctxs = want;
        }
        if (rc) return fail_with(rc, std::string("gpu scan: ") + mi_last_error(ctx));
        mi_batch_keep_sums(m->batch, memfs_verify_on());                          // the tar is framed from HBM: held against what was read
        b = m->batch;
        // (Until round 5 a first content scan of a tree the handle already knew reserved the arena here, once, for what the tree
        //  lists: an arena that grew in steps drained the reader threads and moved every time.  The arena no longer moves --
        //  mi_arena.hip: an address range mapped piece by piece behind the walk -- so there is nothing to prepare.)
    }
    // PIPELINED (default; MI_COMMIT_PIPELINE=0: one phase after the other): the scan -- the end of staging, the kernels, the
    // roots' way back -- runs on a thread of its own (ScanJob) while this thread computes the layer and frames the tar from
    // the bytes that have landed.  The diff waits for the scan only where a root DECIDES (a file the tree holds with a root
    // and an unchanged header); an all-new layer, a first content scan, a COPY of new files never wait, and their commit
    // costs what the reference's costs: the serial TarDigest.  Roots that were only to be RECORDED are filled in at the end.
    static const bool pipeline_on = [] { const char* e = getenv("MI_COMMIT_PIPELINE"); return !(e && *e == '0'); }();
    static const bool force_windows = [] { const char* e = getenv("MI_COMMIT_FORCE_WINDOWS"); return e && *e == '1'; }();   // (tests: as if the
                                                                                                                            //  tree did not fit)
    mi_copy::ScanJob job;
    bool piped = false;
    bool windowed = false;                                                        // a scanned tree that does not fit the device
    std::vector<uint8_t> roots;
    auto start_scan = [&]() -> int {                                              // scan what has been staged
        uint64_t nf = 0, nbytes = 0;
        mi_batch_counts(b, &nf, nullptr, &nbytes);
        m->last.n_scanned_files = nf;
        m->last.scanned_bytes = nbytes;
        if (!nf) return MI_OK;
        if (pipeline_on) {
            job.start(b, nf);
            fs.job = &job;
            piped = true;
            return MI_OK;
        }
        const auto t0 = std::chrono::steady_clock::now();
        int r = mi_batch_run(b);
        if (!r) { roots.resize(nf * 32); r = mi_batch_roots(b, roots.data(), nf); }
        m->last.s_scan = secs_since(t0);
        if (!r) mi_batch_counts(b, nullptr, &m->last.n_chunks, nullptr);
        return r;
    };
    // the scan thread is joined and every promised root settled before anything returns (also on the error paths)
    auto end_scan = [&](std::vector<mi_copy::Node>* layer_nodes) -> int {
        if (!piped) return MI_OK;
        const uint8_t* r = job.wait_roots();
        job.join();
        fs.job = nullptr;
        piped = false;
        fs.settle_pending(r, layer_nodes);
        m->last.s_scan = job.seconds;
        m->last.n_chunks = job.n_chunks;
        return job.rc;
    };
    if (must_scan) {
        std::vector<const char*> bl;
        for (const std::string& s : m->blacklist) bl.push_back(s.c_str());
        uint64_t n = 0;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<mi_tree_entry> walked;
        mi_tree* t = nullptr;
        const mi_walk::Tree* wt = nullptr;
        mi_walk::Tree listing;                                                    // the walk of a tree that is scanned in windows
        if (b) {
            struct timespec now;
            clock_gettime(CLOCK_REALTIME, &now);
            fs.commit_started_ns = (int64_t)now.tv_sec * 1000000000ll + now.tv_nsec;
            if (fs.trust_ctime) {                                                 // files whose inode says "unchanged" are not read
                fs.reserved_mark = mi_copy::next_scan_mark();                     // (... and found held are marked by the readers)
                mi_walk::Tree* wtree = nullptr;
                std::string werr;
                struct Freeze { mi_memtree::Tree& t; Freeze(mi_memtree::Tree& x) : t(x) { t.frozen = true; } ~Freeze() { t.frozen = false; } } freeze(fs.t);   // (the readers look the tree up)
                rc = mi_walk::scan_walk_batch_filtered(b, fs.root, m->blacklist,
                        [&fs, mark = fs.reserved_mark](const std::string& path, const struct stat& sb, const mi_walk::InodeStamp& st) {
                            return fs.content_is_known(path, sb, st, mark);
                        },
                        &wtree, &werr);
                if (rc && rc != MI_ERR_NOMEM) { fs.reserved_mark = 0; return fail_with(rc, "walk " + fs.root + ": " + (werr.empty() ? mi_last_error(ctx) : werr)); }
                if (!rc) n = wtree->entries.size();
            } else if (m->went_windowed || force_windows) {
                rc = MI_ERR_NOMEM;                                                // (it did not fit last time and everything is read again:
            } else {                                                              //  straight to the windows)
                rc = mi_batch_add_tree(b, fs.root.c_str(), fs.root.c_str(), bl.empty() ? nullptr : bl.data(), bl.size(), MI_TREE_SCAN, &n);
                if (rc && rc != MI_ERR_NOMEM) return fail_with(rc, "walk " + fs.root + ": " + mi_last_error(ctx));
            }
            if (rc == MI_ERR_NOMEM) {
                m->went_windowed = true;
                // THE TREE DOES NOT FIT THE DEVICE.  The roots are computed window by window -- the walk once more, without a
                // batch; its regular files through the batch in runs the device has room for -- and the layer's files are
                // read from disk by the writer: a second read for those, the price of a tree larger than HBM.
                windowed = true;
                fs.reserved_mark = 0;                                             // (another walk: another mark)
                (void)mi_batch_reset(b);
                std::string werr;
                rc = mi_walk::scan_walk_listing(fs.root, m->blacklist, &listing, &werr);
                if (rc) return fail_with(rc, "walk " + fs.root + ": " + werr);
                n = listing.entries.size();
                walked.resize(n ? n : 1);
                for (uint64_t i = 0; i < n; ++i) {                                // (mi_tree_entries' rows of the same record)
                    const mi_walk::Entry& e = listing.entries[i];
                    mi_tree_entry& o = walked[i];
                    memset(&o, 0, sizeof o);
                    o.relpath = e.relpath.c_str();
                    o.link_target = e.has_link ? e.link.c_str() : nullptr;
                    o.file_index = e.file_index; o.size = e.size; o.mtime_sec = e.mtime; o.mode = e.mode; o.kind = e.kind;
                    o.uid = e.uid; o.gid = e.gid;
                }
                wt = &listing;                                                    // (the inode stamps: a window's files are hashed files)
                std::vector<WindowFile> wf;
                for (uint64_t i = 0; i < n; ++i)
                    if (walked[i].kind == 1 && walked[i].file_index >= 0)
                        wf.push_back({fs.root == "/" ? "/" + std::string(walked[i].relpath) : fs.root + "/" + walked[i].relpath,
                                      walked[i].size, (uint64_t)walked[i].file_index});
                if ((rc = commit_roots_by_windows(m, b, wf, &roots)))
                    return fail_with(rc, std::string("gpu scan (in windows): ") + mi_last_error(ctx));
                if (m->last.n_windows <= 1) m->went_windowed = false;             // (it has shrunk to one window's worth: one batch next time)
            } else {
                wt = (const mi_walk::Tree*)*mi_batch_tree_slot(b);
                walked.resize(n ? n : 1);
                rc = mi_batch_tree_entries(b, walked.data(), n);
            }
        } else {
            rc = mi_tree_walk(fs.root.c_str(), fs.root.c_str(), bl.empty() ? nullptr : bl.data(), bl.size(), MI_TREE_SCAN, &t, &n);
            if (rc) return fail_with(rc, "walk " + fs.root);
            walked.resize(n ? n : 1);
            rc = mi_tree_entries(t, walked.data(), n);
        }
        m->last.n_walked = n;
        m->last.s_walk_stage = secs_since(t0);
        if (!rc && b && !windowed && (rc = start_scan())) { if (t) mi_tree_free(t); return fail_with(rc, std::string("gpu scan: ") + mi_last_error(ctx)); }
        const auto t1 = std::chrono::steady_clock::now();
        if (!rc) rc = memfs_scan(m, walked.data(), n, roots.empty() ? nullptr : roots.data(), 32, b != nullptr && !windowed, &cl, &ne, wt);
        m->last.s_diff = secs_since(t1);
        if (wt && wt->want_stamps && fs.trust_ctime)                              // files the walk did not read: their content is known
            for (size_t i = 0; i < wt->known.size() && i < n; ++i) m->last.n_content_trusted += wt->known[i] ? 1 : 0;
        if (t) mi_tree_free(t);
        if (rc) { end_scan(nullptr); return fail_with(rc, m->err); }
    } else {
        fs.clear_layer();
        CopyPlan plan;
        const auto t0 = std::chrono::steady_clock::now();
        if (!(b && force_windows)) copy_ops_plan(fs, ops, n_ops, b, &plan);
        if (b && (force_windows || plan.err_rc == MI_ERR_NOMEM)) {
            // THE SOURCES DO NOT FIT THE DEVICE: planned again without a batch (the walks alone), the files numbered across the
            // ops' walks, their roots window by window; the writer reads the layer's files from disk
            windowed = true;
            (void)mi_batch_reset(b);
            plan = CopyPlan();
            copy_ops_plan(fs, ops, n_ops, nullptr, &plan);
            std::vector<WindowFile> wf;
            for (CopyOpPlan& op : plan.ops)
                for (CopySrcPlan& sp : op.srcs)
                    for (mi_walk::Entry& we : sp.walked.entries)
                        if (we.kind == 1 && we.file_index >= 0) {
                            we.file_index = (int64_t)wf.size();
                            wf.push_back({we.relpath == "." ? sp.src : sp.src + "/" + we.relpath, we.size, (uint64_t)we.file_index});
                        }
            if ((rc = commit_roots_by_windows(m, b, wf, &roots)))
                return fail_with(rc, std::string("gpu scan (in windows): ") + mi_last_error(ctx));
        }
        m->last.n_walked = plan.n_walked;
        m->last.s_walk_stage = secs_since(t0);
        // (a plan that stopped at a failure is applied up to it: the failure is the apply step's to raise, in its place.
        //  What was staged until then is scanned all the same -- the ops before the failing one are applied WITH roots.)
        if (b && !windowed && (rc = start_scan())) return fail_with(rc, std::string("gpu scan: ") + mi_last_error(ctx));
        const auto t1 = std::chrono::steady_clock::now();
        std::string e;
        rc = copy_ops_apply(fs, ops, plan, roots.empty() ? nullptr : roots.data(), &e);
        if (rc) {
            end_scan(nullptr);
            if (fs.rc) rc = memfs_fail(m); else { m->err = e; fs.clear_layer(); }
            return fail_with(rc, m->err);
        }
        cl = memfs_take_layer(m);
        ne = cl->nodes.size();
        m->last.s_diff = secs_since(t1);
    }
    m->last.n_layer_entries = ne;
    m->last.n_content_changed = fs.n_content_changed;
    m->last.n_roots_learned = fs.n_roots_learned;
    const auto t2 = std::chrono::steady_clock::now();
    const bool was_piped = piped;
    rc = memfs_commit_write(m, cl, ne, cfg, res, windowed ? nullptr : b, piped);
    m->last.s_write = secs_since(t2);
    {   // the scan's verdict: a file that vanished or shrank since the walk fails the commit here -- after the tree took the
        // layer, as a failing tar write does in the reference (AddLayerByScan updates the tree, then writes)
        const int src = end_scan(&cl->nodes);
        if (src && !rc) { rc = src; m->err = "failed to generate diff layer: write diffs: " + std::string(how) + "gpu scan: " + job.err; }
    }
    m->last.pipelined = was_piped ? 1 : 0;
    if (!rc && b && m->index && m->last.n_scanned_files) {
        // the chunk index (keyvalue.Store seam): which of this commit's chunks no earlier commit held, and how many bytes they
        // are -- what a chunk-addressed store would have to take in for this layer.  A commit over several GPUs feeds the ONE
        // index batch by batch: the batch on the index's own GPU from device memory, the others' digests through the host.
        mi_batch* const* members = &b;
        uint64_t nm = 0;
        mi_batch_group_members(b, &members, nullptr, &nm);
        if (nm == 0) { members = &b; nm = 1; }
        for (uint64_t k = 0; k < nm && !rc; ++k) {
            mi_batch* mb = members[k];
            uint64_t nc = 0, nn = 0, nk = 0;
            mi_batch_counts(mb, nullptr, &nc, nullptr);
            std::vector<uint8_t> known(nc ? nc : 1);
            const mi_chunk_result* rows = nullptr;
            if (nc) rc = mi_batch_chunks_view(mb, &rows, &nc);
            if (!rc && mi_index_same_ctx(m->index, mb)) {
                rc = mi_index_add_batch(m->index, mb, known.data(), nc, &nn, &nk);
            } else if (!rc) {
                std::vector<uint8_t> dg(nc * 32 + 32);
                for (uint64_t i = 0; i < nc; ++i) memcpy(dg.data() + 32 * i, rows[i].sha256, 32);
                rc = mi_index_add_digests(m->index, dg.data(), nc, known.data(), &nn, &nk);
            }
            if (!rc) {
                m->last.n_index_new += nn;
                m->last.n_index_known += nk;
                for (uint64_t i = 0; i < nc; ++i)
                    if (!known[i] && rows[i].dup_of < 0) m->last.index_new_bytes += rows[i].length;
            }
        }
        if (rc) m->err = std::string("failed to generate diff layer: chunk index: ") + mi_last_error(ctx);
    }
    m->last.files_opened += mi_io::content_opens.load() - opens0;
    m->last.file_bytes_read += mi_io::content_bytes.load() - bytes0;
    m->last.s_total = secs_since(t_all);
    if (b) {
        uint64_t moves = 0;
        mi_batch_arena_info(b, &m->last.arena_bytes, &m->last.arena_pieces, &moves);
        m->last.arena_moves = moves - moves0;
        const uint64_t* loads = nullptr;
        uint64_t nm = 0;
        mi_batch_group_members(b, nullptr, &loads, &nm);
        m->last.n_ctxs = nm ? nm : 1;
        m->last.n_split_files = mi_batch_group_splits(b);
        m->last.ctx_bytes_max = m->last.ctx_bytes_min = nm ? loads[0] : m->last.scanned_bytes;
        for (uint64_t k = 1; k < nm; ++k) {
            if (loads[k] > m->last.ctx_bytes_max) m->last.ctx_bytes_max = loads[k];
            if (loads[k] < m->last.ctx_bytes_min) m->last.ctx_bytes_min = loads[k];
        }
    }
    for (mi_copy::Node& nd : cl->nodes) nd.batch_file = -1;                       // (rows of a batch the caller does not hold)
    if (rc) { mi_copy_layer_free(cl); return rc; }
    *committed = 1;
    if (layer_out) *layer_out = cl; else mi_copy_layer_free(cl);
    return MI_OK;
}

extern "C" int mi_memfs_commit_stats(const mi_memfs* m, mi_commit_stats* out) {
    if (!m || !out) return MI_ERR_INVALID;
    *out = m->last;
    return MI_OK;
}
extern "C" int mi_memfs_set_options(mi_memfs* m, uint32_t options) {
    if (!m || (options & ~MI_MEMFS_TRUST_CTIME)) return MI_ERR_INVALID;
    m->fs.trust_ctime = (options & MI_MEMFS_TRUST_CTIME) != 0;
    return MI_OK;
}
extern "C" int mi_memfs_set_index(mi_memfs* m, mi_index* index) {
    if (!m) return MI_ERR_INVALID;
    m->index = index;
    return MI_OK;
}
// The handle's batch, made ahead of its first commit and sized for `bytes` of files in `files` files: fresh device memory costs
// up to 68 ms per GiB to allocate on some boxes (tools/first_use_probe.py) and the ctx's reader threads 55 ms to come up -- a host that knows what
// is coming (the base image it is pulling) pays that beside its own work instead of inside the first commit
extern "C" int mi_memfs_reserve_device(mi_memfs* m, mi_ctx* ctx, uint64_t files, uint64_t bytes) {
    if (!m || !ctx) return MI_ERR_INVALID;
    int rc = MI_OK;
    if (m->batch && m->batch_ctx != ctx) { mi_batch_free(m->batch); m->batch = nullptr; }
    if (!m->batch) {                                                              // (a commit's arena is the piecewise kind: a guess
        rc = mi_batch_begin(ctx, files, 0, &m->batch);                            //  that is too small costs nothing later)
        m->batch_ctx = ctx;
        if (!rc) mi_batch_keep_sums(m->batch, memfs_verify_on());
    } else {
        rc = mi_batch_reset(m->batch);
    }
    if (!rc) rc = mi_batch_reserve_ahead(m->batch, files, bytes);
    if (!rc) rc = mi_batch_prepare_read(m->batch);                                // (the tar writer's windows: 10 ms of pinned allocation)
    if (!rc) mi_batch_expect_host_bytes(m->batch);                                // (the reader threads set up behind the call)
    if (rc) m->err = std::string("reserve device memory: ") + mi_last_error(ctx);
    return rc;
}

// the commit's batch (its arena holds the last scanned tree's bytes) is given back; the next content-aware commit begins anew
extern "C" int mi_memfs_release_device(mi_memfs* m) {
    if (!m) return MI_ERR_INVALID;
    int rc = MI_OK;
    if (m->batch) rc = mi_batch_free(m->batch);
    m->batch = nullptr;
    m->batch_ctx = nullptr;
    return rc;
}
// the chunk root the tree holds for a path (absolute, below the root "/"): what the next isUpdated compares
extern "C" int mi_memfs_root_of(const mi_memfs* m, const char* path, uint8_t* root_out, int* has_root) {
    if (!m || !path || !has_root) return MI_ERR_INVALID;
    *has_root = 0;
    const mi_memtree::Node* nd = const_cast<mi_memtree::Tree&>(m->fs.t).find_walk(mi_walk::abs_path(path));
    if (!nd || nd->ref < 0) return MI_ERR_INVALID;
    const mi_copy::Node& n = m->fs.nodes[nd->ref];
    if (n.has_root) { *has_root = 1; if (root_out) memcpy(root_out, n.root, 32); }
    return MI_OK;
}
extern "C" int mi_copy_layer_roots(const mi_copy_layer* l, uint8_t* roots, uint8_t* has_root, uint64_t cap) {
    if (!l || (cap && (!roots || !has_root))) return MI_ERR_INVALID;
    if (cap < l->nodes.size()) return MI_ERR_CAPACITY;
    for (size_t i = 0; i < l->nodes.size(); ++i) {
        has_root[i] = l->nodes[i].has_root ? 1 : 0;
        if (l->nodes[i].has_root) memcpy(roots + i * 32, l->nodes[i].root, 32); else memset(roots + i * 32, 0, 32);
    }
    return MI_OK;
}

// MemFS.Checkpoint (:132-185): the sources a later stage will COPY --from are moved aside, below new_root, with the
// layout they have below the root; a pattern is expanded like a COPY source, a directory's created target gets the
// source's owner, everything copied keeps its own
extern "C" int mi_memfs_checkpoint(mi_memfs* m, const char* new_root, const char* const* sources, uint64_t n_sources) {
    if (!m || !new_root || (n_sources && !sources)) return MI_ERR_INVALID;
    const std::string root = m->fs.root;
    for (uint64_t i = 0; i < n_sources; ++i) {
        const std::string given = sources[i] ? sources[i] : "";
        std::vector<std::string> matches;
        if (!mi_glob::glob(given, &matches) || matches.empty()) matches.assign(1, given);
        for (std::string src : matches) {
            if (src.empty() || src[0] != '/') src = mi_walk::abs_path(root + "/" + src);
            if (!mi_walk::has_prefix(src, root)) {
                m->err = "trim src " + src + ": failed to trim root prefix " + root + " from path " + src;
                return MI_ERR_INVALID;
            }
            const std::string dst = mi_walk::abs_path(std::string(new_root) + "/" + src.substr(root.size()));
            struct stat followed, fi;
            if (stat(src.c_str(), &followed) != 0) { m->err = "stat " + src + ": " + strerror(errno); return MI_ERR_IO; }
            if (lstat(src.c_str(), &fi) != 0) { m->err = "lstat " + src + ": " + strerror(errno); return MI_ERR_IO; }
            mi_copyexec::Copier c;
            c.blacklist = m->blacklist;
            c.dst_dir = {true, fi.st_uid, fi.st_gid, false};
            if (S_ISDIR(followed.st_mode)) {
                if (!c.CopyDir(src, dst)) { m->err = "copy dir " + src + ": " + c.err; return MI_ERR_IO; }
            } else if (!c.CopyFile(src, dst)) {
                m->err = "copy file " + src + ": " + c.err;
                return MI_ERR_IO;
            }
        }
    }
    return MI_OK;
}

// the tree, sorted by path (directories made up by addAncestors included: they are nodes like any other)
extern "C" int mi_memfs_entries(const mi_memfs* m, mi_tree_entry* out, const char** src_paths, uint64_t cap, uint64_t* n_out) {
    if (!m || !n_out || (cap && !out)) return MI_ERR_INVALID;
    std::vector<std::pair<std::string, int64_t>> flat;
    std::function<void(const mi_memtree::Node&, const std::string&)> collect = [&](const mi_memtree::Node& n, const std::string& p) {
        for (auto& kv : n.children) {
            const std::string q = p + "/" + kv.first;
            if (kv.second->ref >= 0) flat.emplace_back(q, kv.second->ref);
            collect(*kv.second, q);
        }
    };
    collect(m->fs.t.root, "");
    std::sort(flat.begin(), flat.end());
    *n_out = flat.size();
    if (cap < flat.size()) return MI_ERR_CAPACITY;
    for (size_t i = 0; i < flat.size(); ++i) {
        const mi_copy::Node& n = m->fs.nodes[flat[i].second];
        memset(&out[i], 0, sizeof out[i]);
        out[i].relpath = n.e.relpath.c_str();
        out[i].link_target = n.e.has_link ? n.e.link.c_str() : nullptr;
        out[i].file_index = -1;
        out[i].size = n.e.size; out[i].mtime_sec = n.e.mtime; out[i].mode = n.e.mode; out[i].kind = n.e.kind;
        out[i].uid = n.e.uid; out[i].gid = n.e.gid;
        if (src_paths) src_paths[i] = n.src.c_str();
    }
    return MI_OK;
}

extern "C" int mi_copy_layer_entries(const mi_copy_layer* l, mi_tree_entry* out, const char** src_paths, uint64_t cap) {
    if (!l || (cap && !out)) return MI_ERR_INVALID;
    if (cap < l->nodes.size()) return MI_ERR_CAPACITY;
    int64_t n_regular = 0;
    for (size_t i = 0; i < l->nodes.size(); ++i) {
        const mi_copy::Node& n = l->nodes[i];
        memset(&out[i], 0, sizeof out[i]);
        out[i].relpath = n.e.relpath.c_str();
        out[i].link_target = n.e.has_link ? n.e.link.c_str() : nullptr;
        out[i].file_index = n.e.kind == 1 && !mi_walk::has_prefix(mi_walk::base_of(n.e.relpath), ".wh.") ? n_regular++ : -1;   // a whiteout has no content
        out[i].size = n.e.size;
        out[i].mtime_sec = n.e.mtime;
        out[i].mode = n.e.mode;
        out[i].kind = n.e.kind;
        out[i].uid = n.e.uid;
        out[i].gid = n.e.gid;
        if (src_paths) src_paths[i] = n.src.c_str();
    }
    return MI_OK;
}

extern "C" void mi_copy_layer_free(mi_copy_layer* l) { delete l; }
