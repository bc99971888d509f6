// mi_hostpath.h -- what the host-side restatements of lib/snapshot share: the walk's entry record, Go's path rules
// (path.Clean / filepath.Join, pathutils.AbsPath / IsDescendantOfAny), the mount table of lib/mountutils.  Header-only;
// no device code.  Users: mi_tree.hip (walks, stateless diffs), mi_memfs.hip (MemFS, copy ops, untar).
#pragma once
#include "../../include/makisu_mi.h"
#include "mi_local.h"

#include <stdio.h>
#include <sys/stat.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <vector>

// What the library's own threads read of file CONTENT, process-wide: descriptors whose bytes were read, and the bytes
// (the walk's directory readers, the reader threads of mi_stage.hip; the layer writer counts its own).  A commit that reads
// every file once shows n_files and the files' bytes here (mi_commit_stats).
namespace mi_io {
extern std::atomic<uint64_t> content_opens, content_bytes;
}

namespace mi_walk {


// what the kernel says about an inode beyond the tar header's fields: the identity of a regular file's CONTENT between two
// scans (MI_MEMFS_TRUST_CTIME: a file whose inode, size, mtime and ctime -- to the nanosecond -- are what they were when its
// content was last hashed is not read again; ctime cannot be set from user space)
struct InodeStamp {
    uint64_t dev = 0, ino = 0;
    int64_t mtime_ns = 0, ctime_ns = 0;
    bool operator==(const InodeStamp& o) const { return dev == o.dev && ino == o.ino && mtime_ns == o.mtime_ns && ctime_ns == o.ctime_ns; }
};

struct Entry {
    std::string relpath, link;
    bool has_link = false;
    int64_t file_index = -1;
    uint32_t mode = 0;
    uint64_t size = 0;
    int64_t mtime = 0;
    uint8_t kind = 0;      // 0 dir, 1 regular, 2 symlink
    uint32_t uid = 0, gid = 0;
};

// a walk's record.  With a batch attached (want_stamps) every entry also has its inode stamp and the answer the caller's
// content_known gave for it, in two arrays beside the entries: the entry itself stays what the layer merge and the scan keep
// per node (a tree of ten million entries pays for every byte of it)
struct Tree {
    std::vector<Entry> entries;
    bool want_stamps = false;
    std::vector<InodeStamp> stamps;       // [i] of entries[i]; regular files (zeros otherwise)
    std::vector<uint8_t> known;           // [i]: what content_known said -- kContentKnown: a regular file the walk did NOT stage,
                                          // its content is known to the caller; | kEntryHeld: ... and the entry is what the
                                          // caller holds for the path, header included (nothing left for the diff to do)
    void push(Entry&& e, const InodeStamp* st = nullptr, uint8_t known_flags = 0) {
        entries.push_back(std::move(e));
        if (want_stamps) { stamps.push_back(st ? *st : InodeStamp()); known.push_back(known_flags); }
    }
};
constexpr uint8_t kContentKnown = 1, kEntryHeld = 2;
// "is this regular file's content known?" -- asked by whoever stats the file (the directory readers: several threads at a time).
// RULE for everything that runs on a directory reader's thread -- this predicate, a block's release callback, anything a later
// round hooks in: NO HIP CALL and NO DESCRIPTOR OF THE PROCESS.  Those threads leave the process's descriptor table
// (close_range(3, ~0, CLOSE_RANGE_UNSHARE) in mi_tree.hip's worker, also when the walk reads nothing): /dev/kfd, /dev/dri/*, the
// host's sockets and pipes do not exist there; a HIP call would find its driver handles closed.
using KnownFn = std::function<uint8_t(const std::string& path_on_disk, const struct stat& st, const InodeStamp& stamp)>;

// Go's path.Clean for a ROOTED path (what path.Join("/", p) returns): single slashes, no "."
// elements, ".." removes the element before it, ".." at the root disappears.
inline std::string clean_rooted(const std::string& p) {
    std::vector<std::string> parts;
    size_t i = 0;
    while (i < p.size()) {
        while (i < p.size() && p[i] == '/') ++i;
        size_t j = i;
        while (j < p.size() && p[j] != '/') ++j;
        if (j > i) {
            const std::string el = p.substr(i, j - i);
            if (el == "..") { if (!parts.empty()) parts.pop_back(); }
            else if (el != ".") parts.push_back(el);
        }
        i = j;
    }
    std::string out;
    for (const std::string& el : parts) out += "/" + el;
    return out.empty() ? "/" : out;
}
// Go's path.Clean for ANY path (filepath.Join's result): as above, but a relative path keeps its leading ".." elements
// and an empty result is ".".
inline std::string clean_any(const std::string& p) {
    if (!p.empty() && p[0] == '/') return clean_rooted(p);
    std::vector<std::string> parts;
    size_t i = 0;
    while (i < p.size()) {
        while (i < p.size() && p[i] == '/') ++i;
        size_t j = i;
        while (j < p.size() && p[j] != '/') ++j;
        if (j > i) {
            const std::string el = p.substr(i, j - i);
            if (el == "..") { if (!parts.empty() && parts.back() != "..") parts.pop_back(); else parts.push_back(el); }
            else if (el != ".") parts.push_back(el);
        }
        i = j;
    }
    std::string out;
    for (const std::string& el : parts) out += (out.empty() ? "" : "/") + el;
    return out.empty() ? "." : out;
}
inline std::string abs_path(const std::string& p) {          // pathutils.AbsPath (lib/pathutils/path.go:41-43)
    return clean_rooted(p);                                  // path.Join("/", strings.TrimRight(p, "/"))
}
// AbsPath of a relative path as the walks and tar readers write them: when it is already clean (no empty, "." or ".."
// element) that is "/" + the path without trailing slashes; anything else takes the general route
inline std::string abs_path_of_rel(const char* rel) {
    if (rel[0] == '.' && rel[1] == 0) return "/";
    size_t n = strlen(rel);
    while (n && rel[n - 1] == '/') --n;
    bool clean = n > 0 && rel[0] != '/';
    for (size_t i = 0; clean && i < n; ++i) {
        if (rel[i] == '/' && (i + 1 >= n || rel[i + 1] == '/')) clean = false;
        if (rel[i] == '.' && (i == 0 || rel[i - 1] == '/')) {
            const size_t k = rel[i + 1] == '.' ? i + 2 : i + 1;
            if (k >= n || rel[k] == '/') clean = false;
        }
    }
    if (!clean) return abs_path(rel);
    std::string out;
    out.reserve(n + 1);
    out.push_back('/');
    out.append(rel, n);
    return out;
}
// the same into a buffer that keeps its capacity (a walk's million paths: no allocation per path)
inline void abs_path_of_rel_into(const char* rel, std::string* out) {
    if (rel[0] == '.' && rel[1] == 0) { out->assign("/"); return; }
    size_t n = strlen(rel);
    while (n && rel[n - 1] == '/') --n;
    bool clean = n > 0 && rel[0] != '/';
    for (size_t i = 0; clean && i < n; ++i) {
        if (rel[i] == '/' && (i + 1 >= n || rel[i + 1] == '/')) clean = false;
        if (rel[i] == '.' && (i == 0 || rel[i - 1] == '/')) {
            const size_t k = rel[i + 1] == '.' ? i + 2 : i + 1;
            if (k >= n || rel[k] == '/') clean = false;
        }
    }
    if (!clean) { *out = abs_path(rel); return; }
    out->assign(1, '/');
    out->append(rel, n);
}
inline std::string dir_of(const std::string& p) {            // path.Dir for clean absolute paths
    size_t i = p.find_last_of('/');
    if (i == std::string::npos) return ".";
    if (i == 0) return "/";
    return p.substr(0, i);
}
inline std::string base_of(const std::string& p) {
    size_t i = p.find_last_of('/');
    return i == std::string::npos ? p : p.substr(i + 1);
}
inline bool has_prefix(const std::string& s, const std::string& pre) {
    return s.size() >= pre.size() && memcmp(s.data(), pre.data(), pre.size()) == 0;
}
inline bool is_descendant_of_any(const std::string& path, const std::vector<std::string>& anc) {
    const std::string p = abs_path(path);
    for (const std::string& a0 : anc) {
        const std::string a = abs_path(a0);
        std::string d = dir_of(p);
        if (d.back() != '/') d += "/";
        if (p == a || a == "/" || has_prefix(d, a + "/")) return true;
    }
    return false;
}
// The same question for the walks, which ask it of EVERY path they list: the blacklist is cleaned once (AbsPath of each entry),
// and a path that is already clean and absolute -- what a walk builds by joining names onto its root -- is compared as it stands:
// no allocation, no path.Join per blacklist entry (a root file system of 114 000 entries with the 26 other top-level directories
// blacklisted: 0.21 s of walk, 0.14 s of it in this question before; IsDescendantOfAny's answer, lib/pathutils/path.go:24-35)
inline bool is_clean_abs(const std::string& p) {
    if (p.empty() || p[0] != '/') return false;
    if (p.size() == 1) return true;
    if (p.back() == '/') return false;
    for (size_t i = 0; i + 1 < p.size(); ++i)
        if (p[i] == '/') {
            if (p[i + 1] == '/') return false;
            if (p[i + 1] == '.') {
                const size_t k = i + 2 < p.size() && p[i + 2] == '.' ? i + 3 : i + 2;
                if (k >= p.size() || p[k] == '/') return false;              // a "." or ".." element
            }
        }
    return true;
}
inline std::vector<std::string> cleaned_paths(const std::vector<std::string>& paths) {
    std::vector<std::string> out;
    for (const std::string& a : paths) out.push_back(abs_path(a));
    return out;
}
inline bool is_descendant_of_any_cleaned(const std::string& path, const std::vector<std::string>& anc_clean,
                                         const std::vector<std::string>& anc_raw) {
    if (anc_clean.empty()) return false;
    if (!is_clean_abs(path)) return is_descendant_of_any(path, anc_raw);
    for (const std::string& a : anc_clean) {
        if (a.size() == 1) return true;                                      // "/": everything is below it
        if (path.size() >= a.size() && memcmp(path.data(), a.data(), a.size()) == 0 && (path.size() == a.size() || path[a.size()] == '/'))
            return true;
    }
    return false;
}
inline std::string rel_to(const std::string& base, const std::string& path) {   // filepath.Rel, descendants only
    const std::string b = abs_path(base), p = abs_path(path);
    if (p == b) return ".";
    if (b == "/") return p.substr(1);
    if (has_prefix(p, b + "/")) return p.substr(b.size() + 1);
    return std::string();                                                        // outside: caller errors
}

// Sorting paths the way Go compares strings (bytewise), for inputs that arrive nearly sorted -- a walk's order, a layer's
// keys in the order a walk put them in: a STABLE NATURAL MERGE SORT of indices.  Maximal non-decreasing runs are found
// first and merged pairwise, so the cost is a few passes instead of log2(n) (users: mi_entries_commit_order in
// mi_tree.hip, the layer's commit order in mi_memfs.hip).
struct KeyRef { const char* p; uint32_t len; };
inline bool key_less(const KeyRef& a, const KeyRef& b) {       // bytewise, like Go's string comparison
    const uint32_t m = a.len < b.len ? a.len : b.len;
    const int c = m ? memcmp(a.p, b.p, m) : 0;
    return c < 0 || (c == 0 && a.len < b.len);
}
// stable: idx[0, n) by keys; tmp = scratch of n words
inline void natural_merge_sort(uint64_t* idx, uint64_t* tmp, size_t n, const KeyRef* keys) {
    if (n < 2) return;
    auto less = [keys](uint64_t a, uint64_t b) { return key_less(keys[a], keys[b]); };
    std::vector<size_t> runs, next;                              // run boundaries: k runs = k + 1 entries
    runs.push_back(0);
    for (size_t i = 1; i < n; ++i)
        if (less(idx[i], idx[i - 1])) runs.push_back(i);
    runs.push_back(n);
    uint64_t *src = idx, *dst = tmp;
    while (runs.size() > 2) {
        next.clear();
        next.push_back(0);
        size_t r = 0;
        for (; r + 2 < runs.size(); r += 2) {
            std::merge(src + runs[r], src + runs[r + 1], src + runs[r + 1], src + runs[r + 2], dst + runs[r], less);
            next.push_back(runs[r + 2]);
        }
        if (r + 1 < runs.size()) {                               // an odd run at the end travels as it is
            std::copy(src + runs[r], src + runs[r + 1], dst + runs[r]);
            next.push_back(runs[r + 1]);
        }
        std::swap(src, dst);
        runs.swap(next);
    }
    if (src != idx) std::copy(src, src + n, idx);
}

// mountutils' table (lib/mountutils/mountutils.go:54-93): targets of /proc/mounts except "/";
// a missing file means "no mountpoints", a line with fewer than four fields is an error that
// fails every walk ("cannot parse mounts file").  MI_MOUNTS_FILE names another file, the way
// the reference's tests swap mountInfo.mountsFile (mountutils_test.go:25-45).
struct MountTable {
    std::set<std::string> targets;
    std::string error;
};
inline const MountTable& mountpoints() {
    static MountTable mt;
    static std::once_flag once;                              // like the reference's sync.Once
    std::call_once(once, [] {
        const char* over = getenv("MI_MOUNTS_FILE");
        const std::string file = over && *over ? over : "/proc/mounts";
        FILE* f = fopen(file.c_str(), "r");
        if (!f) return;                                      // "Skipping mountmanager init"
        char* line = nullptr;                                // getline: overlay mounts list every lower
        size_t cap = 0;                                      // layer on ONE line, easily > 8 KiB
        ssize_t got;
        while ((got = getline(&line, &cap, f)) >= 0) {
            size_t n = (size_t)got;
            while (n && (line[n - 1] == '\n')) line[--n] = 0;
            if (n == 0) continue;
            char* sp1 = strchr(line, ' ');
            char* sp2 = sp1 ? strchr(sp1 + 1, ' ') : nullptr;
            char* sp3 = sp2 ? strchr(sp2 + 1, ' ') : nullptr;
            if (!sp3) { mt.error = "cannot parse mounts file " + file; break; }
            std::string target(sp1 + 1, sp2);
            if (target != "/") mt.targets.insert(target);    // "/" skipped as the reference does
        }
        free(line);
        fclose(f);
    });
    return mt;
}


// the snapshot walk of `src` (scan rules, no blacklist), entries relative to src ("." first); absolute symlink targets
// lose link_root (createHeader trims by the MemFS root).  Defined in mi_tree.hip, beside the walkers.
int scan_walk_collect(const std::string& src, const std::string& link_root, Tree* out, std::string* err);
// the snapshot walk of a root with its blacklist and NO batch, inode stamps recorded (a tree that is scanned in windows)
int scan_walk_listing(const std::string& root, const std::vector<std::string>& blacklist, Tree* out, std::string* err);
int scan_walk_collect_batch(const std::string& src, const std::string& link_root, Tree* out, std::string* err, mi_batch* b);
// mi_batch_add_tree with the scan rules and a say in which files are staged: known(path on disk, stat, stamp) & kContentKnown means
// "this file's content is known, do not read it" -- called from the walk's directory readers, several at a time
int scan_walk_batch_filtered(mi_batch* b, const std::string& root, const std::vector<std::string>& blacklist,
                             const KnownFn& known, Tree** tree_out, std::string* err);

}  // namespace mi_walk
