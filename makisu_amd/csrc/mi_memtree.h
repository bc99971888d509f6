// mi_memtree.h -- the reference's in-memory tree (memFSNode) and the operations every layer-building path goes through.
// Header-only; user: mi_memfs.hip (the MemFS handle: layer merge, scan layer, copy ops).
#pragma once
#include "mi_hostpath.h"

#include <functional>
#include <map>
#include <memory>
#include <string_view>

// The in-memory tree of the reference's own shape -- memFSNode: a header and a children map (lib/snapshot/mem_fs.go:33-47)
// -- with the four operations every layer-building path goes through: isUpdated's walk (:487-503), addAncestors
// (:505-566), contentMemFile.updateMemFS (lib/snapshot/mem_layer.go:50-76) and whiteoutMemFile.updateMemFS (:104-125).
// They walk part by part through nodes of ANY type, and what they do to a file or symlink that has children (or is
// somebody's ancestor) is not what a flat path map would do, so both users -- the layer merge and the copy-op layer --
// share this one.  Nodes carry a caller-owned payload index.
namespace mi_memtree {

struct Node {
    int64_t ref = -1;                          // caller's payload; -1 = none
    uint8_t kind = 0;                          // 0 dir, 1 regular, 2 symlink, 3 hard link, 4 special
    uint32_t seen = 0;                         // caller's mark (the scan: "this scan's walk lists the path")
    std::string link;                          // symlink target
    std::map<std::string, std::unique_ptr<Node>, std::less<>> children;       // std::less<>: looked up by string_view
};

struct Tree {
    Node root;
    std::string err;                           // why the last failing call failed, in the reference's words
    // memLayer.addHeader's bookkeeping (l.files[...] = ...): every header that goes through addHeader -- the entry
    // itself, each existing ancestor re-added on the way, each directory created
    std::function<void(const std::string& dst, int64_t ref)> on_add;
    // the header of a directory addAncestors creates (createHeader from lastAncestor's FileInfo, ModTime = now, the
    // given uid/gid, :551-559) -> its payload
    std::function<int64_t(const std::string& dst, const Node& last_ancestor, uint32_t uid, uint32_t gid)> make_dir;

    static std::vector<std::string> parts(const std::string& p) {               // pathutils.SplitPath
        std::vector<std::string> out;
        size_t i = 0;
        while (i < p.size()) {
            while (i < p.size() && p[i] == '/') ++i;
            size_t j = i;
            while (j < p.size() && p[j] != '/') ++j;
            if (j > i) out.push_back(p.substr(i, j - i));
            i = j;
        }
        return out;
    }
    static std::string join_abs(const std::vector<std::string>& ps, size_t n) {  // AbsPath(filepath.Join(parts[:n]...))
        std::string q;
        for (size_t k = 0; k < n; ++k) q += "/" + ps[k];
        return mi_walk::abs_path(q);
    }
    // Lookups arrive in walk order -- a directory, then its contents: the node of the last path's PARENT is kept, and a
    // path below the same parent costs one lookup in that node's children instead of a walk from the root through maps
    // that a million nodes have pushed out of every cache (a scan of 10^6 unchanged entries: 2.9 -> 0.36 us per entry, profiles/r04_host_scale.txt).
    // The kept node is dropped when the tree changes AT OR ABOVE its path (a node replaced, erased or emptied there);
    // a change elsewhere -- a leaf put below it, above all -- leaves it standing.  `gen` counts every change of shape.
    uint64_t gen = 1;
    // While a content-aware scan's walk runs with MI_MEMFS_TRUST_CTIME its directory readers -- several threads -- look paths up
    // in this tree and mark nodes (Fs::content_is_known): nothing may change its shape meanwhile.  The commit holds the tree
    // `frozen` for that time; a change of shape while it is (a mutation somebody adds beside the walk some day) is a defect that
    // would leave those threads with dangling nodes -- it stops the process with a message instead.
    bool frozen = false;
    void changing() {
        if (frozen) {
            fprintf(stderr, "makisu_mi: the MemFS tree was changed while a scan's walk was reading it from its directory readers (mi_memtree.h: frozen)\n");
            abort();
        }
        ++gen;
    }
    // One kept parent PER DEPTH: a walk descends and comes back (a/b, a/b/x, a/b/y, a/c), and the directory it comes back to is
    // still kept at its depth -- with a PLACE among its children: siblings arrive in name order, the order of the children
    // map, so the next lookup below the same parent is the next child (or a few steps on), not a search among thousands of
    // siblings whose map nodes no cache holds.  The place is an iterator: any change of shape forgets every place (an erased
    // child may be the one it points at); the kept nodes follow the rule above.
    using Kids = decltype(Node::children);
    struct Kept { std::string dir; Node* node = nullptr; Kids::iterator at; bool placed = false; };
    std::vector<Kept> kept;                                                     // [depth of dir] ("/a/b": 2)
    void shape_changed(const std::string& at) {
        changing();
        for (Kept& k : kept) {
            k.placed = false;
            if (k.node && k.dir.size() >= at.size() && memcmp(k.dir.data(), at.data(), at.size()) == 0 &&
                (k.dir.size() == at.size() || k.dir[at.size()] == '/' || at == "/"))
                k.node = nullptr;
        }
    }
    void shape_reset() { changing(); kept.clear(); }
    // where a clean absolute path splits into parent and name; npos = take the general way
    static size_t parent_cut(const std::string& p) {
        const size_t cut = p.find_last_of('/');
        return (cut == std::string::npos || cut == 0 || cut + 1 >= p.size() || p[0] != '/') ? std::string::npos : cut;
    }
    Kept* kept_parent(const std::string& p, size_t cut) {                       // the node of p[0, cut), kept for the next call
        size_t depth = 0;
        for (size_t i = 0; i < cut; ++i) depth += p[i] == '/';
        if (depth >= 64) depth = 63;                                            // (deeper directories share the last slot)
        if (kept.size() <= depth) kept.resize(depth + 1);
        Kept& k = kept[depth];
        if (k.node && k.dir.size() == cut && memcmp(k.dir.data(), p.data(), cut) == 0) return &k;
        k.dir.assign(p, 0, cut);
        k.node = find_walk(k.dir);
        k.placed = false;
        return k.node ? &k : nullptr;
    }
    Node* parent_node(const std::string& p, size_t cut) {
        Kept* k = kept_parent(p, cut);
        return k ? k->node : nullptr;
    }
    Node* find(const std::string& p) {                                          // isUpdated's walk; nullptr = "new"
        const size_t cut = parent_cut(p);
        if (cut == std::string::npos) return find_walk(p);
        Kept* k = kept_parent(p, cut);
        if (!k) return nullptr;
        Kids& kids = k->node->children;
        const std::string_view name(p.data() + cut + 1, p.size() - cut - 1);
        if (k->placed) {
            for (int steps = 0; k->at != kids.end() && steps < 4 && std::string_view(k->at->first) < name; ++steps) ++k->at;
            if (k->at == kids.end()) {                                          // past the last child: a name behind it is new
                if (kids.empty() || std::string_view(std::prev(kids.end())->first) < name) return nullptr;
                k->at = kids.lower_bound(name);
            } else if (std::string_view(k->at->first) != name) {
                k->at = kids.lower_bound(name);
            }
        } else {
            k->at = kids.lower_bound(name);
            k->placed = true;
        }
        if (k->at == kids.end() || std::string_view(k->at->first) != name) return nullptr;
        Node* found = k->at->second.get();
        ++k->at;                                                                // the next sibling: where the next lookup begins
        return found;
    }
    Node* find_walk(const std::string& p) {
        Node* cur = &root;
        size_t i = 0;
        while (i < p.size()) {                                                  // SplitPath's parts, without the vector
            while (i < p.size() && p[i] == '/') ++i;
            size_t j = i;
            while (j < p.size() && p[j] != '/') ++j;
            if (j > i) {
                auto it = cur->children.find(std::string_view(p.data() + i, j - i));
                if (it == cur->children.end()) return nullptr;
                cur = it->second.get();
            }
            i = j;
        }
        return cur;
    }
    // a listed tree: the node at its path, parents that are not listed created on the way (no payload)
    void load(const std::string& p, int64_t ref, uint8_t kind, const char* link) {
        shape_changed(p);
        Node* cur = &root;
        for (const std::string& part : parts(p)) {
            std::unique_ptr<Node>& slot = cur->children[part];
            if (!slot) slot.reset(new Node);
            cur = slot.get();
        }
        cur->ref = ref; cur->kind = kind; cur->link = link ? link : "";
    }
    // contentMemFile.updateMemFS: the node at dst is replaced; the new node takes over the old node's children iff
    // the NEW header is a directory; a missing part before the last one is an error
    bool put(const std::string& dst, int64_t ref, uint8_t kind, const std::string& link) {
        if (on_add) on_add(dst, ref);
        const size_t cut = parent_cut(dst);
        if (cut != std::string::npos) {                                         // the parent by the kept node: the last part only
            if (Node* parent = parent_node(dst, cut)) {
                const std::string_view name(dst.data() + cut + 1, dst.size() - cut - 1);
                std::unique_ptr<Node> nn(new Node);
                nn->ref = ref; nn->kind = kind; nn->link = link;
                Kids& kids = parent->children;                                  // (names come sorted: mostly behind the last one)
                auto it = kids.empty() || std::string_view(std::prev(kids.end())->first) < name ? kids.end() : kids.lower_bound(name);
                if (it != parent->children.end() && it->first == name) {
                    if (kind == 0) nn->children = std::move(it->second->children);
                    shape_changed(dst);                                         // (before the old node goes)
                    it->second = std::move(nn);
                } else {
                    changing();                                                 // a new leaf: nobody's kept parent
                    parent->children.emplace_hint(it, std::string(name), std::move(nn));   // (names come sorted: at the end)
                }
                return true;
            }
        }                                                                       // (a missing parent: the walk names the part)
        shape_changed(dst);
        const std::vector<std::string> ps = parts(dst);
        Node* cur = &root;
        for (size_t i = 0; i < ps.size(); ++i) {
            auto it = cur->children.find(ps[i]);
            const bool last = i + 1 == ps.size();
            if (it != cur->children.end() && !last) { cur = it->second.get(); continue; }
            if (it == cur->children.end() && !last) {
                err = "missing intermediate directory " + ps[i] + " in " + dst;
                return false;
            }
            std::unique_ptr<Node> nn(new Node);
            nn->ref = ref; nn->kind = kind; nn->link = link;
            if (it != cur->children.end()) {
                if (kind == 0) nn->children = std::move(it->second->children);
                it->second = std::move(nn);
            } else {
                cur->children[ps[i]] = std::move(nn);
            }
        }
        return true;
    }
    // whiteoutMemFile.updateMemFS
    bool wipe(const std::string& del) {
        shape_changed(del);
        const std::vector<std::string> ps = parts(del);
        Node* cur = &root;
        for (size_t i = 0; i < ps.size(); ++i) {
            auto it = cur->children.find(ps[i]);
            const bool last = i + 1 == ps.size();
            if (it != cur->children.end()) {
                if (last) cur->children.erase(it);
                else cur = it->second.get();
            } else if (!last) {
                err = "missing intermediate dir " + ps[i] + " in " + del;
                return false;
            }                                                                   // else "Trying to whiteout nonexistent path"
        }
        return true;
    }
    // l.addHeader(src, dst, hdr).updateMemFS(tree) (mem_layer.go:197-212): a ".wh.<name>" base name is a whiteout of
    // its sibling <name>, filed under THAT path; anything else is content
    bool add(const std::string& dst, int64_t ref, uint8_t kind, const std::string& link) {
        const size_t cut = dst.find_last_of('/');
        if (dst.compare(cut == std::string::npos ? 0 : cut + 1, 4, ".wh.") != 0) return put(dst, ref, kind, link);
        const std::string name = mi_walk::base_of(dst);
        if (on_add) on_add(dst, ref);
        const std::string dir = mi_walk::dir_of(dst);
        return wipe((dir == "/" ? "" : dir) + "/" + name.substr(4));
    }
    // addAncestors.  Re-adding an existing ancestor "as it is" through updateMemFS changes nothing for a directory and
    // drops the children of anything else; a symlink sends the walk to its target (filepath.Join(linkname, the
    // remaining parts), from the tree's root) and ends it; any other non-directory lets the walk go on one part further
    // WITHOUT descending (the switch at :535-549 has no case for it); what is then still missing of dst's own prefix
    // is created as directories.  resolved = "the resolved dst path to the best of its knowledge".
    bool chain_plain = true;                   // the last add_ancestors met directories only (its chain = dst's own prefixes)
    bool add_ancestors(const std::string& dst, bool inclusive, int depth, uint32_t uid, uint32_t gid,
                       std::string* resolved) {
        if (depth >= 1024) {                       // (by now dst is the link's target joined to itself a thousand times)
            err = "symlink loop at " + (dst.size() > 160 ? dst.substr(0, 160) + "..." : dst);
            return false;
        }
        if (depth == 0) chain_plain = true;
        const std::vector<std::string> ps = parts(dst);
        const size_t end = inclusive ? ps.size() : (ps.empty() ? 0 : ps.size() - 1);
        Node* cur = &root;
        const Node* last_ancestor = &root;
        std::string cur_path;                                                   // "" = the root
        size_t i = 0;
        for (; i < end; ++i) {
            auto it = cur->children.find(ps[i]);
            if (it == cur->children.end()) break;
            Node* n = it->second.get();
            const std::string n_path = cur_path + "/" + ps[i];
            if (on_add) on_add(n_path, n->ref);
            if (n->kind == 0) { last_ancestor = n; cur = n; cur_path = n_path; continue; }
            chain_plain = false;
            if (!n->children.empty()) { n->children.clear(); shape_changed(n_path); }
            if (n->kind == 2) {
                std::string target = n->link;
                for (size_t k = i + 1; k < ps.size(); ++k) target += "/" + ps[k];
                target = mi_walk::clean_any(target);
                if (!add_ancestors(target, inclusive, depth + 1, uid, gid, resolved)) {
                    // (the reference wraps the error once per level; the outermost wrap is the one that says where)
                    if (depth == 0) err = "get symlink target ancestors " + target + ": " + err;
                    return false;
                }
                return true;
            }
        }
        for (size_t j = i; j < end; ++j) {
            const std::string q = join_abs(ps, j + 1);
            const int64_t ref = make_dir ? make_dir(q, *last_ancestor, uid, gid) : -1;
            if (!put(q, ref, 0, std::string())) { err = "update memfs with ancestor " + q + ": " + err; return false; }
        }
        if (resolved) *resolved = dst;
        return true;
    }
};

}  // namespace mi_memtree
