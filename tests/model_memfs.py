"""A line-by-line python model of the reference's MemFS layer building, for differential tests (test infrastructure):
memFSNode / isUpdated / addAncestors / both updateMemFS / addHeader / addToLayer (lib/snapshot/mem_fs.go:33-57, 343-421,
440-566; lib/snapshot/mem_layer.go:50-125, 197-228) on entry dicts.  Written from the Go source, statement by statement --
including what the statements do in corners nobody meant (a non-directory ancestor loses its children when it is re-added;
the switch in addAncestors that does not descend into it; the resolved path that only the createDst branch uses)."""
import posixpath

import makisu_amd as M


class ReferenceFails(Exception):
    """the reference returns an error here (the build stops)"""


def abs_path(p):                                              # pathutils.AbsPath (lib/pathutils/path.go:41-43)
    out = []
    for el in p.split("/"):
        if el == "..":
            out and out.pop()
        elif el not in ("", "."):
            out.append(el)
    return "/" + "/".join(out)


def join(*ps):                                                # filepath.Join: Clean of the joined non-empty elements
    s = "/".join(p for p in ps if p)
    if not s:
        return ""
    rooted = s.startswith("/")
    out = []
    for el in s.split("/"):
        if el == "..":
            if out and out[-1] != "..":
                out.pop()
            elif not rooted:
                out.append(el)
        elif el not in ("", "."):
            out.append(el)
    return ("/" if rooted else "") + "/".join(out) or ("/" if rooted else ".")


def similar(a, b):
    """tario.IsSimilarHeader, ignoreTime false (lib/tario/compare.go:24-120); hard-link targets as absolute paths"""
    if a["kind"] != b["kind"]:
        return False
    if a["kind"] == M.KIND_SYMLINK:
        return a.get("link_target") == b.get("link_target")
    same = a["mtime_sec"] == b["mtime_sec"] and a.get("uid", 0) == b.get("uid", 0) and a.get("gid", 0) == b.get("gid", 0) and \
        (a["mode"] & 0o7777) == (b["mode"] & 0o7777)
    if a["kind"] == M.KIND_HARDLINK:
        return same and abs_path(a["link_target"]) == abs_path(b["link_target"])
    if a["kind"] == M.KIND_FILE:
        return same and a["size"] == b["size"]
    return same


class Node:
    def __init__(self, hdr, dst, src="", made_up=False):
        self.hdr, self.dst, self.src, self.children, self.made_up = hdr, dst, src, {}, made_up


def parts(p):                                                 # pathutils.SplitPath
    t = p.strip("/")
    return t.split("/") if t else []


class ModelFS:
    NOW = 1 << 40                                             # the clock: mtime of created directories

    def __init__(self, root_hdr=None):
        self.tree = Node(root_hdr or {"kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 1, "uid": 0, "gid": 0, "size": 0,
                                      "link_target": None, "relpath": ""}, "/")
        self.layer = {}                                       # memLayer.files: key -> ("content", node) | ("whiteout", path)

    # -- memLayer.addHeader(...).updateMemFS(tree) (mem_layer.go:197-212, 50-76, 104-125)
    def add_header(self, node):
        d, b = posixpath.split(node.dst)
        if b.startswith(".wh."):
            deleted = posixpath.join(d, b[4:])
            self.layer[deleted] = ("whiteout", node.dst)
            self._delete(deleted)
        else:
            self.layer[node.dst] = ("content", node)
            self._put(node)

    def _put(self, node):
        cur, ps = self.tree, parts(node.dst)
        for i, part in enumerate(ps):
            last = i == len(ps) - 1
            if part in cur.children:
                if last:
                    old = cur.children[part]
                    cur.children[part] = node
                    if node.hdr["kind"] == M.KIND_DIR:
                        node.children.update(old.children)
                else:
                    cur = cur.children[part]
            elif last:
                cur.children[part] = node
            else:
                raise ReferenceFails("missing intermediate directory %s in %s" % (part, node.dst))

    def _delete(self, p):
        cur, ps = self.tree, parts(p)
        for i, part in enumerate(ps):
            if part in cur.children:
                if i == len(ps) - 1:
                    del cur.children[part]
                else:
                    cur = cur.children[part]
            elif i != len(ps) - 1:
                raise ReferenceFails("missing intermediate dir %s in %s" % (part, p))

    def is_updated(self, p, hdr):                             # (:487-503)
        cur = self.tree
        for part in parts(p):
            if part not in cur.children:
                return True
            cur = cur.children[part]
        return not similar(cur.hdr, hdr)

    def add_ancestors(self, dst, inclusive, depth=0, uid=0, gid=0):   # (:505-566) -> the resolved dst
        if depth >= 1024:
            raise ReferenceFails("symlink loop at " + dst)
        last_ancestor = self.tree
        cur, ps = self.tree, parts(dst)
        end = len(ps) if inclusive else len(ps) - 1
        i = 0
        while i < end:
            n = cur.children.get(ps[i])
            if n is None:
                break
            self.add_header(Node(n.hdr, n.dst, n.src, n.made_up))     # l.addHeader(n.src, n.dst, n.hdr).updateMemFS(fs.tree)
            if n.hdr["kind"] == M.KIND_DIR:
                last_ancestor = n
                cur = n
            elif n.hdr["kind"] == M.KIND_SYMLINK:
                remaining = join(*ps[i + 1:])
                target = join(n.hdr["link_target"], remaining)
                return self.add_ancestors(target, inclusive, depth + 1, uid, gid)
            i += 1
        for j in range(i, end):
            q = abs_path(join(*ps[:j + 1]))
            hdr = {"kind": M.KIND_DIR, "mode": 0o40000 | (last_ancestor.hdr["mode"] & 0o7777), "mtime_sec": self.NOW, "uid": uid,
                   "gid": gid, "size": 0, "link_target": None, "relpath": q.lstrip("/")}
            self.add_header(Node(hdr, q, "/", made_up=True))          # addHeader("", curr, hdr): src = AbsPath("")
        return dst

    def maybe_add(self, src, dst, hdr):                       # maybeAddToLayer, createWhiteout = false (:440-458)
        if self.is_updated(dst, hdr) and dst != "/":
            self.add_ancestors(abs_path(dst), False)
            self.add_header(Node(hdr, abs_path(dst), abs_path(src)))

    # -- UpdateFromTarReader, untar = false (:165-255).  src_of(p): the reference passes AbsPath(hdr.Name), the path the
    #    entry is untarred to under the root "/"; the library joins it to ITS root (include/makisu_mi.h says so)
    def update_from_tar(self, layer, src_of=lambda p: p):
        self.layer = {}
        links = {}
        for e in layer:
            p = abs_path(e["relpath"])
            if e["kind"] == M.KIND_HARDLINK:
                links[p] = e
            else:
                self.maybe_add(src_of(p), p, e)
        for p in sorted(links):                               # (Go ranges over the map in no particular order)
            self.maybe_add(src_of(p), p, links[p])

    # -- createLayerByScan (:315-341): walked = [(src, dst, hdr)] in filepath.Walk order, the root first
    def scan(self, walked, on_disk):
        self.layer = {}
        for src, dst, hdr in walked:
            n = self.tree                                     # isUpdated -> (updated, node)
            for part in parts(dst):
                n = n.children.get(part) if n is not None else None
            updated = n is None or not similar(n.hdr, hdr)
            if updated and dst != "/":
                self.add_ancestors(abs_path(dst), False)
                self.add_header(Node(hdr, abs_path(dst), abs_path(src)))
            if hdr["kind"] == M.KIND_DIR and n is not None:   # "Only one whiteout file is needed for a deleted subtree."
                for child in list(n.children.values()):
                    if not on_disk(child.src):
                        d, b = posixpath.split(child.dst)
                        if b.startswith(".wh."):
                            raise ReferenceFails("base name contains whiteout prefix: " + child.dst)
                        self.layer[child.dst] = ("whiteout", posixpath.join(d, ".wh." + b))
                        self._delete(child.dst)
                        self.add_ancestors(child.dst, False)

    # -- addToLayer (:343-421); walk_entries(src) -> the snapshot walk of a source: [(currSrc, entry dict)], src itself first
    def add_to_layer(self, op, walk_entries, is_dir):
        dst = op["dst"]
        create_dst = True
        if len(op["srcs"]) == 1 and not is_dir(join(op["src_root"], op["srcs"][0])):
            create_dst = False
        if create_dst:
            resolved = self.add_ancestors(abs_path(dst), True, 0, op["uid"], op["gid"])
            if not resolved.endswith("/"):
                resolved += "/"
            dst = resolved
        for s in op["srcs"]:
            src = join(op["src_root"], s)
            for curr_src, e in walk_entries(src):
                if curr_src == src:
                    if e["kind"] == M.KIND_DIR:
                        continue
                    curr_dst = dst if not dst.endswith("/") else join(dst, posixpath.basename(src))
                else:
                    curr_dst = join(dst, curr_src[len(src):])
                hdr = dict(e, uid=op["uid"], gid=op["gid"], relpath=abs_path(curr_dst).lstrip("/"))
                self.maybe_add(curr_src, curr_dst, hdr)

    def flat(self, made_up=True):
        out = {}

        def walk(n, p):
            for name, c in n.children.items():
                q = p.rstrip("/") + "/" + name
                if made_up or not c.made_up:
                    out[q] = c
                walk(c, q)
        walk(self.tree, "/")
        return out
