"""A python model of MemFS.UpdateFromTarReader (lib/snapshot/mem_fs.go:165-255) on a NESTED tree, the reference's own
shape (memFSNode: header + children map), and unit cases of the layer merge:

  * every header except hard links goes through maybeAddToLayer in stream order, the hard links after them (:214-236);
  * maybeAddToLayer (:440-458): isUpdated walks the children maps part by part (:487-503) -- a missing part means
    "new", otherwise tario.IsSimilarHeader decides; "/" is never added; a changed path first gets its ancestors
    (addAncestors :505-566: existing ones are re-added as they are, MISSING ones are created), then
    memLayer.addHeader(...).updateMemFS (mem_layer.go:50-76, 104-125): a ".wh.<name>" base name deletes the sibling
    <name>, anything else replaces the node and takes over the old node's children iff the NEW header is a directory.

The library has ONE implementation of this merge since round 4 -- the MemFS handle (mi_memfs_update_from_entries;
the stateless mi_entries_apply_layer, which could not name the directories addAncestors creates, is gone).  The
differential test of the handle against this model over generated layer sequences -- entries without parents, files
and symlinks that get children (and lose them when they are re-added as somebody's ancestor), directories arriving on
files, whiteouts of what is not there, "./" entries, the same path twice, the reference's two failures with its own
words -- is tests/test_host_memfs.py::test_update_from_entries_equals_the_model_made_up_directories_included; the
cases below go through makisu_amd.apply_layer, a harness helper that merges two entry lists on a fresh handle."""
import posixpath

import pytest

from hypothesis import strategies as st

import makisu_amd as M

NAMES = ["a", "b", "c"]
TYPE = {M.KIND_DIR: 0o40000, M.KIND_FILE: 0o100000, M.KIND_SYMLINK: 0o120000, M.KIND_HARDLINK: 0o100000}


def _abs(p):                                                  # pathutils.AbsPath (lib/pathutils/path.go:41-43)
    out = []
    for el in p.split("/"):
        if el == "..":
            out and out.pop()
        elif el not in ("", "."):
            out.append(el)
    return "/" + "/".join(out)


class ReferenceFails(Exception):
    """UpdateFromTarReader returns an error: "add hdr from tar to layer: ..." (mem_fs.go:225, 237)."""


def _similar(a, b):
    """tario.IsSimilarHeader with ignoreTime false (lib/tario/compare.go:24-120), hard-link targets as absolute paths."""
    if a["kind"] != b["kind"]:
        return False
    if a["kind"] == M.KIND_SYMLINK:
        return a["link_target"] == b["link_target"]
    same = a["mtime_sec"] == b["mtime_sec"] and a["uid"] == b["uid"] and a["gid"] == b["gid"] and \
        (a["mode"] & 0o7777) == (b["mode"] & 0o7777)
    if a["kind"] == M.KIND_HARDLINK:
        return same and _abs(a["link_target"]) == _abs(b["link_target"])
    if a["kind"] == M.KIND_FILE:
        return same and a["size"] == b["size"]
    return same


class Node:
    def __init__(self, hdr, dst, made_up=False):
        self.hdr, self.dst, self.children, self.made_up = hdr, dst, {}, made_up


def model_update_from_tar(tree, layer):
    def parts(p):                                             # pathutils.SplitPath
        t = p.strip("/")
        return t.split("/") if t else []

    def is_updated(p, hdr):
        cur = tree
        for part in parts(p):
            if part not in cur.children:
                return True
            cur = cur.children[part]
        return not _similar(cur.hdr, hdr)

    def put(node):                                            # contentMemFile.updateMemFS (mem_layer.go:50-76)
        cur, ps = tree, parts(node.dst)
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

    def delete(p):                                            # whiteoutMemFile.updateMemFS (mem_layer.go:104-125)
        cur, ps = tree, parts(p)
        for i, part in enumerate(ps):
            if part in cur.children:
                if i == len(ps) - 1:
                    del cur.children[part]
                else:
                    cur = cur.children[part]
            elif i != len(ps) - 1:
                raise ReferenceFails("missing intermediate dir %s in %s" % (part, p))
            # else "Trying to whiteout nonexistent path"

    def add_ancestors(dst, depth=0):                          # mem_fs.go:505-566, line by line
        if depth >= 1024:
            raise ReferenceFails("symlink loop at " + dst)
        cur, ps = tree, parts(dst)
        end, i = len(ps) - 1, 0
        while i < end:
            n = cur.children.get(ps[i])
            if n is None:
                break
            put(Node(n.hdr, n.dst, n.made_up))                # re-added as it is, under ITS OWN path
            if n.hdr["kind"] == M.KIND_DIR:
                cur = n
            elif n.hdr["kind"] == M.KIND_SYMLINK:             # "Add ancestors of symlink target too", then return
                add_ancestors(posixpath.join(n.hdr["link_target"], *ps[i + 1:]), depth + 1)
                return
            # any other type: the walk goes on one part further WITHOUT descending (curr is unchanged, :535-549)
            i += 1
        for j in range(i, end):                               # "Create missing intermediate dir for unresolved part"
            q = _abs("/".join(ps[:j + 1]))
            put(Node({"kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 1 << 40, "uid": 0, "gid": 0, "size": 0,
                      "link_target": None, "relpath": q.lstrip("/")}, q, made_up=True))

    def maybe_add(e):
        dst = _abs(e["relpath"])
        if not is_updated(dst, e) or dst == "/":
            return
        add_ancestors(dst)
        d, b = posixpath.split(dst)
        if b.startswith(".wh."):
            delete(posixpath.join(d, b[4:]))
        else:
            put(Node(e, dst))

    links = {}
    for e in layer:
        if e["kind"] == M.KIND_HARDLINK:
            links[_abs(e["relpath"])] = e                     # a map keyed by path: the last header of a path wins
        else:
            maybe_add(e)
    for p in sorted(links):                                   # Go ranges over the map in NO particular order; the library
        maybe_add(links[p])                                   # takes the paths sorted, one of the orders Go may take


def flatten(tree):
    out = {}

    def walk(n, p):
        for name, c in n.children.items():
            q = p.rstrip("/") + "/" + name
            if not c.made_up:
                out[q] = c.hdr
            walk(c, q)
    walk(tree, "/")
    return out


@st.composite
def _entry(draw, allow_marker=True):
    depth = draw(st.integers(1, 3))
    ps = [draw(st.sampled_from(NAMES)) for _ in range(depth)]
    kind = draw(st.sampled_from([M.KIND_DIR, M.KIND_DIR, M.KIND_FILE, M.KIND_FILE, M.KIND_SYMLINK, M.KIND_HARDLINK]))
    if allow_marker and draw(st.integers(0, 6)) == 0:
        ps[-1] = ".wh." + ps[-1]
        kind = M.KIND_FILE
    rel = "/".join(ps)
    if draw(st.integers(0, 9)) == 0:
        rel = draw(st.sampled_from(["./", ".", "/"]))
        kind = M.KIND_DIR
    rel = draw(st.sampled_from(["", "/", "./"])) + rel if not rel.startswith((".", "/")) else rel
    if kind == M.KIND_DIR and draw(st.booleans()):
        rel += "/"
    e = {"relpath": rel, "kind": kind, "mode": TYPE[kind] | draw(st.sampled_from([0o755, 0o700])),
         "mtime_sec": draw(st.sampled_from([5, 6])), "uid": draw(st.sampled_from([0, 1000])), "gid": 0,
         "size": draw(st.sampled_from([0, 10])) if kind == M.KIND_FILE else 0, "link_target": None}
    if kind == M.KIND_SYMLINK:
        e["link_target"] = draw(st.sampled_from(["/a", "/b/c", "/c", "b"]))
    elif kind == M.KIND_HARDLINK:
        e["link_target"] = draw(st.sampled_from(["a/b", "/a/b", "c"]))
    return e


def test_the_root_of_a_layer_never_replaces_the_root():
    """mem_fs.go:447: "Root itself is not added to layers" -- a layer's "./" header changes nothing."""
    D = lambda p, **kw: dict({"relpath": p, "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 100, "size": 0}, **kw)   # noqa: E731
    base = [D(""), D("a")]                                      # (the root is the handle's own node, not an entry of the tree)
    for name in ("./", ".", "/", ""):
        out = M.apply_layer(base, [D(name, mode=0o40700, mtime_sec=7), D("b")])
        assert [(e["relpath"], e["mode"]) for e in out] == [("a", 0o40755), ("b", 0o40755)]
    assert [e["relpath"] for e in M.apply_layer([], [D("./"), D("b")])] == ["b"]


def test_a_directory_takes_over_the_children_whatever_it_replaces():
    """mem_layer.go:59-64: the children are copied iff the NEW header is a directory; a non-directory drops them, also
    when the path itself was never listed."""
    D = lambda p: {"relpath": p, "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 100, "size": 0}                   # noqa: E731
    F = lambda p: {"relpath": p, "kind": M.KIND_FILE, "mode": 0o100644, "mtime_sec": 100, "size": 3}                 # noqa: E731
    names = lambda x: [e["relpath"] for e in x]                                                                      # noqa: E731
    fs = M.apply_layer([], [F("a"), F("a/x")])                       # a file that got a child (isUpdated walks any node)
    assert names(fs) == ["a", "a/x"]
    assert names(M.apply_layer(fs, [D("a")])) == ["a", "a/x"]        # the directory keeps it
    assert names(M.apply_layer(fs, [dict(F("a"), size=4)])) == ["a"]  # another file does not
    fs = M.apply_layer([], [F("p/q/r")])                             # p and p/q were never listed: addAncestors made them
    assert names(fs) == ["p", "p/q", "p/q/r"] and [e["kind"] for e in fs] == [M.KIND_DIR, M.KIND_DIR, M.KIND_FILE]
    assert names(M.apply_layer(fs, [F("p/q")])) == ["p", "p/q"]
    assert names(M.apply_layer(fs, [D("p/q")])) == ["p", "p/q", "p/q/r"]


def test_entries_below_a_symlink_the_way_the_reference_treats_them():
    """addAncestors re-adds every existing ancestor through updateMemFS (mem_fs.go:531-534): a directory comes back with
    its children, a symlink (or file) WITHOUT them -- so below a symlink only the latest entry survives; one level
    deeper updateMemFS finds no node to descend into and the build fails (mem_layer.go:70-72); a symlink that leads
    back to itself ends at depth 1024 (mem_fs.go:506-508)."""
    D = lambda p: {"relpath": p, "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 100, "size": 0}                   # noqa: E731
    F = lambda p: {"relpath": p, "kind": M.KIND_FILE, "mode": 0o100644, "mtime_sec": 100, "size": 3}                 # noqa: E731
    L = lambda p, t: {"relpath": p, "kind": M.KIND_SYMLINK, "mode": 0o120777, "mtime_sec": 100, "size": 0,           # noqa: E731
                      "link_target": t}
    names = lambda x: [e["relpath"] for e in x]                                                                      # noqa: E731
    base = M.apply_layer([], [D("usr"), D("usr/lib"), L("lib", "/usr/lib")])
    assert names(M.apply_layer(base, [F("lib/x")])) == ["lib", "lib/x", "usr", "usr/lib"]
    assert names(M.apply_layer(base, [F("lib/x"), F("lib/y")])) == ["lib", "lib/y", "usr", "usr/lib"]
    with pytest.raises(M.MiError) as ei:
        M.apply_layer(base, [F("lib/d/z")])
    assert "add hdr from tar to layer: update memfs with file /lib/d/z: missing intermediate directory d in /lib/d/z" \
        in str(ei.value)
    with pytest.raises(M.MiError) as ei:
        M.apply_layer(base, [L("loop", "/loop"), F("loop/x")])
    assert "add ancestors of /loop/x: " in str(ei.value) and "symlink loop at /loop/x" in str(ei.value)
    # the target's ancestors are created (entries of the tree like any other), the entry itself stays where it is
    assert names(M.apply_layer([], [L("l", "/t/u"), F("l/f")])) == ["l", "l/f", "t", "t/u"]
