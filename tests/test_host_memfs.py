"""CPU tests of mi_memfs_*: the reference's MemFS (lib/snapshot/mem_fs.go) as a handle -- one tree for the life of a build.

Replayed: TestAddLayerByScanWhiteout (mem_fs_test.go:1038-1116), TestCreateLayerByScan's Simple / Symlink / Whiteout
(:572-686), TestUpdateMemFS (:164-344), TestAddLayersEqual's intent (:1118-1196), a FROM + COPY + RUN sequence the way
build_stage.go drives MemFS (UpdateFromTar, AddLayerByCopyOps, AddLayerByScan) -- and, on generated sequences, the handle
against the stateless calls it generalises (mi_snapshot_diff, mi_entries_apply_layer) and against the line-by-line model
of tests/test_host_apply_properties.py, made-up directories included."""
import os
import shutil

import pytest
from hypothesis import given, settings, strategies as st

import makisu_amd as M
from test_host_apply_properties import Node, _abs, _entry, flatten, model_update_from_tar, ReferenceFails


def _mk(root, spec):
    for p, kind, content in spec:
        full = os.path.join(root, p.lstrip("/"))
        if kind == "d":
            os.makedirs(full, exist_ok=True)
        elif kind == "f":
            os.makedirs(os.path.dirname(full), exist_ok=True)
            with open(full, "w") as f:
                f.write(content)
            os.chmod(full, 0o755)
        else:
            os.makedirs(os.path.dirname(full), exist_ok=True)
            os.symlink(content, full)


def _names(layer):
    return ["/" + e["relpath"] for e in layer]


def test_add_layer_by_scan_whiteout_replayed(tmp_path):
    """six entries under /test1 -> a layer of 6; RemoveAll(/test1) -> a layer of ONE entry, the whiteout of /test1"""
    root = str(tmp_path)
    _mk(root, [("/test1", "d", ""), ("/test1/test2", "d", ""), ("/test1/test2/test3.txt", "f", "hello"), ("/test1/test4", "d", ""),
               ("/test1/test4/test5", "d", ""), ("/test1/test4/test5/test6.txt", "f", "hello")])
    with M.MemFS(root) as fs:
        l1 = fs.scan()
        assert len(l1) == 6 and _names(l1) == sorted(_names(l1))
        assert [e["src"] for e in l1] == [root + p for p in _names(l1)]
        assert fs.scan() == []                                  # nothing changed: an empty layer
        shutil.rmtree(os.path.join(root, "test1"))
        l2 = fs.scan()
        assert _names(l2) == ["/.wh.test1"] and l2[0]["src"] == "" and l2[0]["file_index"] == -1
        assert fs.entries() == [] and fs.scan() == []


def test_the_walks_word_is_taken_for_a_node_made_during_the_scan(tmp_path):
    """isOnDisk is an lstat of the node's source; a path this scan's WALK lists has just been lstat'ed, so the walk's
    word is taken instead.  For the nodes the tree held when the scan began that word is a mark on the node; for a node
    made DURING the scan it is the set of the walk's paths.  A walk handed over out of order -- a new file before its
    directory -- makes the directory's deletion check meet such a node: listed by the walk (and, here, not on disk at
    all: the entries are made up), it gets no whiteout."""
    root = str(tmp_path)
    D = lambda p: {"relpath": p, "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 100, "size": 0}        # noqa: E731
    F = lambda p, t=100: {"relpath": p, "kind": M.KIND_FILE, "mode": 0o100644, "mtime_sec": t, "size": 5}   # noqa: E731
    with M.MemFS(root) as fs:
        assert _names(fs.add_layer_by_scan([D("d"), F("d/old")])) == ["/d", "/d/old"]
        # second scan: d/new comes BEFORE d; d/old is listed too (a marked node), d/gone never existed
        layer = fs.add_layer_by_scan([F("d/new"), D("d"), F("d/old")])
        assert _names(layer) == ["/d", "/d/new"]                 # d as the new file's ancestor; no whiteout of d/new
        assert [e["relpath"] for e in fs.entries()] == ["d", "d/new", "d/old"]
        # and a path the walk does NOT list, whose source is not on disk, is whited out (the same check, the other way)
        layer = fs.add_layer_by_scan([D("d"), F("d/new")])
        assert _names(layer) == ["/d", "/d/.wh.old"]


def test_create_layer_by_scan_replayed(tmp_path):
    """Simple: new paths with their (new) directories; Symlink: the target as written; Whiteout: a removed file in a
    directory that stays -> the directory is carried as an ancestor beside the whiteout"""
    root = str(tmp_path)
    _mk(root, [("/test1/test2/test3.txt", "f", "hello"), ("/test1/link", "l", "test2/test3.txt"),
               ("/test1/abs", "l", os.path.join(root, "test1/test2"))])
    with M.MemFS(root) as fs:
        l1 = fs.scan()
        by = {"/" + e["relpath"]: e for e in l1}
        assert sorted(by) == ["/test1", "/test1/abs", "/test1/link", "/test1/test2", "/test1/test2/test3.txt"]
        assert by["/test1/link"]["link_target"] == "test2/test3.txt"
        assert by["/test1/abs"]["link_target"] == "/test1/test2"          # createHeader trims the root (mem_layer.go:176-184)
        os.unlink(os.path.join(root, "test1/test2/test3.txt"))
        l2 = fs.scan()
        # the directory's mtime changed with the unlink: it is in the layer as a changed entry, its parent as an ancestor
        assert _names(l2) == ["/test1", "/test1/test2", "/test1/test2/.wh.test3.txt"]
        assert [e["relpath"] for e in fs.entries()] == ["test1", "test1/abs", "test1/link", "test1/test2"]


def test_update_mem_fs_through_the_real_path():
    """TestUpdateMemFS (mem_fs_test.go:164-344) through UpdateFromTarReader -> maybeAddToLayer: where the test-only
    MemFS.merge fails with "missing intermediate directory" (SkipDirCausesError), the real path CREATES the directory --
    mode of the nearest ancestor, mtime = the clock, uid/gid 0 (addAncestors :551-559) -- and here it is part of the tree."""
    D = lambda p, mode=0o755, **kw: dict({"relpath": p, "kind": M.KIND_DIR, "mode": 0o40000 | mode, "mtime_sec": 100, "size": 0}, **kw)   # noqa: E731
    F = lambda p, mode=0o755: {"relpath": p, "kind": M.KIND_FILE, "mode": 0o100000 | mode, "mtime_sec": 100, "size": 5}                   # noqa: E731
    rel = lambda fs: [e["relpath"] for e in fs.entries()]                                                                                 # noqa: E731
    with M.MemFS("/tmp", now_sec=777) as fs:
        assert fs.update_from_entries([D("/test1"), D("/test1/test2")]) == 2 and rel(fs) == ["test1", "test1/test2"]       # Simple
        assert fs.update_from_entries([F("/test1", 0o777)]) == 1 and rel(fs) == ["test1"]                                    # Mutation
        fs.reset()
        assert fs.update_from_entries([D("test1/"), D("test1/test2/")]) == 2 and rel(fs) == ["test1", "test1/test2"]         # TrailingSlashes
        assert fs.update_from_entries([D("test1/"), D("test1/test2/")]) == 0                                                 # similar: nothing merged
        fs.reset()
        assert fs.update_from_entries([D("/test1", 0o700), D("/test1/test2/test3")]) == 3                                   # SkipDir...: + the made-up one
        made = fs.entries()[1]
        assert made["relpath"] == "test1/test2" and made["kind"] == M.KIND_DIR and made["mode"] & 0o7777 == 0o700
        assert (made["mtime_sec"], made["uid"], made["gid"], made["src"]) == (777, 0, 0, "/")
        fs.reset()
        l1 = [D("/test11"), D("/test11/test12"), F("/test11/test12/test.txt")]
        fs.update_from_entries(l1)
        # WhiteoutExistingDir: two headers in the merged layer -- the whiteout and /test11, carried as its ancestor
        assert fs.update_from_entries([D("/test11"), D("/test11/.wh.test12")]) == 2 and rel(fs) == ["test11"]
        fs.reset()
        fs.update_from_entries(l1)
        fs.update_from_entries([D("/test11"), D("/test11/.wh.test13")])                                                     # WhiteoutNonexistent...
        assert rel(fs) == ["test11", "test11/test12", "test11/test12/test.txt"]
        with pytest.raises(M.MiError) as ei:                                                                                 # the reference's failure
            fs.update_from_entries([{"relpath": "lnk", "kind": M.KIND_SYMLINK, "mode": 0o120777, "mtime_sec": 1, "size": 0,
                                     "link_target": "/test11"}, F("lnk/a/b")])
        assert "add hdr from tar to layer: update memfs with file /lnk/a/b: missing intermediate directory a in /lnk/a/b" in str(ei.value)
        assert fs.update_from_entries([F("/after")]) == 1                                                                    # the handle stays usable


def test_from_copy_run_sequence(tmp_path):
    """The calls build_stage.go makes for `FROM base; COPY src /app/; RUN touch/rm`: the base layer's headers merged (the
    files are on disk: modifyfs), a copy layer whose nodes remember their SOURCE, a scan layer.  isOnDisk asks about a node's
    src (mem_fs.go:49-57): /app/a.txt was not copied to disk (no modifyfs for this step) -- yet the scan writes no whiteout
    for it, because its source still exists; once the source is gone too, the next scan does."""
    root = str(tmp_path / "rootfs")
    ctx = str(tmp_path / "ctx")
    _mk(root, [("/etc/conf", "f", "base"), ("/usr/bin/tool", "f", "tool"), ("/app", "d", "")])
    _mk(ctx, [("/src/a.txt", "f", "A"), ("/src/sub/b.txt", "f", "B")])
    base = [e for e in M.tree_walk(root, root, (), M.TREE_SCAN, full=True) if e["relpath"] != "."]
    with M.MemFS(root) as fs:
        assert fs.update_from_entries(base) == len(base)
        assert fs.scan() == []                                                   # disk == tree
        op = {"src_root": ctx, "srcs": ["src"], "dst": "/app/", "uid": 0, "gid": 0}
        layer = fs.add_layer_by_copy_ops([op])
        assert _names(layer) == ["/app", "/app/a.txt", "/app/sub", "/app/sub/b.txt"]
        assert {e["relpath"]: e["src"] for e in layer}["app/a.txt"] == ctx + "/src/a.txt"
        assert _names(fs.add_layer_by_copy_ops([op])) == ["/app"]                # the same copy again: only the destination,
                                                                                 # which addAncestors always carries
        # RUN: one file appears, one base file disappears; the copied files are NOT on disk under /app
        _mk(root, [("/app/gen.txt", "f", "generated")])
        os.unlink(os.path.join(root, "usr/bin/tool"))
        l3 = fs.scan()
        names = _names(l3)
        assert "/app/gen.txt" in names and "/usr/bin/.wh.tool" in names
        assert not any(".wh.a.txt" in n or ".wh.sub" in n for n in names)         # their sources exist
        shutil.rmtree(os.path.join(ctx, "src"))
        os.utime(os.path.join(root, "app"), (5, 5))                              # /app is "changed", so it is looked at again
        l4 = fs.scan()
        assert "/app/.wh.a.txt" in _names(l4) and "/app/.wh.sub" in _names(l4)


def test_copy_layer_and_scan_layer_agree(tmp_path):
    """TestAddLayersEqual's intent (mem_fs_test.go:1118-1196): the layer of a copy and the layer of a scan after doing
    that copy on disk hold the same paths with the same headers (mtime of created directories aside)."""
    root = str(tmp_path / "rootfs")
    ctx = str(tmp_path / "ctx")
    os.makedirs(root)
    _mk(ctx, [("/c/x", "f", "1"), ("/c/d/y", "f", "22"), ("/c/l", "l", "x")])
    st = os.lstat(ctx)
    with M.MemFS(root) as a, M.MemFS(root) as b:
        la = a.add_layer_by_copy_ops([{"src_root": ctx, "srcs": ["c"], "dst": "/dst/", "uid": st.st_uid, "gid": st.st_gid}])
        shutil.copytree(os.path.join(ctx, "c"), os.path.join(root, "dst"), symlinks=True)
        lb = b.scan()
        key = lambda e: (e["relpath"], e["kind"], e["mode"], e["size"], e["link_target"], e["uid"], e["gid"])   # noqa: E731
        assert [key(e) for e in la] == [key(e) for e in lb]


# ---- generated sequences: the handle against the stateless calls and the model ------------------------------------------

def _plain(e):
    return {k: e.get(k) for k in ("relpath", "kind", "mode", "mtime_sec", "uid", "gid", "size", "link_target")}


@settings(max_examples=600, deadline=None, derandomize=True, database=None)
@given(st.lists(st.lists(_entry(), min_size=0, max_size=7), min_size=1, max_size=4))
def test_update_from_entries_equals_the_model_made_up_directories_included(layers):
    tree = Node({"kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 1, "uid": 0, "gid": 0, "size": 0, "link_target": None,
                 "relpath": ""}, "/")
    with M.MemFS("/", now_sec=1 << 40) as fs:
        for layer in layers:
            try:
                model_update_from_tar(tree, layer)
            except ReferenceFails as e:
                with pytest.raises(M.MiError) as ei:
                    fs.update_from_entries(layer)
                assert str(e)[:150] in str(ei.value)
                return
            fs.update_from_entries(layer)
            got = {"/" + e["relpath"]: e for e in fs.entries()}
            want = {}

            def walk(n, p):
                for name, c in n.children.items():
                    q = p.rstrip("/") + "/" + name
                    want[q] = c
                    walk(c, q)
            walk(tree, "/")
            assert sorted(got) == sorted(want)
            for p, node in want.items():
                g = got[p]
                if node.made_up:
                    assert (g["kind"], g["mtime_sec"], g["uid"], g["gid"]) == (M.KIND_DIR, 1 << 40, 0, 0), p
                else:
                    h = node.hdr
                    assert (g["kind"], g["mode"], g["mtime_sec"], g["uid"], g["size"]) == \
                        (h["kind"], h["mode"], h["mtime_sec"], h["uid"], h["size"]), p
                    if h["kind"] == M.KIND_SYMLINK:
                        assert g["link_target"] == h["link_target"]
                    elif h["kind"] == M.KIND_HARDLINK:
                        assert g["link_target"] == _abs(h["link_target"])
        assert flatten(tree).keys() <= got.keys()


from test_host_diff_properties import tree_pairs  # noqa: E402


def _materialize(root, entries):
    """the entry list as a real tree: sizes, modes, link targets and (last, deepest first) the mtimes"""
    for e in entries:
        full = os.path.join(root, e["relpath"])
        if e["kind"] == M.KIND_DIR:
            os.mkdir(full)
        elif e["kind"] == M.KIND_SYMLINK:
            os.symlink(e["link_target"], full)
        else:
            with open(full, "wb") as f:
                f.write(b"x" * e["size"])
        if e["kind"] != M.KIND_SYMLINK:
            os.chmod(full, e["mode"] & 0o7777)
    for e in sorted(entries, key=lambda e: -e["relpath"].count("/")):
        os.utime(os.path.join(root, e["relpath"]), (e["mtime_sec"], e["mtime_sec"]), follow_symlinks=False)


@settings(max_examples=150, deadline=None, derandomize=True, database=None)
@given(tree_pairs())
def test_scan_layer_of_the_handle_equals_the_stateless_diff(tmp_path_factory, pair):
    """mi_memfs_add_layer_by_scan against mi_snapshot_diff: the tree holds `before`, the disk holds `after` (generated,
    mutated, then really created: the handle asks the disk whether a node's source is gone)."""
    before, after = pair
    root = str(tmp_path_factory.mktemp("memfs_root"))
    _materialize(root, after)
    walked = M.tree_walk(root, root, (), M.TREE_SCAN, full=True)
    assert [e["relpath"] for e in walked[1:]] == [e["relpath"] for e in after]
    flags, wh = M.snapshot_diff(before, walked, disk_root=root)          # (with the root: its children can be deleted too)
    want_content = {"/" + e["relpath"] for e, f in zip(walked, flags) if f != M.DIFF_SAME}
    want_wh = {"/" + e["relpath"] for e, w in zip(before, wh) if w}
    with M.MemFS(root) as fs:
        fs.update_from_entries(before)
        layer = fs.add_layer_by_scan(walked)
        got_content, got_wh = set(), set()
        for e in layer:
            d, b = os.path.split("/" + e["relpath"])
            if b.startswith(".wh."):
                got_wh.add(os.path.join(d, b[4:]))
            else:
                got_content.add("/" + e["relpath"])
        assert got_wh == want_wh and got_content == want_content
        assert [("/" + e["relpath"]) for e in fs.entries()] == sorted("/" + e["relpath"] for e in after)   # the tree IS the disk now
        assert fs.add_layer_by_scan(walked) == []


def test_paths_longer_than_any_small_buffer(tmp_path):
    """a path of 3 000 bytes (a layer's map looks keys up through a 512-byte scratch buffer; the tree, the memo and the
    walk's path buffer have no such limit either): merged, merged again unchanged, changed, scanned, whited out"""
    D = lambda p: {"relpath": p, "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 100, "size": 0}        # noqa: E731
    F = lambda p, t=100: {"relpath": p, "kind": M.KIND_FILE, "mode": 0o100644, "mtime_sec": t, "size": 5}   # noqa: E731
    deep = "/".join(["d" * 250] * 11)                            # 11 levels of 250 bytes
    chain = ["/".join(deep.split("/")[:k]) for k in range(1, 12)]
    with M.MemFS(str(tmp_path)) as fs:
        assert fs.update_from_entries([D(c) for c in chain] + [F(deep + "/a"), F(deep + "/b")]) == 13
        assert fs.update_from_entries([F(deep + "/a"), F(deep + "/b")]) == 0
        assert fs.update_from_entries([F(deep + "/a", 200), F(deep + "/b", 200), F(deep + "/c")]) == 11 + 3      # the chain is carried once
        walked = [D(".")] + [D(c) for c in chain] + [F(deep + "/a", 200), F(deep + "/c")]
        layer = fs.add_layer_by_scan(walked)
        assert _names(layer) == ["/" + c for c in chain] + ["/" + deep + "/.wh.b"]
        assert [e["relpath"] for e in fs.entries()] == chain + [deep + "/a", deep + "/c"]


def test_merge_and_scan_of_sixty_thousand_entries_equal_the_stateless_diff(tmp_path):
    """C2's entry count is 100 000, C4's ten million: at that size the handle answers
    directory-by-directory input from what it keeps between two entries -- the parent node of the last lookup, the last
    addAncestors chain, marks for "the walk lists this path".  300 directories x 200 files three levels down, merged as a
    layer, then scanned against a walk in which files changed, went, came, one directory went with everything in it
    and one came: the layer is the stateless diff's (mi_snapshot_diff shares none of that state), the tree is the walk."""
    import random
    rng = random.Random(11)
    D = lambda p: {"relpath": p, "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 100, "size": 0, "uid": 0, "gid": 0}          # noqa: E731
    F = lambda p, t=100: {"relpath": p, "kind": M.KIND_FILE, "mode": 0o100644, "mtime_sec": t, "size": 5, "uid": 0, "gid": 0}    # noqa: E731
    before = [D("top"), D("top/mid")]
    for d in range(300):
        dn = "top/mid/d%03d" % d
        before.append(D(dn))
        before += [F("%s/f%03d" % (dn, k)) for k in range(200)]
    after = []
    for e in before:
        p = e["relpath"]
        if p.startswith("top/mid/d123"):
            continue                                              # a directory gone with all it held
        if e["kind"] == M.KIND_FILE and rng.random() < 0.002:
            continue                                              # a file gone
        after.append(F(p, 200) if e["kind"] == M.KIND_FILE and rng.random() < 0.003 else e)
    after += [D("top/mid/new")] + [F("top/mid/new/n%02d" % k) for k in range(50)] + [F("top/mid/d007/zz-added")]
    after.sort(key=lambda e: e["relpath"].split("/"))              # filepath.Walk order
    root = str(tmp_path)                                           # nothing of it is on disk: a missing path IS gone
    walked = [dict(D("."), relpath=".")] + after
    flags, wh = M.snapshot_diff(before, walked)
    want = sorted(["/" + e["relpath"] for e, f in zip(walked, flags) if f != M.DIFF_SAME] +
                  [os.path.join(os.path.dirname("/" + e["relpath"]), ".wh." + os.path.basename(e["relpath"])) for e, w in zip(before, wh) if w],
                  key=lambda q: os.path.join(os.path.dirname(q), os.path.basename(q)[4:]) if os.path.basename(q).startswith(".wh.") else q)
    with M.MemFS(root) as fs:
        assert fs.update_from_entries(before) == len(before)
        layer = fs.add_layer_by_scan(walked)
        assert _names(layer) == want and len(want) > 300
        assert "/top/mid/.wh.d123" in want and not any(q.startswith("/top/mid/d123/") for q in want)     # one whiteout for the subtree
        assert [e["relpath"] for e in fs.entries()] == sorted(e["relpath"] for e in after)
        assert fs.add_layer_by_scan(walked) == []


def test_plain_c_build_stage(tmp_path):
    """tests/cabi/memfs_driver.c: NewMemFS, a scan layer, a RUN step, another scan layer -- from plain C against the
    header; the layers are those the python harness gets, and extracting one over the other reproduces the root."""
    import hashlib
    import subprocess
    import tarfile
    exe = str(tmp_path / "memfs_driver")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(repo, "include"),
                           os.path.join(repo, "tests", "cabi", "memfs_driver.c"), "-o", exe,
                           "-L", os.path.join(repo, "makisu_amd"), "-lmakisu_mi", "-Wl,-rpath," + os.path.join(repo, "makisu_amd")])
    root = str(tmp_path / "root")
    _mk(root, [("/etc/passwd", "f", "root"), ("/etc/deep/er/x.conf", "f", "x"), ("/bin/tool", "f", "t" * 70000),
               ("/bin/alias", "l", "tool")])
    twin = str(tmp_path / "twin")
    shutil.copytree(root, twin, symlinks=True)
    cmd = "cd %s && rm -rf etc/deep bin/tool && echo more >> etc/passwd && mkdir new && echo hi > new/f"
    out = subprocess.run([exe, root, cmd % root, str(tmp_path / "1.tar"), str(tmp_path / "2.tar")], check=True,
                         capture_output=True, text=True).stdout.splitlines()
    names = {k: [ln.split(" ", 2)[2] for ln in out if ln.startswith("E %d " % k)] for k in (1, 2)}
    layers = {int(ln.split()[1]): ln.split()[2:] for ln in out if ln.startswith("L ")}
    for k in (1, 2):
        assert layers[k][1] == hashlib.sha256((tmp_path / ("%d.tar" % k)).read_bytes()).hexdigest()
        assert int(layers[k][0]) == len(names[k])
    assert names[2] == ["bin", "bin/.wh.tool", "etc", "etc/.wh.deep", "etc/passwd", "new", "new/f"]
    with M.MemFS(twin) as fs:                                    # the same stage through the python harness, on a twin
        assert [e["relpath"] for e in fs.scan()] == names[1]
        subprocess.check_call(cmd % twin, shell=True)
        assert [e["relpath"] for e in fs.scan()] == names[2]
        assert int([ln for ln in out if ln.startswith("T ")][0].split()[1]) == len(fs.entries())
    with tarfile.open(tmp_path / "2.tar") as tf:
        assert [m.name.rstrip("/") for m in tf.getmembers()] == names[2]
        assert tf.extractfile("etc/passwd").read() == b"rootmore\n" and tf.getmember("bin/.wh.tool").size == 0


def test_content_roots_make_the_next_scan_content_aware(tmp_path):
    """isUpdated's seam.  A same-size, same-second edit is invisible to tario.IsSimilarHeader; with
    the chunk roots of each scan kept in the tree (here: stand-in 32-byte values, on the GPU box the batch's roots), the
    next scan sees it -- and only it."""
    import numpy as np
    root = str(tmp_path)
    _mk(root, [("/a/x", "f", "1111"), ("/a/y", "f", "2222"), ("/b", "f", "3333")])
    for p in ("a/x", "a/y", "b", "a"):
        os.utime(os.path.join(root, p), (1000, 1000))
    walked = M.tree_walk(root, root, (), M.TREE_SCAN, full=True)
    files = [e for e in walked if e["kind"] == M.KIND_FILE]
    assert [e["file_index"] for e in files] == [0, 1, 2]
    r1 = np.arange(3 * 32, dtype=np.uint8).reshape(3, 32)
    with M.MemFS(root) as fs:
        assert len(fs.add_layer_by_scan(walked, r1)) == 4
        with open(os.path.join(root, "a/y"), "w") as f:            # same size; mtime put back: the header is identical
            f.write("zzzz")
        os.utime(os.path.join(root, "a/y"), (1000, 1000))
        os.utime(os.path.join(root, "a"), (1000, 1000))
        walked2 = M.tree_walk(root, root, (), M.TREE_SCAN, full=True)
        assert [{k: v for k, v in e.items()} for e in walked2] == walked
        r2 = r1.copy()
        r2[1, 7] ^= 1                                                # what the GPU would report for the new content
        assert fs.add_layer_by_scan(walked2, None) == []             # the reference's rule: nothing changed
        assert [e["relpath"] for e in fs.add_layer_by_scan(walked2, r2)] == ["a", "a/y"]
        assert fs.add_layer_by_scan(walked2, r2) == []               # the new root is the tree's now


def test_commit_layer_in_one_call(tmp_path):
    """step.commitLayer (common.go:67-111) on the handle: nothing to do / by copy ops / by scan -- the DigestPair's numbers
    are those of the blob it wrote, and the same layer written by hand through the layer writer has the same TarDigest."""
    import gzip
    import hashlib
    import io
    import tarfile
    root, ctx = str(tmp_path / "root"), str(tmp_path / "ctx")
    _mk(root, [("/etc/conf", "f", "base")])
    _mk(ctx, [("/src/a.txt", "f", "A" * 5000), ("/src/sub/b.txt", "f", "B")])
    with M.MemFS(root) as fs:
        assert fs.commit_layer() is None                                        # "Nothing to do, return."
        fd = os.open(str(tmp_path / "1.tar.gz"), os.O_WRONLY | os.O_CREAT, 0o644)
        pair = fs.commit_layer(must_scan=True, out_fd=fd)
        os.close(fd)
        blob = (tmp_path / "1.tar.gz").read_bytes()
        raw = gzip.decompress(blob)
        assert pair["gzip_digest"].hex() == hashlib.sha256(blob).hexdigest() and pair["gzip_bytes"] == len(blob)
        assert pair["tar_digest"].hex() == hashlib.sha256(raw).hexdigest() and pair["tar_bytes"] == len(raw)
        with tarfile.open(fileobj=io.BytesIO(raw)) as tf:
            assert [m.name.rstrip("/") for m in tf.getmembers()] == ["etc", "etc/conf"] == [e["relpath"] for e in pair["layer"]]
        op = {"src_root": ctx, "srcs": ["src"], "dst": "/app/", "uid": 0, "gid": 0}
        fd = os.open(str(tmp_path / "2.tar"), os.O_WRONLY | os.O_CREAT, 0o644)
        pair2 = fs.commit_layer(ops=[op], out_fd=fd, gzip_level=M.GZIP_OFF)
        os.close(fd)
        raw2 = (tmp_path / "2.tar").read_bytes()
        assert pair2["tar_digest"].hex() == hashlib.sha256(raw2).hexdigest() and pair2["n_entries"] == 4
        with tarfile.open(fileobj=io.BytesIO(raw2)) as tf:
            assert tf.extractfile("app/a.txt").read() == b"A" * 5000
        empty = fs.commit_layer(must_scan=True)                                  # nothing changed on disk: an EMPTY layer is
        assert empty["n_entries"] == 0 and empty["tar_bytes"] == 1024            # still a layer (the tar trailer, 5f70bf18...)
        assert empty["tar_digest"].hex() == "5f70bf18a086007016e948b04aed3b82103a36bea41755b6cddfaf10ace3c6ef"
    with M.MemFS(root) as twin:                                                  # by hand: the same TarDigest
        layer = twin.scan()
        with M.Layer(out_fd=-1) as lw:
            for e in layer:
                lw.add(e, e["src"] if e["kind"] == M.KIND_FILE else None)
            assert lw.finish()["tar_digest"] == pair["tar_digest"]


@settings(max_examples=500, deadline=None, derandomize=True, database=None)
@given(st.lists(st.sampled_from(["a", "b", ".", "..", "", "a.", ".a", "..a", "a..", "...", "a b"]), min_size=1, max_size=5),
       st.sampled_from(["", "/", "./", "//"]), st.sampled_from(["", "/", "//", "/."]))
def test_scan_and_merge_read_a_path_the_same_way(tmp_path_factory, parts, head, tail):
    """both doors normalise an entry's name like pathutils.AbsPath (path.Join("/", TrimRight(p, "/"))): the scan takes a
    short cut for names that are already clean, the merge never does"""
    rel = head + "/".join(parts) + tail
    root = str(tmp_path_factory.mktemp("norm"))
    e = {"relpath": rel, "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 3, "uid": 0, "gid": 0, "size": 0}
    with M.MemFS(root) as a, M.MemFS(root) as b:
        a.add_layer_by_scan([e])
        b.update_from_entries([e])
        assert [x["relpath"] for x in a.entries()] == [x["relpath"] for x in b.entries()], rel


def test_c1_scan_and_commit_of_the_reference_build_context(tmp_path):
    """BASELINE.json configs[0]: "lib/snapshot scan+commit of testdata/build-context ... (plumbing, no GPU)" -- the
    reference's own fixture tree (28 files, 10 355 bytes; tests/golden/build_context_c1.json carries it) through NewMemFS,
    AddLayerByScan and commitLayer: every file arrives in the layer tar with its bytes (SHA-256 per file as the fixture
    states), in sort.Strings order, directories before their content; the DigestPair is that of the blob; a second
    commit is the empty layer; a COPY of the tree elsewhere frames to the same members under the new prefix."""
    import base64
    import gzip
    import hashlib
    import io
    import json
    import tarfile
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "build_context_c1.json")))["entries"]
    root = tmp_path / "root"
    for e in gold:
        p = root / "ctx" / e["path"]
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(base64.b64decode(e["b64"]))
    assert len(gold) == 28 and sum(e["size"] for e in gold) == 10355
    with M.MemFS(str(root)) as fs:
        fd = os.open(str(tmp_path / "layer.tar.gz"), os.O_WRONLY | os.O_CREAT, 0o644)
        pair = fs.commit_layer(must_scan=True, out_fd=fd)
        os.close(fd)
        blob = (tmp_path / "layer.tar.gz").read_bytes()
        raw = gzip.decompress(blob)
        assert pair["tar_digest"] == "sha256:" + hashlib.sha256(raw).hexdigest()
        assert pair["gzip_digest"] == "sha256:" + hashlib.sha256(blob).hexdigest() and pair["gzip_bytes"] == len(blob)
        with tarfile.open(fileobj=io.BytesIO(raw)) as tf:
            members = tf.getmembers()
            names = [m.name.rstrip("/") for m in members]
            assert names == sorted(names) and names[0] == "ctx"
            files = {m.name: hashlib.sha256(tf.extractfile(m).read()).hexdigest() for m in members if m.isfile()}
        assert files == {"ctx/" + e["path"]: e["sha256"] for e in gold}
        assert all(m.uname == "" and m.gname == "" for m in members)                 # createHeader blanks the names
        assert fs.commit_layer(must_scan=True)["tar_bytes"] == 1024                   # nothing changed: the empty layer
        op = {"src_root": str(root), "srcs": ["ctx"], "dst": "/app/", "uid": 0, "gid": 0}
        copied = fs.commit_layer(ops=[op])
        assert [e["relpath"] for e in copied["layer"]][1:] == ["app/" + n[len("ctx/"):] for n in names[1:]]
        assert copied["n_entries"] == len(names)


def test_a_file_named_like_a_whiteout_is_keyed_by_the_path_it_deletes(tmp_path):
    """memLayer.addHeader (mem_layer.go:197-212): a path whose base name has the ".wh." prefix is filed under the path it
    DELETES -- rangeFiles sorts those keys (:232-244), so /d/.wh.m is committed where /d/m would be (between /d/a and
    /d/z), not where its own name sorts ('.' < 'a'); and at the root the key is /gone, not //gone."""
    root = str(tmp_path)
    _mk(root, [("/d/a", "f", "1"), ("/d/.wh.m", "f", ""), ("/d/z", "f", "2"), ("/.wh.gone", "f", ""), ("/e", "f", "3"), ("/h", "f", "4")])
    with M.MemFS(root) as fs:
        layer = fs.scan()
        assert _names(layer) == ["/d", "/d/a", "/d/.wh.m", "/d/z", "/e", "/.wh.gone", "/h"]
        assert [e["file_index"] for e in layer] == [-1, 0, -1, 1, 2, -1, 3]      # a whiteout has no content


def test_a_hard_link_that_did_not_change_is_not_merged_again(tmp_path):
    """isUpdated through tario.IsSimilarHeader's hard-link case (compare.go:62-83): same target, owner, mode and second ->
    similar -> the second merge of the same member adds nothing; another target does."""
    base = [{"relpath": "bin", "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 5},
            {"relpath": "bin/busybox", "kind": M.KIND_FILE, "mode": 0o100755, "size": 9, "mtime_sec": 5},
            {"relpath": "bin/sh", "kind": M.KIND_HARDLINK, "mode": 0o100755, "mtime_sec": 5, "link_target": "bin/busybox"}]
    with M.MemFS(str(tmp_path)) as fs:
        assert fs.update_from_entries(base) == 3
        assert fs.update_from_entries(base) == 0
        assert fs.update_from_entries([dict(base[2], mtime_sec=6)]) == 2         # the link and its directory
        assert fs.update_from_entries([dict(base[2], mtime_sec=6, link_target="/bin/busybox")]) == 0   # AbsPath either way
        assert fs.update_from_entries([dict(base[2], mtime_sec=6, link_target="bin/other")]) == 2


def test_merged_header_count_is_the_number_of_distinct_keys(tmp_path):
    """"Merged %d headers from tar to memfs" (mem_fs.go:250) is len(l.files): a path that occurs twice in a layer, an
    ancestor re-added by several children, a whiteout filed under the path it deletes -- each key counts once.  The merge
    keeps fingerprints of the keys instead of the map (nothing reads it again); the count is the map's."""
    D = lambda p: {"relpath": p, "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 100, "size": 0}                 # noqa: E731
    F = lambda p, t=100: {"relpath": p, "kind": M.KIND_FILE, "mode": 0o100644, "mtime_sec": t, "size": 5}         # noqa: E731
    with M.MemFS(str(tmp_path)) as fs:
        assert fs.update_from_entries([D("a"), F("a/x"), F("a/x", 101), F("a/y"), D("b"), F("b/z")]) == 5     # a/x twice: one key
        assert fs.update_from_entries([F("a/x", 102), F("a/.wh.x")]) == 2                                          # "a" + "a/x" (both headers file under a/x)
        assert [e["relpath"] for e in fs.entries()] == ["a", "a/y", "b", "b/z"]
        many = [D("m")] + [x for d in range(300) for x in [D("m/d%03d" % d)] + [F("m/d%03d/f%03d" % (d, k)) for k in range(40)]]
        assert fs.update_from_entries(many) == len(many)
        assert fs.update_from_entries(many) == 0                                                                   # all similar: nothing merged
        touched = [dict(e, mtime_sec=200) for e in many if e["kind"] == M.KIND_FILE][::7]
        assert fs.update_from_entries(touched) == len(touched) + len({"m"} | {e["relpath"].rsplit("/", 1)[0] for e in touched})


def test_directories_whose_names_begin_alike_in_a_tars_arbitrary_order(tmp_path):
    """The tree keeps, per depth, the directory of the last lookup and a place among its children -- matched by the directory's
    whole path, not a prefix of it.  A tar may list "a", "ab", "abc" and "a/b", "a/bb" in any order, jumping between them; every
    entry must land below ITS directory, a later layer's changes on the right nodes, and the scan of the same tree finds nothing."""
    D = lambda p: {"relpath": p, "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 100, "size": 0}                 # noqa: E731
    F = lambda p, t=100: {"relpath": p, "kind": M.KIND_FILE, "mode": 0o100644, "mtime_sec": t, "size": 5}         # noqa: E731
    first = [D("a"), D("ab"), D("abc"), F("a/x"), F("ab/x"), F("abc/x"), F("a/y"), F("abc/y"), F("ab/y"), D("a/b"), D("a/bb"),
             F("a/bb/q"), F("a/b/q"), F("a/bb/r"), F("a/b/r"), F("abc/a"), F("a/a"), F("ab/a")]
    with M.MemFS(str(tmp_path)) as fs:
        assert fs.update_from_entries(first) == len(first)
        assert [e["relpath"] for e in fs.entries()] == sorted(e["relpath"] for e in first)
        assert fs.update_from_entries(first) == 0
        second = [F("ab/x", 101), F("a/x", 102), F("abc/x", 103), F("a/b/q", 104), F("a/bb/q", 105), F("a/y", 106), F("abc/y", 107)]
        assert fs.update_from_entries(second) == len(second) + len({"a", "ab", "abc", "a/b", "a/bb"})
        got = {e["relpath"]: e["mtime_sec"] for e in fs.entries()}
        want = {e["relpath"]: e["mtime_sec"] for e in first}
        want.update({e["relpath"]: e["mtime_sec"] for e in second})
        assert got == want
        for e in fs.entries():                                                      # the same tree on disk: the scan finds nothing
            full = os.path.join(str(tmp_path), e["relpath"])
            if e["kind"] == M.KIND_DIR:
                os.makedirs(full, exist_ok=True)
            else:
                os.makedirs(os.path.dirname(full), exist_ok=True)
                with open(full, "w") as f:
                    f.write("12345")
                os.chmod(full, 0o644)
        for e in sorted(fs.entries(), key=lambda e: -e["relpath"].count("/")):
            os.utime(os.path.join(str(tmp_path), e["relpath"]), (e["mtime_sec"], e["mtime_sec"]))
        os.chown(str(tmp_path), 0, 0) if os.geteuid() == 0 else None
        res = fs.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF)
        assert [e["relpath"] for e in res["layer"]] == []


def test_a_failing_copy_op_leaves_what_the_reference_has_applied_by_then(tmp_path):
    """addToLayer (mem_fs.go:343-421) handles an op's sources one after the other: a source that does not exist fails the op
    BEFORE its destination chain is created (the stat comes first); a source that cannot be resolved -- a link leaving the
    context -- fails it AFTER the chain and the sources before it were applied.  The ops are planned (disk) and applied
    (tree) in two steps here; every failure keeps its place."""
    ctx, root = str(tmp_path / "ctx"), str(tmp_path / "root")
    os.makedirs(root)
    _mk(ctx, [("/good/a.txt", "f", "a"), ("/good/sub/b.txt", "f", "b"), ("/escape", "l", "/etc")])
    rel = lambda fs: [e["relpath"] for e in fs.entries()]                                  # noqa: E731
    with M.MemFS(root) as fs:
        ops = [{"src_root": ctx, "srcs": ["good"], "dst": "/one/"}, {"src_root": ctx, "srcs": ["missing"], "dst": "/two/"}]
        with pytest.raises(M.MiError) as ei:
            fs.add_layer_by_copy_ops(ops)
        assert "stat src" in str(ei.value)
        assert rel(fs) == ["one", "one/a.txt", "one/sub", "one/sub/b.txt"]               # op 1 applied; nothing of op 2, not even /two
    with M.MemFS(root) as fs:
        ops = [{"src_root": ctx, "srcs": ["good", "escape"], "dst": "/three/"}]
        with pytest.raises(M.MiError) as ei:
            fs.add_layer_by_copy_ops(ops)
        assert "eval symlinks for escape" in str(ei.value) and "outside of root" in str(ei.value)
        assert rel(fs) == ["three", "three/a.txt", "three/sub", "three/sub/b.txt"]       # the chain and the first source ARE in the tree
        assert fs.add_layer_by_copy_ops(ops[:0] + [{"src_root": ctx, "srcs": ["good"], "dst": "/three/"}])[0]["relpath"] == "three"
