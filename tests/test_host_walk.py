"""CPU tests of the host-side walks (csrc/mi_tree.hip, no GPU needed): filepath.Walk order,
checksumPathContents' skip rule (lib/builder/step/add_copy_step.go:194-238) and the snapshot
walk's shouldSkip (lib/snapshot/utils.go:37-75), against an independent Python emulation.
The table cases for IsDescendantOfAny follow lib/pathutils/path.go:24-35."""
import os
import stat

import numpy as np
import pytest


def _go_walk(root):
    st = os.lstat(root)
    yield root, st
    if stat.S_ISDIR(st.st_mode):
        for name in sorted(os.listdir(root), key=os.fsencode):
            yield from _go_walk(os.path.join(root, name))


def _special(st):
    return stat.S_ISFIFO(st.st_mode) or stat.S_ISSOCK(st.st_mode) or stat.S_ISCHR(st.st_mode) or stat.S_ISBLK(st.st_mode)


@pytest.fixture()
def tree(tmp_path):
    b = tmp_path / "ctx"
    os.makedirs(b / "a" / "deep" / "er")
    os.makedirs(b / "a-b")
    os.makedirs(b / "a.d")
    os.makedirs(b / "empty")
    os.makedirs(b / "z" / ".wh..wh.plnk")
    for rel, data in [("a/x.txt", b"x" * 10), ("a/deep/er/f", b"deep"), ("a-b/y", b""), ("a.d/q", b"q"),
                      ("a.txt", b"t"), ("z/last", b"l" * 3), ("z/.wh..wh.plnk/hidden", b"h"),
                      ("z/.wh.gone", b""), ("Z-upper", b"U"), ("\xc3\xa9-utf8".encode("latin1").decode("utf8"), b"e")]:
        (b / rel).write_bytes(data)
    os.symlink("x.txt", b / "a" / "lnk")
    os.symlink("/no/such", b / "dangling")
    os.symlink("a", b / "dirlink")                      # a symlink to a directory is NOT descended into
    os.mkfifo(b / "a" / "fifo")
    # every kind utils.IsSpecialFile names (lib/utils/utils.go:161-163); device nodes where this process may make them
    import socket
    sk = socket.socket(socket.AF_UNIX)
    sk.bind(str(b / "a" / "sock"))
    sk.close()
    try:
        os.mknod(b / "a" / "chr", 0o600 | stat.S_IFCHR, os.makedev(1, 3))
        os.mknod(b / "z" / "blk", 0o600 | stat.S_IFBLK, os.makedev(7, 0))
    except PermissionError:
        pass
    return b


def test_context_walk_matches_go_order(tree, engine_lib):
    import makisu_amd
    got = makisu_amd.tree_walk(str(tree))
    want = [(os.path.relpath(p, tree), st) for p, st in _go_walk(str(tree)) if not _special(st)]
    assert [g[0] for g in got] == [w[0] for w in want]
    assert got[0][0] == "."
    kinds = {g[0]: g[4] for g in got}
    assert kinds["a"] == 0 and kinds["a/x.txt"] == 1 and kinds["a/lnk"] == 2 and kinds["dirlink"] == 2
    assert not {"a/fifo", "a/sock", "a/chr", "z/blk"} & set(kinds)      # utils.IsSpecialFile
    assert not any(k.startswith("dirlink/") for k in kinds)
    links = {g[0]: g[1] for g in got if g[4] == 2}
    assert links == {"a/lnk": "x.txt", "dangling": "/no/such", "dirlink": "a"}
    # Walk order is per-directory lexical, NOT a sort of full paths: "a/..." comes before "a-b"
    names = [g[0] for g in got]
    assert names.index("a/x.txt") < names.index("a-b") < names.index("a.d") < names.index("a.txt")
    assert names != sorted(names)
    # regular files get running ordinals in visit order and stat sizes
    regs = [g for g in got if g[4] == 1]
    assert [g[2] for g in regs] == list(range(len(regs)))
    assert {g[0]: g[3] for g in regs}["a/x.txt"] == 10
    # context mode keeps whiteout-looking names: only special files are skipped
    assert "z/.wh..wh.plnk/hidden" in kinds and "z/.wh.gone" in kinds


def test_scan_walk_should_skip(tree, engine_lib):
    import makisu_amd
    os.unlink(tree / "dangling")        # absolute target outside the root: createHeader would fail (below)
    bl = [str(tree / "a" / "deep"), str(tree / "a.d") + "/"]
    got = [g[0] for g in makisu_amd.tree_walk(str(tree), rel_base=str(tree.parent), blacklist=bl,
                                              mode=makisu_amd.TREE_SCAN)]
    assert got[0] == "ctx"
    assert "ctx/z/.wh.gone" in got                      # a plain whiteout marker is kept
    assert not any(".wh..wh." in g for g in got)        # AUFS metadata pruned with its subtree
    assert not any(g == "ctx/a/deep" or g.startswith("ctx/a/deep/") for g in got)
    assert not any(g == "ctx/a.d" or g.startswith("ctx/a.d/") for g in got)   # trailing "/" normalised (AbsPath)
    assert "ctx/a/x.txt" in got and not {"ctx/a/fifo", "ctx/a/sock", "ctx/a/chr", "ctx/z/blk"} & set(got)
    # blacklisting "/" prunes everything (ancestor == "/" rule, path.go:29)
    assert makisu_amd.tree_walk(str(tree), blacklist=["/"], mode=makisu_amd.TREE_SCAN) == []
    # a path that only shares a name prefix with a blacklisted dir is NOT a descendant
    got2 = [g[0] for g in makisu_amd.tree_walk(str(tree), blacklist=[str(tree / "a")], mode=makisu_amd.TREE_SCAN)]
    assert "a-b" in got2 and "a.txt" in got2 and "a" not in got2 and "a/x.txt" not in got2


def test_scan_walk_trims_symlink_root(tree, engine_lib):
    """memLayer.createHeader (lib/snapshot/mem_layer.go:171-185): in the snapshot walk an ABSOLUTE
    symlink target loses the root prefix through pathutils.TrimRoot (the cases of
    lib/pathutils/path_test.go:60-79: inside the root, the root itself, outside -> error); relative
    targets and the context walk keep the raw os.Readlink value."""
    import makisu_amd
    os.unlink(tree / "dangling")
    root = str(tree)
    os.symlink(root + "/a/x.txt", tree / "abs-in-root")
    os.symlink(root + "/a//deep/../deep/er/", tree / "abs-unclean")      # AbsPath = path.Join: cleaned
    os.symlink(root + "/", tree / "abs-root")
    os.symlink(root, tree / "abs-root-bare")
    got = {e["relpath"]: e["link_target"] for e in
           makisu_amd.tree_walk(root, mode=makisu_amd.TREE_SCAN, full=True) if e["kind"] == 2}
    assert got["abs-in-root"] == "/a/x.txt"
    assert got["abs-unclean"] == "/a/deep/er"
    assert got["abs-root"] == "/" and got["abs-root-bare"] == "/"
    assert got["a/lnk"] == "x.txt" and got["dirlink"] == "a"             # relative: untouched
    raw = {e["relpath"]: e["link_target"] for e in makisu_amd.tree_walk(root, full=True) if e["kind"] == 2}
    assert raw["abs-in-root"] == root + "/a/x.txt"                        # context walk: raw target
    # TrimRoot is a plain string-prefix test (strings.HasPrefix): a sibling that merely shares the
    # prefix passes it, as in the reference
    os.symlink(root + "-sibling/f", tree / "abs-prefix-quirk")
    got = {e["relpath"]: e["link_target"] for e in
           makisu_amd.tree_walk(root, mode=makisu_amd.TREE_SCAN, full=True) if e["kind"] == 2}
    assert got["abs-prefix-quirk"] == "/-sibling/f"
    # outside the root: "failed to trim root prefix" -> the scan fails
    os.symlink("/no/such", tree / "dangling")
    with pytest.raises(makisu_amd.MiError) as ei:
        makisu_amd.tree_walk(root, mode=makisu_amd.TREE_SCAN)
    assert ei.value.code == -1
    # with the real root "/" every absolute target is inside: it is only cleaned
    assert {e["relpath"]: e["link_target"] for e in
            makisu_amd.tree_walk(root, rel_base="/", mode=makisu_amd.TREE_SCAN, full=True)
            if e["kind"] == 2}[root.lstrip("/") + "/dangling"] == "/no/such"


def test_walk_single_file_and_errors(tree, engine_lib):
    import makisu_amd
    # COPY of a single file: Walk visits just the file, relpath relative to the context dir
    one = makisu_amd.tree_walk(str(tree / "a" / "x.txt"), rel_base=str(tree))
    assert [(g[0], g[4]) for g in one] == [("a/x.txt", 1)]
    with pytest.raises(makisu_amd.MiError) as ei:
        makisu_amd.tree_walk(str(tree / "missing"))
    assert ei.value.code == -5
    with pytest.raises(makisu_amd.MiError) as ei:       # "write path is outside of context dir" (:205-209)
        makisu_amd.tree_walk(str(tree / "a"), rel_base=str(tree / "z"))
    assert ei.value.code == -1


# ---- mi_entry_similar: the cases of lib/tario/compare_test.go, on real lstat results ---------
def _entries(root):
    import makisu_amd
    return {e["relpath"]: e for e in makisu_amd.tree_walk(str(root), full=True)}


def _touch(p, t):
    import os
    os.utime(p, (t, t), follow_symlinks=False)


def test_entry_similar_mirrors_compare_test(tmp_path):
    import os
    import time
    import makisu_amd
    sim = makisu_amd.entry_similar
    now = int(time.time())
    (tmp_path / "dir1").mkdir()
    (tmp_path / "dir2").mkdir()
    (tmp_path / "dir2" / "child").write_bytes(b"x")          # DifferentContentConsideredSimilar (dirs)
    (tmp_path / "f1").write_bytes(b"test1")
    (tmp_path / "f2").write_bytes(b"test2")                  # same size, different content
    (tmp_path / "f3").write_bytes(b"test33")
    os.symlink(str(tmp_path / "f1"), tmp_path / "l1")
    os.symlink(str(tmp_path / "f1"), tmp_path / "l2")
    os.symlink(str(tmp_path / "f2"), tmp_path / "l3")
    for name in ("dir1", "dir2", "f1", "f2", "f3", "l1", "l2", "l3"):
        _touch(tmp_path / name, now - 7200)
    e = _entries(tmp_path)
    # TestIsSimilar
    assert not sim(e["dir1"], e["f1"])                       # DirAndFileConsideredDifferent
    assert sim(dict(e["f1"], relpath=""), dict(e["f2"], relpath=""))   # RootsConsideredSimilar
    assert not sim(e["dir1"], e["l1"])                       # DirAndSymlinkConsideredDifferent
    assert not sim(e["f1"], e["l1"])                         # FileAndSymlinkConsideredDifferent
    hard = dict(e["f1"], kind=makisu_amd.KIND_HARDLINK, link_target=str(tmp_path / "f1"), size=0)
    assert not sim(e["f1"], hard)                            # FileAndHardLinkConsideredDifferent
    # TestIsSimilarSymlink
    assert sim(e["l1"], e["l2"]) and sim(e["l1"], e["l2"], ignore_time=True)    # NoChange
    assert not sim(e["l1"], e["l3"])                         # DifferentLinkTargetConsideredDifferent
    _touch(tmp_path / "l2", now)                             # symlinks only compare the target
    assert sim(e["l1"], _entries(tmp_path)["l2"])
    # TestIsSimilarHardlink
    hard2 = dict(hard)
    assert sim(hard, hard2)                                  # NoChange
    assert not sim(hard, dict(hard, link_target=str(tmp_path / "f2")))          # DifferentLinkTarget
    assert not sim(hard, dict(hard, mtime_sec=hard["mtime_sec"] + 5))
    assert sim(hard, dict(hard, mtime_sec=hard["mtime_sec"] + 5), ignore_time=True)
    # TestIsSimilarDirectory
    assert sim(e["dir1"], e["dir2"])                         # NoChange / DifferentContentConsideredSimilar
    _touch(tmp_path / "dir1", now - 3600)
    e2 = _entries(tmp_path)
    assert not sim(e2["dir1"], e2["dir2"])                   # DifferentModTimeConsideredDifferent
    assert sim(e2["dir1"], e2["dir2"], ignore_time=True)     # ...SimilarIfIgnored
    _touch(tmp_path / "dir1", now - 7200)
    os.chmod(tmp_path / "dir1", 0o777)
    e2 = _entries(tmp_path)
    assert not sim(e2["dir1"], e2["dir2"])                   # DifferentModeConsideredDifferent
    # TestIsSimilarRegularFile
    assert sim(e["f1"], e["f2"])                             # NoChange / DifferentContentButSameSize
    assert not sim(e["f1"], e["f3"])                         # DifferentSizeConsideredDifferent
    _touch(tmp_path / "f1", now - 3600)
    e2 = _entries(tmp_path)
    assert not sim(e2["f1"], e2["f2"])                       # DifferentModTimeConsideredDifferent
    assert sim(e2["f1"], e2["f2"], ignore_time=True)         # ...SimilarIfIgnored
    _touch(tmp_path / "f1", now - 7200)
    os.chmod(tmp_path / "f1", 0o777)
    e2 = _entries(tmp_path)
    assert not sim(e2["f1"], e2["f2"])                       # DifferentModeConsideredDifferent
    assert not sim(e["f1"], dict(e["f2"], uid=e["f2"]["uid"] + 1))
    assert not sim(e["f1"], dict(e["f2"], gid=e["f2"]["gid"] + 1))
    # sub-second mtime differences vanish (Truncate(1s) in compare.go:67-69)
    os.utime(tmp_path / "f2", ns=((now - 7200) * 10**9 + 5 * 10**8,) * 2)
    assert sim(e["f1"], _entries(tmp_path)["f2"])
    # unsupported type
    with pytest.raises(makisu_amd.MiError):
        sim(dict(e["f1"], kind=9), e["f1"])


def test_entry_similar_content_roots():
    """The content-aware extension: same metadata + different chunk roots = updated."""
    import hashlib
    import makisu_amd
    a = {"relpath": "etc/passwd", "size": 5, "mtime_sec": 100, "mode": 0o100644, "kind": 1, "uid": 0, "gid": 0}
    r1, r2 = hashlib.sha256(b"test1").digest(), hashlib.sha256(b"test2").digest()
    assert makisu_amd.entry_similar(a, dict(a))                               # the reference's answer
    assert makisu_amd.entry_similar(a, dict(a), root_a=r1, root_b=r1)
    assert not makisu_amd.entry_similar(a, dict(a), root_a=r1, root_b=r2)     # same size, new content
    assert makisu_amd.entry_similar(a, dict(a), root_a=r1, root_b=None)       # one side unknown: metadata only
    d = dict(a, kind=0)
    assert makisu_amd.entry_similar(d, dict(d), root_a=r1, root_b=r2)         # roots only matter for files


def test_commit_order_is_sorted_dst_paths_not_walk_order(tree, engine_lib):
    """memLayer.rangeFiles (lib/snapshot/mem_layer.go:232-244): sort.Strings over the absolute dst
    paths; addHeader keys a whiteout marker by the path it deletes (:190-211)."""
    import makisu_amd
    os.unlink(tree / "dangling")
    rels = [g[0] for g in makisu_amd.tree_walk(str(tree), mode=makisu_amd.TREE_SCAN)]

    def key(r):
        dst = "/" if r == "." else "/" + r
        d, b = dst.rsplit("/", 1)
        return ((d + "/" + b[4:]) if b.startswith(".wh.") else dst).encode()

    order = makisu_amd.commit_order(rels)
    assert sorted(order) == list(range(len(rels)))
    assert order == sorted(range(len(rels)), key=lambda i: key(rels[i]))
    names = [rels[i] for i in order]
    assert names[0] == "."                                             # "/" first
    assert names.index("a-b") < names.index("a/x.txt")                 # '-' < '/': differs from Walk
    assert rels.index("a-b") > rels.index("a/x.txt")
    assert names.index("z/.wh.gone") < names.index("z/last")           # keyed as /z/gone
    # equal keys (a whiteout marker and the file it hides) keep input order; empty input is fine
    assert makisu_amd.commit_order(["d/.wh.x", "d/x", "d/.wh.x"]) == [0, 1, 2]
    assert makisu_amd.commit_order([]) == []
    import ctypes as C
    out = (C.c_uint64 * 2)()
    ents = (makisu_amd.TreeEntry * 2)()
    ents[0].relpath, ents[1].relpath = b"b", b"a"
    L = makisu_amd.load_library()
    assert L.mi_entries_commit_order(None, 2, out) == -1 and L.mi_entries_commit_order(ents, 2, None) == -1      # MI_ERR_INVALID
    assert L.mi_entries_commit_order(None, 0, None) == 0
    assert L.mi_entries_commit_order(ents, 2, out) == 0 and list(out) == [1, 0]


@pytest.mark.parametrize("shape", ["walk", "random", "reversed"])
@pytest.mark.parametrize("threads", ["1", "2", "3", "5"])
def test_commit_order_at_scale_runs_blocks_and_threads(engine_lib, monkeypatch, shape, threads):
    """300 000 entries (C2 holds 100 000, C4 ten million: SURVEY 8a a7): the order is found by merging the input's
    non-decreasing runs, block-wise on several threads from 131 072 entries on -- the same answer as Python's stable
    sort of the keys for an input in walk order (few runs), shuffled (runs of two) and reversed (runs of one), with
    whiteout markers beside the paths they hide (equal keys: input order) and names on both sides of '/'"""
    import random
    import makisu_amd
    monkeypatch.setenv("MI_WALK_THREADS", threads)
    rng = random.Random(7)
    rels = ["."]
    for d in range(3000):
        dn = "d%04d" % d
        rels.append(dn)
        rels += ["%s/f%05d%s" % (dn, k, rng.choice(["", ".txt", "-x", " y"])) for k in range(96)]
        rels += ["%s.d" % dn, "%s-e" % dn, "%s/.wh.f%05d" % (dn, rng.randrange(96)), "%s/.wh.gone" % dn]
    if shape == "walk":
        rels.sort(key=lambda r: r.split("/"))
    elif shape == "random":
        rng.shuffle(rels)
    else:
        rels.sort(reverse=True)

    def key(r):
        dst = "/" if r == "." else "/" + r
        d, b = dst.rsplit("/", 1)
        return ((d + "/" + b[4:]) if b.startswith(".wh.") else dst).encode()

    assert len(rels) > 2 * 131072
    assert makisu_amd.commit_order(rels) == sorted(range(len(rels)), key=lambda i: key(rels[i]))


# ---- mi_snapshot_diff: createLayerByScan + maybeAddToLayer on two walks ----------------------
def _diff_names(before, after, **kw):
    import makisu_amd
    flags, wh = makisu_amd.snapshot_diff(before, after, **kw)
    changed = sorted(e["relpath"] for e, f in zip(after, flags) if f == makisu_amd.DIFF_CHANGED)
    carried = sorted(e["relpath"] for e, f in zip(after, flags) if f == makisu_amd.DIFF_ANCESTOR)
    whiteouts = sorted(e["relpath"] for e, w in zip(before, wh) if w)
    return changed, carried, whiteouts


def test_snapshot_diff_on_real_trees(tmp_path, engine_lib):
    """Two lstat walks of the same directory before/after edits -- the cases of the reference's
    scan tests (lib/snapshot/mem_fs_test.go: TestCreateLayerByScan :572-686 add / modify / delete,
    TestAddLayerByScanWhiteout :1038-1116 one whiteout per deleted subtree)."""
    import shutil
    import time
    import makisu_amd
    root = tmp_path / "fs"
    os.makedirs(root / "etc" / "conf.d")
    os.makedirs(root / "usr" / "lib" / "deep" / "er")
    os.makedirs(root / "var")
    (root / "etc" / "passwd").write_bytes(b"root:x:0:0\n")
    (root / "etc" / "conf.d" / "a.conf").write_bytes(b"a=1\n")
    (root / "usr" / "lib" / "libx.so").write_bytes(b"\x7fELF" + bytes(100))
    (root / "usr" / "lib" / "deep" / "er" / "f").write_bytes(b"f")
    (root / "var" / "log").write_bytes(b"")
    os.symlink("passwd", root / "etc" / "alias")
    old = int(time.time()) - 7200
    for dp, dns, fns in os.walk(root):
        for n in dns + fns:
            os.utime(os.path.join(dp, n), (old, old), follow_symlinks=False)
    os.utime(root, (old, old))
    before = makisu_amd.tree_walk(str(root), mode=makisu_amd.TREE_SCAN, full=True)
    assert _diff_names(before, before) == ([], [], [])                     # nothing happened

    (root / "etc" / "conf.d" / "b.conf").write_bytes(b"b=2\n")            # added (dir mtime changes too)
    (root / "etc" / "passwd").write_bytes(b"root:x:0:0\nme:x:1:1\n")      # modified: new size
    os.unlink(root / "etc" / "alias")
    os.symlink("conf.d/a.conf", root / "etc" / "alias")                    # symlink retargeted
    shutil.rmtree(root / "usr" / "lib" / "deep")                           # a subtree goes away
    os.unlink(root / "var" / "log")
    os.mkdir(root / "var" / "log")                                         # file replaced by a directory
    for p in (root / "etc", root / "usr" / "lib", root / "var"):           # undo the dir mtime bumps we do not
        os.utime(p, (old, old))                                            # want to test here
    after = makisu_amd.tree_walk(str(root), mode=makisu_amd.TREE_SCAN, full=True)
    changed, carried, whiteouts = _diff_names(before, after)
    assert changed == ["etc/alias", "etc/conf.d", "etc/conf.d/b.conf", "etc/passwd", "var/log"]
    assert carried == ["etc", "usr", "usr/lib", "var"]                     # ancestors of changes AND of the whiteout
    assert whiteouts == ["usr/lib/deep"]                                   # one for the whole subtree
    # ignore_time: the conf.d directory (mtime bumped by the new file) no longer counts
    changed_it, _, _ = _diff_names(before, after, ignore_time=True)
    assert "etc/conf.d" not in changed_it and "etc/conf.d/b.conf" in changed_it
    # a directory replaced by a FILE: the children get no whiteouts (the file overwrites the dir)
    shutil.rmtree(root / "etc" / "conf.d")
    (root / "etc" / "conf.d").write_bytes(b"now a file")
    after2 = makisu_amd.tree_walk(str(root), mode=makisu_amd.TREE_SCAN, full=True)
    changed2, _, whiteouts2 = _diff_names(before, after2)
    assert "etc/conf.d" in changed2 and not any(w.startswith("etc/conf.d") for w in whiteouts2)


def test_snapshot_diff_content_aware(engine_lib):
    """Same size, same second, same owner, different bytes: invisible to the reference's rule
    (compare.go:101-103), caught once both sides carry chunk roots."""
    import hashlib
    f = {"relpath": "app/bin", "size": 5, "mtime_sec": 100, "mode": 0o100755, "kind": 1, "file_index": 0}
    d = {"relpath": "app", "mode": 0o40755, "kind": 0, "mtime_sec": 100}
    top = {"relpath": ".", "mode": 0o40755, "kind": 0, "mtime_sec": 100}
    r1 = np.frombuffer(hashlib.sha256(b"test1").digest(), dtype=np.uint8)
    r2 = np.frombuffer(hashlib.sha256(b"test2").digest(), dtype=np.uint8)
    assert _diff_names([top, d, f], [top, d, f]) == ([], [], [])
    assert _diff_names([top, d, f], [top, d, f], roots_before=r1, roots_after=r1) == ([], [], [])
    assert _diff_names([top, d, f], [top, d, f], roots_before=r1, roots_after=r2) == (["app/bin"], ["app"], [])
    assert _diff_names([top, d, f], [top, d, f], roots_before=r1, roots_after=None) == ([], [], [])
    # deleting everything below the root: whiteouts only for the top-level children
    assert _diff_names([top, d, f], [top]) == ([], [], ["app"])
    # an entry of an unsupported kind on the old side is the reference's "unsupported type" error
    import makisu_amd
    with pytest.raises(makisu_amd.MiError):
        makisu_amd.snapshot_diff([dict(f, kind=9)], [f])


def test_snapshot_diff_mirrors_add_layer_by_scan_whiteout(tmp_path, engine_lib):
    """lib/snapshot/mem_fs_test.go:1038-1116 (TestAddLayerByScanWhiteout): six entries under /test1
    make a 6-entry layer (the root is not added); removing /test1 makes a 1-entry layer: one
    whiteout for the whole subtree."""
    import shutil
    import makisu_amd
    root = tmp_path / "fs"
    os.makedirs(root / "test1" / "test2")
    (root / "test1" / "test2" / "test3.txt").write_bytes(b"hello")
    os.makedirs(root / "test1" / "test4" / "test5")
    (root / "test1" / "test4" / "test5" / "test6.txt").write_bytes(b"hello")
    empty = makisu_amd.tree_walk(str(tmp_path / "fs"), mode=makisu_amd.TREE_SCAN, full=True)[:1]   # just "."
    first = makisu_amd.tree_walk(str(root), mode=makisu_amd.TREE_SCAN, full=True)
    changed, carried, whiteouts = _diff_names(empty, first)
    assert len(changed) == 6 and not carried and not whiteouts              # require.Equal(6, ...count())
    shutil.rmtree(root / "test1")
    second = makisu_amd.tree_walk(str(root), mode=makisu_amd.TREE_SCAN, full=True)
    changed, carried, whiteouts = _diff_names(first, second, ignore_time=True)
    assert (changed, carried, whiteouts) == ([], [], ["test1"])             # require.Equal(1, ...count())


def _layer_members(before, after):
    """The set of tar entry names the diff puts in the layer (whiteouts as dir/.wh.<name>)."""
    changed, carried, whiteouts = _diff_names(before, after)
    wh = [os.path.join(os.path.dirname(w), ".wh." + os.path.basename(w)) for w in whiteouts]
    return sorted(changed + carried + wh)


def test_snapshot_diff_mirrors_create_layer_by_scan(tmp_path, engine_lib):
    """lib/snapshot/mem_fs_test.go:572-686 (TestCreateLayerByScan: Simple, Symlink, Whiteout): the
    layer each scan must produce, given what the previous scan saw."""
    import shutil
    import makisu_amd

    def scan(root):
        return makisu_amd.tree_walk(str(root), mode=makisu_amd.TREE_SCAN, full=True)

    # Simple
    root = tmp_path / "simple"
    os.makedirs(root)
    s0 = scan(root)
    os.makedirs(root / "test1")
    (root / "test1" / "test.txt").write_bytes(b"hello")
    s1 = scan(root)
    assert _layer_members(s0, s1) == ["test1", "test1/test.txt"]
    os.makedirs(root / "test1" / "test2" / "test3")
    s2 = scan(root)
    assert _layer_members(s1, s2) == ["test1", "test1/test2", "test1/test2/test3"]   # not test.txt

    # Symlink
    root = tmp_path / "symlink"
    os.makedirs(root / "test11" / "test12" / "ignore1")
    s1 = scan(root)
    assert _layer_members(scan(tmp_path / "simple")[:1], s1) == ["test11", "test11/test12", "test11/test12/ignore1"]
    os.makedirs(root / "test21" / "test22" / "ignore2")
    os.symlink(str(root / "test11"), root / "test21" / "test22" / "link")
    s2 = scan(root)
    assert _layer_members(s1, s2) == ["test21", "test21/test22", "test21/test22/ignore2", "test21/test22/link"]
    assert [e["link_target"] for e in s2 if e["relpath"].endswith("/link")] == ["/test11"]   # root-trimmed

    # Whiteout
    root = tmp_path / "whiteout"
    os.makedirs(root / "test11" / "test12")
    (root / "test11" / "test12" / "test.txt").write_bytes(b"hello")
    (root / "test11" / "test14.txt").write_bytes(b"hello")
    s1 = scan(root)
    shutil.rmtree(root / "test11" / "test12")
    os.unlink(root / "test11" / "test14.txt")
    s2 = scan(root)
    assert _layer_members(s1, s2) == ["test11", "test11/.wh.test12", "test11/.wh.test14.txt"]


def _walk_in_subprocess(code, env_extra):
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n%s" % (root, code)],
                          env=env, capture_output=True, text=True)


def test_scan_walk_skips_mountpoints_from_the_mounts_table(tmp_path, engine_lib):
    """mountutils.IsMountpoint (lib/mountutils/mountutils.go:54-93) with a swapped mounts file, as
    in mountutils_test.go:25-45: an exact target match is a mountpoint (its subtree is pruned), a
    longer name is not; "/" lines are ignored; the context walk does not look at mounts; a line
    with fewer than four fields fails the walk (:102-116 "Bad /proc/mounts format")."""
    root = tmp_path / "fs"
    os.makedirs(root / "etc" / "hosts")                     # a directory mountpoint with content below
    (root / "etc" / "hosts" / "inside").write_bytes(b"x")
    (root / "etc" / "hosts.txt").write_bytes(b"not a mountpoint")
    (root / "etc" / "hostname").write_bytes(b"file mountpoint")
    mounts = tmp_path / "mounts"
    mounts.write_text("overlay / overlay rw,lowerdir=" + ":".join("/var/lib/l%04d" % i for i in range(2000)) + " 0 0\n"   # one 30 KB line
                      "overlay / overlay rw 0 0\n"
                      "cgroup %s/etc/hostname etx4 ro,nosuid,nodev,noexec,mode=755 0 0\n"
                      "cgroup %s/etc/hosts etx4 ro,nosuid,nodev,noexec,mode=755 0 0\n" % (root, root))
    code = ("import makisu_amd, json\n"
            "print(json.dumps([[g[0] for g in makisu_amd.tree_walk(%r, mode=m)] for m in (makisu_amd.TREE_SCAN, makisu_amd.TREE_CONTEXT)]))"
            % str(root))
    out = _walk_in_subprocess(code, {"MI_MOUNTS_FILE": str(mounts)})
    assert out.returncode == 0, out.stderr
    import json
    scan, context = json.loads(out.stdout.strip().splitlines()[-1])
    assert scan == [".", "etc", "etc/hosts.txt"]
    assert context == [".", "etc", "etc/hostname", "etc/hosts", "etc/hosts/inside", "etc/hosts.txt"]
    # a missing mounts file = no mountpoints (:118-127)
    out = _walk_in_subprocess(code, {"MI_MOUNTS_FILE": str(tmp_path / "absent")})
    assert json.loads(out.stdout.strip().splitlines()[-1])[0] == context
    # bad format: every scan walk fails, the context walk still works
    mounts.write_text("cgroup /etc/hostname\n")
    code_bad = ("import makisu_amd\n"
                "print(len(makisu_amd.tree_walk(%r)))\n"
                "try:\n"
                "    makisu_amd.tree_walk(%r, mode=makisu_amd.TREE_SCAN)\n"
                "except makisu_amd.MiError as e:\n"
                "    print('ERR', e.code)\n" % (str(root), str(root)))
    out = _walk_in_subprocess(code_bad, {"MI_MOUNTS_FILE": str(mounts)})
    assert out.stdout.split() == ["6", "ERR", "-5"], out.stdout + out.stderr


def test_plain_c_host_consumer(tmp_path, engine_lib):
    """tests/cabi/host_driver.c: the host-side entry points (walk, commit order, layer diff,
    header comparison) used from plain C against the header, compared with the Python binding."""
    import shutil
    import subprocess
    import makisu_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_driver")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cabi", "host_driver.c"), "-o", exe,
                           "-L", os.path.join(root, "makisu_amd"), "-lmakisu_mi",
                           "-Wl,-rpath," + os.path.join(root, "makisu_amd")])
    a, b = tmp_path / "before", tmp_path / "after"
    os.makedirs(a / "d" / "sub")
    (a / "d" / "sub" / "f").write_bytes(b"1")
    (a / "d-x").write_bytes(b"2")
    (a / "gone").write_bytes(b"3")
    shutil.copytree(a, b, symlinks=True)
    os.unlink(b / "gone")
    (b / "d" / "sub" / "f").write_bytes(b"12")               # size change
    (b / "d" / "new").write_bytes(b"n")
    out = subprocess.run([exe, str(a), str(b)], check=True, capture_output=True, text=True).stdout.splitlines()
    after = makisu_amd.tree_walk(str(b), mode=makisu_amd.TREE_SCAN, full=True)
    before = makisu_amd.tree_walk(str(a), mode=makisu_amd.TREE_SCAN, full=True)
    rels = [e["relpath"] for e in after]
    assert [l[2:] for l in out if l.startswith("O ")] == [rels[i] for i in makisu_amd.commit_order(rels)]
    flags, wh = makisu_amd.snapshot_diff(before, after, ignore_time=True)
    assert sorted(l[2:] for l in out if l.startswith("C ")) == sorted(
        e["relpath"] for e, f in zip(after, flags) if f == makisu_amd.DIFF_CHANGED) == ["d/new", "d/sub/f"]
    assert sorted(l[2:] for l in out if l.startswith("A ")) == ["d", "d/sub"]
    assert [l[2:] for l in out if l.startswith("W ")] == ["gone"]
    assert out[-1] == "S 1"                                   # the two top directories: same owner/mode, mtime ignored


# ---- the parallel enumeration gives filepath.Walk's sequence ------------------------
def _random_tree(root, rng, n_files):
    """~n_files empty files under a randomized directory tree with everything the walks care about:
    names that sort differently by byte than by locale, symlinks (relative, absolute into the tree,
    dangling), a fifo, AUFS whiteout-meta names, a directory to blacklist, deep chains, empty dirs."""
    alphabet = ["a", "B", "-", ".", "_", "z", "0", "9", "é", "Z"]
    dirs = [root]
    made = 0
    os.makedirs(os.path.join(root, "skipme", "inner"))
    open(os.path.join(root, "skipme", "inner", "never"), "w").close()
    while made < n_files:
        d = dirs[int(rng.integers(0, len(dirs)))]
        depth = d.count(os.sep) - root.count(os.sep)
        name = "".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), int(rng.integers(1, 9))))
        if name in (".", ".."):
            continue
        p = os.path.join(d, name)
        if os.path.lexists(p):
            continue
        r = rng.random()
        if r < 0.06 and depth < 12:
            os.mkdir(p)
            dirs.append(p)
        elif r < 0.08:
            os.symlink("../" + name + ".target", p)               # relative, dangling
        elif r < 0.09:
            os.symlink(os.path.join(root, "skipme"), p)          # absolute, inside the root
        elif r < 0.093:
            os.mkfifo(p)
        elif r < 0.096:
            open(os.path.join(d, ".wh..wh." + name), "w").close()
        else:
            n = int(rng.integers(1, 400))                         # a burst of files in this directory
            for k in range(n):
                open("%s.%d" % (p, k), "w").close()
            made += n
    return made


@pytest.mark.parametrize("mode", ["context", "scan"])
def test_parallel_walk_equals_the_sequential_walk_on_a_random_tree(tmp_path, engine_lib, mode, monkeypatch):
    """200 000 files (MI_WALK_TEST_FILES overrides): the N-thread enumeration + ordered assembly returns the
    entries of the one-thread filepath.Walk restatement -- same paths, same order, same fields -- in both
    walk modes, at several thread counts; and it is what the default configuration runs."""
    import makisu_amd as M
    import shutil
    import tempfile
    import time
    n_files = int(os.environ.get("MI_WALK_TEST_FILES", "200000"))
    base = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path))
    try:
        root = os.path.join(base, "tree")
        os.mkdir(root)
        made = _random_tree(root, np.random.default_rng(11), n_files)
        kw = dict(blacklist=[os.path.join(root, "skipme")], mode=M.TREE_SCAN, full=True) if mode == "scan" else \
            dict(mode=M.TREE_CONTEXT, full=True)
        monkeypatch.setenv("MI_WALK_THREADS", "1")
        t0 = time.perf_counter()
        want = M.tree_walk(root, None, **kw)
        t_seq = time.perf_counter() - t0
        n_reg = sum(1 for e in want if e["kind"] == M.KIND_FILE)
        assert n_reg >= 0.95 * made and len(want) > n_reg        # (bursts may land on the same names)
        if mode == "scan":
            assert not any("skipme" in e["relpath"].split("/") or os.path.basename(e["relpath"]).startswith(".wh..wh.")
                           for e in want)
        assert not any(stat.S_ISFIFO(e["mode"]) for e in want)
        times = {}
        for nt in ("2", "5", "16"):
            monkeypatch.setenv("MI_WALK_THREADS", nt)
            t0 = time.perf_counter()
            got = M.tree_walk(root, None, **kw)
            times[nt] = time.perf_counter() - t0
            assert len(got) == len(want)
            assert got == want, next((i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b)
        monkeypatch.delenv("MI_WALK_THREADS")
        assert M.tree_walk(root, None, **kw) == want             # the default thread count
        print("walk of %d entries: 1 thread %.3f s, %s" % (len(want), t_seq, {k: round(v, 3) for k, v in times.items()}))
    finally:
        shutil.rmtree(base, ignore_errors=True)


def test_parallel_walk_reports_the_first_error_in_walk_order(tmp_path, engine_lib, monkeypatch):
    """Two absolute symlinks that point outside the scan root, in different directories: a sequential
    walk fails at the first one it reaches; the parallel one may have SEEN both -- it must still name
    the first in filepath.Walk order, and list nothing."""
    import makisu_amd as M
    root = tmp_path / "r"
    for d in ("a/deep/er", "b", "c/x"):
        os.makedirs(root / d)
    for i in range(300):
        (root / "a" / "deep" / ("f%03d" % i)).write_bytes(b"")
    os.symlink("/outside/one", root / "b" / "bad1")
    os.symlink("/outside/two", root / "a" / "deep" / "er" / "bad0")      # a/... sorts before b/...
    msgs = []
    for nt in ("1", "8"):
        monkeypatch.setenv("MI_WALK_THREADS", nt)
        with pytest.raises(M.MiError) as ei:
            M.tree_walk(str(root), None, mode=M.TREE_SCAN, full=True)
        assert ei.value.code == -1
        msgs.append(str(ei.value))
    assert msgs[0] == msgs[1]


def test_compare_fs_case_through_the_snapshot_diff(engine_lib):
    """lib/snapshot/mem_fs_test.go:1198-1289 (TestCompareFS: two trees sharing /common; /common/test1 only in the first,
    /common/test2 only in the second, /common/world with another mode in each): compareNode's three sets are what
    mi_snapshot_diff reports between the two entry lists -- gone from `after` (a whiteout), new or different in `after`
    (changed) -- in either direction."""
    import makisu_amd
    D = lambda p: {"relpath": p, "kind": makisu_amd.KIND_DIR, "mode": 0o40755, "mtime_sec": 5}                      # noqa: E731
    F = lambda p, m: {"relpath": p, "kind": makisu_amd.KIND_FILE, "mode": 0o100000 | m, "size": 5, "mtime_sec": 5}  # noqa: E731
    fs1 = makisu_amd.apply_layer([], [D("/common"), D("/common/test1"), F("/common/world", 0o711)])
    fs2 = makisu_amd.apply_layer([], [D("/common"), D("/common/test2"), F("/common/world", 0o755)])
    changed, carried, whiteouts = _diff_names(fs1, fs2)
    assert whiteouts == ["common/test1"]                                    # missing2: only the first tree has it
    assert changed == ["common/test2", "common/world"]                      # missing1 + diff
    assert carried == ["common"]                                            # the unchanged parent travels with them
    changed, carried, whiteouts = _diff_names(fs2, fs1)
    assert whiteouts == ["common/test2"] and changed == ["common/test1", "common/world"]


def test_blacklist_spelled_uncleanly_and_the_prefix_trap(tmp_path):
    """pathutils.IsDescendantOfAny (lib/pathutils/path.go:24-35) cleans both sides (AbsPath) and compares ELEMENTS: a blacklist
    entry spelled with "//", "/./" or a trailing "/" skips what its clean form skips, and "/r/pre" does not skip "/r/prefix".
    (The walks clean the blacklist once and compare clean paths as they stand; an unclean ROOT takes the general way.)"""
    import makisu_amd as M
    root = str(tmp_path / "r")
    for rel in ("skip/a", "skip/deep/b", "other/x/c", "other/y", "pre/z", "prefix/keep", "keep"):
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        open(p, "w").write("1")
    bl = [root + "/skip/", root + "//other/./x", root + "/pre"]
    want = [".", "keep", "other", "other/y", "prefix", "prefix/keep"]
    for r in (root, root + "/", root + "//", os.path.join(str(tmp_path), ".", "r")):
        for threads in ("1", "4"):
            code = ("import sys, makisu_amd as M; print([e['relpath'] for e in M.tree_walk(%r, %r, %r, M.TREE_SCAN, full=True)])" % (r, root, bl))
            import subprocess
            import sys
            out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MI_WALK_THREADS=threads, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                                 capture_output=True, text=True, timeout=120)
            assert out.returncode == 0, out.stderr[-800:]
            assert eval(out.stdout.strip()) == want, (r, threads, out.stdout)
