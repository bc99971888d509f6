"""CPU tests of mi_snapshot_copy_ops: MemFS.AddLayerByCopyOps on entry lists.

The cases of the reference's TestCreateLayerByCopy (lib/snapshot/mem_fs_test.go:688-1036) replayed
on real trees -- where the reference asserts findNode(fs, dst).src == <source path>, these assert
that the layer holds dst with that source -- plus addAncestors' symlink cases from TestGetAncestors
(:340-570) and the isUpdated filter (a copy onto an identical header adds nothing).
"""
import os

import pytest

import makisu_amd as M

UID, GID = 1234, 5678            # validChown in the reference's tests


def _tree(root, spec):
    """spec: list of (path, kind, content); builds it under root, returns the walk's entries."""
    for p, kind, content in spec:
        full = root / p.lstrip("/")
        if kind == "d":
            full.mkdir(parents=True, exist_ok=True)
        elif kind == "f":
            full.parent.mkdir(parents=True, exist_ok=True)
            full.write_text(content)
            full.chmod(0o755)
        else:
            full.parent.mkdir(parents=True, exist_ok=True)
            os.symlink(content, full)
    ents = M.tree_walk(str(root), None, (), M.TREE_SCAN, full=True)
    return [e for e in ents if e["relpath"] != "."]


def _op(root, srcs, dst):
    return {"src_root": str(root), "srcs": [s.lstrip("/") for s in srcs], "dst": dst, "uid": UID, "gid": GID}


def _by_dst(layer):
    return {"/" + e["relpath"]: e for e in layer}


def test_file_to_dir_file(tmp_path):                         # "file dir/file"
    tree = _tree(tmp_path, [("/test1", "d", ""), ("/test1/test.txt", "f", "hello")])
    layer = M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/test1/test.txt"], "/test2/test.txt")], now_sec=77)
    got = _by_dst(layer)
    assert set(got) == {"/test2", "/test2/test.txt"}
    assert got["/test2/test.txt"]["src"] == str(tmp_path / "test1/test.txt")
    assert (got["/test2/test.txt"]["uid"], got["/test2/test.txt"]["gid"]) == (UID, GID)
    # the missing parent: created with default owner 0/0 (maybeAddToLayer's addAncestors(..., 0, 0, 0)),
    # the clock's time, and no source
    d = got["/test2"]
    assert d["kind"] == M.KIND_DIR and d["src"] == "/" and d["mtime_sec"] == 77 and (d["uid"], d["gid"]) == (0, 0)
    assert [e["relpath"] for e in layer] == ["test2", "test2/test.txt"]          # commit order


def test_file_to_dir_slash(tmp_path):                        # "file dir/"
    tree = _tree(tmp_path, [("/test1", "d", ""), ("/test1/test.txt", "f", "hello")])
    got = _by_dst(M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/test1/test.txt"], "/dst/")], now_sec=5))
    assert set(got) == {"/dst", "/dst/test.txt"}
    assert got["/dst/test.txt"]["src"] == str(tmp_path / "test1/test.txt")
    # a single non-directory source never runs the createDst branch (mem_fs.go:357-366): /dst is only
    # created as the file's missing ancestor, owner 0/0
    assert (got["/dst"]["uid"], got["/dst"]["gid"]) == (0, 0)


def test_file_file_to_dir(tmp_path):                         # "file file dir/"
    tree = _tree(tmp_path, [("/test1", "d", ""), ("/test1/test2.txt", "f", "hello"), ("/test1/test3.txt", "f", "hello")])
    got = _by_dst(M.copy_ops_layer(tree, str(tmp_path),
                                   [_op(tmp_path, ["/test1/test2.txt", "/test1/test3.txt"], "/dst/")]))
    assert got["/dst/test2.txt"]["src"] == str(tmp_path / "test1/test2.txt")
    assert got["/dst/test3.txt"]["src"] == str(tmp_path / "test1/test3.txt")
    assert got["/dst"]["kind"] == M.KIND_DIR
    assert (got["/dst"]["uid"], got["/dst"]["gid"]) == (UID, GID)                # createDst: the op's owner


def test_dir_to_dir_copies_contents(tmp_path):               # "dir dir/"
    tree = _tree(tmp_path, [("/test1/test2", "d", ""), ("/test1/test2/test.txt", "f", "hello")])
    got = _by_dst(M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/test1/test2"], "/dst/")]))
    assert set(got) == {"/dst", "/dst/test.txt"}             # the directory itself is not copied
    assert got["/dst/test.txt"]["src"] == str(tmp_path / "test1/test2/test.txt")


def test_dir_dir_and_file_dir_to_dir(tmp_path):              # "dir dir dir/", "file dir dir/"
    spec = [("/test1/test2", "d", ""), ("/test1/test2/test3.txt", "f", "hello"), ("/test1/test4/test5", "d", ""),
            ("/test1/test4/test5/test6.txt", "f", "hello")]
    tree = _tree(tmp_path, spec)
    for srcs in (["/test1/test2", "/test1/test4"], ["/test1/test2/test3.txt", "/test1/test4"]):
        got = _by_dst(M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, srcs, "/dst/")]))
        assert set(got) == {"/dst", "/dst/test3.txt", "/dst/test5", "/dst/test5/test6.txt"}
        assert got["/dst/test3.txt"]["src"] == str(tmp_path / "test1/test2/test3.txt")
        assert got["/dst/test5"]["src"] == str(tmp_path / "test1/test4/test5")
        assert got["/dst/test5/test6.txt"]["src"] == str(tmp_path / "test1/test4/test5/test6.txt")


def test_workdir_relative_destination(tmp_path):             # "workdir": resolveDestination("/wrk", "dst/")
    spec = [("/test1/test2", "d", ""), ("/test1/test2/test3.txt", "f", "hello"), ("/test1/test4/test5", "d", ""),
            ("/test1/test4/test5/test6.txt", "f", "hello")]
    tree = _tree(tmp_path, spec)
    got = _by_dst(M.copy_ops_layer(tree, str(tmp_path),
                                   [_op(tmp_path, ["/test1/test2/test3.txt", "/test1/test4"], "/wrk/dst/")]))
    assert {"/wrk", "/wrk/dst", "/wrk/dst/test3.txt", "/wrk/dst/test5", "/wrk/dst/test5/test6.txt"} == set(got)


def test_unchanged_copy_adds_nothing_and_existing_ancestors_are_carried(tmp_path):
    tree = _tree(tmp_path, [("/app", "d", ""), ("/app/conf", "d", ""), ("/app/conf/a.txt", "f", "same"),
                            ("/src/a.txt", "f", "same"), ("/src/b.txt", "f", "new")])
    # make /src/a.txt's header identical to /app/conf/a.txt's: same size, mode, mtime; the op's owner
    # equals the tree entry's owner (the test runs as this uid/gid)
    st = os.lstat(tmp_path / "app/conf/a.txt")
    os.utime(tmp_path / "src/a.txt", (st.st_mtime, st.st_mtime))
    tree = M.tree_walk(str(tmp_path), None, (), M.TREE_SCAN, full=True)[1:]
    op = {"src_root": str(tmp_path), "srcs": ["src"], "dst": "/app/conf/", "uid": st.st_uid, "gid": st.st_gid}
    got = _by_dst(M.copy_ops_layer(tree, str(tmp_path), [op]))
    assert "/app/conf/a.txt" not in got                      # isUpdated: similar header, not added
    assert got["/app/conf/b.txt"]["src"] == str(tmp_path / "src/b.txt")
    # the existing directories on the way are part of the layer, with their own source paths
    assert got["/app"]["src"] == str(tmp_path / "app") and got["/app/conf"]["src"] == str(tmp_path / "app/conf")


def test_destination_through_a_symlink(tmp_path):
    """TestGetAncestors FollowSymlinkFullResolve: /lnk -> /real; copying into /lnk/sub/ carries the
    link, then creates below its target."""
    tree = _tree(tmp_path, [("/real", "d", ""), ("/lnk", "l", str(tmp_path / "real")), ("/src/f", "f", "x")])
    assert [e for e in tree if e["relpath"] == "lnk"][0]["link_target"] == "/real"     # root-trimmed by the walk
    got = _by_dst(M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/src"], "/lnk/sub/")]))
    assert "/lnk" in got and got["/lnk"]["kind"] == M.KIND_SYMLINK
    assert "/real" in got and "/real/sub" in got and got["/real/sub"]["kind"] == M.KIND_DIR
    assert got["/real/sub/f"]["src"] == str(tmp_path / "src/f")                   # dst resolved through the link
    # a single FILE source skips the createDst branch: its destination stays spelled through the
    # link and updateMemFS refuses it ("missing intermediate directory"), as in the reference
    with pytest.raises(M.MiError) as ei:
        M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/src/f"], "/lnk/sub/")])
    assert "missing intermediate directory" in str(ei.value)


def test_file_copied_to_a_name_right_below_a_symlink(tmp_path):
    """`COPY f /lib/f` on an image whose /lib is a link to usr/lib (single file source: no createDst, the destination
    stays spelled through the link).  maybeAddToLayer -> addAncestors carries the link and the existing ancestors of
    its target (mem_fs.go:531-548), then contentMemFile.updateMemFS descends into the LINK's node -- any node will
    do, mem_layer.go:57-68 -- and hangs the file below it: the layer holds lib, lib/f and usr, usr/lib."""
    tree = _tree(tmp_path, [("/usr/lib", "d", ""), ("/lib", "l", str(tmp_path / "usr/lib")), ("/src/f", "f", "x")])
    got = _by_dst(M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/src/f"], "/lib/f")]))
    assert sorted(got) == ["/lib", "/lib/f", "/usr", "/usr/lib"]
    assert got["/lib"]["kind"] == M.KIND_SYMLINK and got["/lib/f"]["src"] == str(tmp_path / "src/f")
    # a second file below the link in the same layer: the link is re-added on the way and comes back WITHOUT the child
    # it had in the tree (updateMemFS lets only directories keep children) -- the layer still holds both files
    got = _by_dst(M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/src/f"], "/lib/f"),
                                                          _op(tmp_path, ["/src/f"], "/lib/g")]))
    assert sorted(got) == ["/lib", "/lib/f", "/lib/g", "/usr", "/usr/lib"]


def test_a_source_named_like_a_whiteout_is_filed_under_the_path_it_deletes(tmp_path):
    """memLayer.addHeader (mem_layer.go:197-212): a dst whose base name starts with ".wh." becomes a whiteoutMemFile
    keyed by the DELETED path, so rangeFiles' sort.Strings (:232-244) places it there -- and it removes that path
    from the tree, so a later copy of the same file is new again."""
    tree = _tree(tmp_path, [("/dst", "d", ""), ("/dst/zz", "f", "old"), ("/src/a", "f", "1"), ("/src/.wh.zz", "f", ""),
                            ("/src/m", "f", "2"), ("/again/zz", "f", "old")])
    layer = M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/src"], "/dst/")])
    assert [e["relpath"] for e in layer] == ["dst", "dst/a", "dst/m", "dst/.wh.zz"]
    assert [e["relpath"] for e in layer] == [layer[k]["relpath"] for k in M.commit_order([e["relpath"] for e in layer])]
    st = os.lstat(tmp_path / "dst/zz")
    os.utime(tmp_path / "again/zz", (st.st_mtime, st.st_mtime))
    own = {"uid": st.st_uid, "gid": st.st_gid}
    same = dict(_op(tmp_path, ["/again/zz"], "/dst/zz"), **own)
    assert M.copy_ops_layer(tree, str(tmp_path), [same]) == []                        # similar header: nothing to add
    both = M.copy_ops_layer(tree, str(tmp_path), [dict(_op(tmp_path, ["/src/.wh.zz"], "/dst/"), **own), same])
    assert "dst/zz" in [e["relpath"] for e in both]        # (the marker and the file share one key: the file, added last, stays)


def test_symlink_loop_and_errors(tmp_path):
    tree = _tree(tmp_path, [("/a", "l", str(tmp_path / "a" / "b")), ("/src/f", "f", "x")])   # /a -> /a/b -> ...
    with pytest.raises(M.MiError) as ei:
        M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/src/f"], "/a/c/")])
    assert "symlink loop" in str(ei.value)
    with pytest.raises(M.MiError) as ei:                     # stat src fails
        M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/nope"], "/dst/")])
    assert ei.value.code == -5
    os.symlink("/etc", tmp_path / "src" / "out")             # a source symlink leaving the root
    with pytest.raises(M.MiError) as ei:
        M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, ["/src/out/passwd", "/src/f"], "/dst/")])
    assert "outside of root" in str(ei.value)


def test_copy_layer_feeds_the_layer_writer(tmp_path):
    """The COPY step end to end on the host side: copy-ops layer -> mi_layer_* tar -> read back."""
    import io
    import tarfile
    tree = _tree(tmp_path / "ctx", [("/app/main.py", "f", "print(1)\n"), ("/app/lib/util.py", "f", "x = 2\n")])
    layer = M.copy_ops_layer([], str(tmp_path / "ctx"), [_op(tmp_path / "ctx", ["/app"], "/srv/app/")], now_sec=9)
    out = tmp_path / "layer.tar"
    fd = os.open(out, os.O_WRONLY | os.O_CREAT, 0o644)
    with M.Layer(out_fd=fd, gzip_level=M.GZIP_OFF) as lw:
        for e in layer:
            lw.add(e, e["src"] if e["kind"] == M.KIND_FILE else None)
        pair = lw.finish()
    os.close(fd)
    with tarfile.open(out) as tf:
        names = tf.getnames()
        assert names == ["srv", "srv/app", "srv/app/lib", "srv/app/lib/util.py", "srv/app/main.py"]
        assert tf.extractfile("srv/app/main.py").read() == b"print(1)\n"
        assert tf.getmember("srv/app/main.py").uid == UID
    assert pair["n_entries"] == 5


def test_new_copy_operation_checks_replayed(tmp_path):
    """lib/snapshot/copy_op_test.go:36-72 (TestNewCopyOperation: four parameter sets NewCopyOperation must refuse) and
    resolveDestination (copy_op.go:149-159), through mi_copy_op_resolve; mi_snapshot_copy_ops refuses the same on the
    resolved dst."""
    import makisu_amd as M
    import pytest
    for n_srcs, work_dir, dst in ((0, "", "/test2/test.txt"),            # srcs cannot be empty
                                  (2, "", "/target/test"),               # several sources, dst not in directory format
                                  (2, "", "target/test"),                # ... and relative
                                  (2, "wrk/", "target/test/")):          # relative dst, work dir not absolute
        with pytest.raises(M.MiError) as ei:
            M.copy_op_resolve(n_srcs, work_dir, dst)
        assert "check copy param" in str(ei.value)
    assert M.copy_op_resolve(1, "", "/test2/test.txt") == "/test2/test.txt"
    assert M.copy_op_resolve(1, "/work", "test2/test.txt") == "/work/test2/test.txt"
    assert M.copy_op_resolve(2, "/work", "sub/") == "/work/sub/"          # the trailing "/" is preserved
    assert M.copy_op_resolve(2, "/work/", ".") == "/work/" and M.copy_op_resolve(1, "/work/a", "..") == "/work/"
    assert M.copy_op_resolve(1, "/work", "a//b/../c") == "/work/a/c"      # filepath.Join cleans
    # the layer builder itself
    src = tmp_path / "src"
    (src / "dir").mkdir(parents=True)
    (src / "file").write_bytes(b"x")
    (src / "dir" / "a").write_bytes(b"y")
    root = tmp_path / "root"
    root.mkdir()
    tree = M.tree_walk(str(root), mode=M.TREE_SCAN, full=True)
    op = {"src_root": str(src), "srcs": ["file", "dir/"], "dst": "/target/test"}
    with pytest.raises(M.MiError) as ei:
        M.copy_ops_layer(tree, str(root), [op])
    assert "destination must end with" in str(ei.value)
    with pytest.raises(M.MiError) as ei:
        M.copy_ops_layer(tree, str(root), [dict(op, srcs=[])])
    assert "srcs cannot be empty" in str(ei.value)
    with pytest.raises(M.MiError):
        M.copy_ops_layer(tree, str(root), [dict(op, dst="target/test/")])           # not absolute: resolve it first
    got = M.copy_ops_layer(tree, str(root), [dict(op, dst="/target/test/")])
    assert [e["relpath"] for e in got] == ["target", "target/test", "target/test/a", "target/test/file"]


def test_copy_layer_equals_scan_layer_of_the_same_copy(tmp_path):
    """The intent of lib/snapshot/mem_fs_test.go:1118-1197 (TestAddLayersEqual: the layer of a COPY built from copy ops,
    the layer a scan finds after the same copy was really made, and the layer committed from the entries themselves are
    one and the same tarball): here the copy-op layer (mi_snapshot_copy_ops) and the scan layer (walk + mi_snapshot_diff)
    of the same copy, both framed by mi_layer, must be byte-identical."""
    import hashlib
    import shutil
    import makisu_amd as M
    now = 1_600_000_000
    src = tmp_path / "srcroot"
    (src / "test1" / "test2").mkdir(parents=True)
    (src / "test1" / "test4" / "test5").mkdir(parents=True)
    (src / "test1" / "test2" / "test3.txt").write_bytes(b"hello")
    (src / "test1" / "test4" / "test5" / "test6.txt").write_bytes(b"hello")
    for p, _, fs in os.walk(src):
        for n in fs:
            os.utime(os.path.join(p, n), (now - 50, now - 50))
        os.utime(p, (now - 60, now - 60))
    root = tmp_path / "root"
    root.mkdir()
    before = M.tree_walk(str(root), mode=M.TREE_SCAN, full=True)
    uid, gid = os.getuid(), os.getgid()
    dst = M.copy_op_resolve(2, "/wrk", "dst/")
    assert dst == "/wrk/dst/"
    lay1 = M.copy_ops_layer(before, str(root), [{"src_root": str(src), "srcs": ["/test1/test2/test3.txt", "/test1/test4"],
                                                 "dst": dst, "uid": uid, "gid": gid}], now_sec=now)
    assert [e["relpath"] for e in lay1] == ["wrk", "wrk/dst", "wrk/dst/test3.txt", "wrk/dst/test5", "wrk/dst/test5/test6.txt"]

    def frame(entries, src_of, name):
        out = tmp_path / name
        fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            with M.Layer(out_fd=fd, gzip_level=M.GZIP_OFF) as layer:
                for e in entries:
                    layer.add(e, src_of(e) if e["kind"] == M.KIND_FILE else None)
                pair = layer.finish()
        finally:
            os.close(fd)
        return out.read_bytes(), pair

    raw1, pair1 = frame(lay1, lambda e: e["src"], "copy.tar")
    # really make the copy, with the attributes the copy would give (directories the step creates: now, 0755)
    os.makedirs(root / "wrk" / "dst" / "test5")
    shutil.copy2(src / "test1" / "test2" / "test3.txt", root / "wrk" / "dst" / "test3.txt")
    shutil.copy2(src / "test1" / "test4" / "test5" / "test6.txt", root / "wrk" / "dst" / "test5" / "test6.txt")
    for rel in ("wrk", "wrk/dst"):
        os.chmod(root / rel, 0o755)
        os.utime(root / rel, (now, now))
    st5 = os.lstat(src / "test1" / "test4" / "test5")
    os.chmod(root / "wrk" / "dst" / "test5", st5.st_mode & 0o7777)
    os.utime(root / "wrk" / "dst" / "test5", (st5.st_mtime, st5.st_mtime))
    after = M.tree_walk(str(root), mode=M.TREE_SCAN, full=True)
    flags, wh = M.snapshot_diff(before, after)
    lay2 = [e for e, f in zip(after, flags) if f != M.DIFF_SAME and e["relpath"] not in (".", "")]
    assert not any(wh) and [e["relpath"] for e in lay2] == [e["relpath"] for e in lay1]
    raw2, pair2 = frame(lay2, lambda e: os.path.join(str(root), e["relpath"]), "scan.tar")
    assert raw1 == raw2 and pair1["tar_sha256"] == pair2["tar_sha256"] == hashlib.sha256(raw1).digest()


def test_eval_symlinks_cases_replayed(tmp_path):
    """lib/snapshot/utils_test.go:86-147 (TestEvalSymlink: no_symlinks, simple_case, layered): a copy's source is resolved
    through the symlinks inside the source root -- relative ones, absolute ones that point into the root, a linked
    directory in the middle of the path -- and the layer reads the bytes from where the chain ends."""
    r = tmp_path / "r"
    (r / "dir1").mkdir(parents=True)
    (r / "dir1" / "tmp1").write_text("one")
    (r / "test1").write_text("t1")
    os.symlink("test1", r / "link2")                         # relative
    os.symlink(str(r / "link2"), r / "link3")                # absolute, inside the root, to another link
    (r / "dir2").mkdir()
    os.symlink(str(r / "dir1"), r / "dir2" / "dir3")         # a linked directory on the way
    def src_of(path):                                        # noqa: E306
        layer = M.copy_ops_layer([], str(tmp_path / "fsroot"), [_op(r, [path], "/dst/x")])
        return _by_dst(layer)["/dst/x"]["src"]
    (tmp_path / "fsroot").mkdir()
    assert src_of("dir1/tmp1") == str(r / "dir1" / "tmp1")
    assert src_of("link2") == str(r / "test1") and src_of("link3") == str(r / "test1")
    assert src_of("dir2/dir3/tmp1") == str(r / "dir1" / "tmp1")


def test_source_errors_of_eval_symlinks(tmp_path):
    """evalSymlinks (lib/snapshot/utils.go:249-327): a cycle of links ends at "too many links" (walkLink: more than 255
    walked), a source that is not there at "walk link: lstat", a link out of the root at "points outside of root".  A
    SINGLE source is os.Stat'ed first (mem_fs.go:358-360): the kernel's own verdict comes before any of these."""
    tree = _tree(tmp_path, [("/d", "d", ""), ("/d/f", "f", "x"), ("/a", "l", "b"), ("/b", "l", "a")])
    os.symlink("/etc", tmp_path / "out")                      # (after the walk: a scan refuses such a link itself)
    for src, words in (("/a", "too many links"), ("/missing", "lstat"), ("/d/missing/deeper", "lstat"),
                       ("/out", "outside of root")):
        with pytest.raises(M.MiError) as ei:
            M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, [src, "/d/f"], "/dst/")])
        assert words in str(ei.value), (src, str(ei.value))
        if src != "/out":
            with pytest.raises(M.MiError) as ei:
                M.copy_ops_layer(tree, str(tmp_path), [_op(tmp_path, [src], "/dst/")])
            assert "stat src" in str(ei.value), (src, str(ei.value))
    # 255 links in a row are still walked (beyond the kernel's own limit of 40 nothing can be stat'ed: two sources) ...
    chain = [("/c0", "f", "end"), ("/other", "f", "o")] + [("/c%d" % i, "l", "c%d" % (i - 1)) for i in range(1, 256)]
    tree = _tree(tmp_path / "chain", chain)
    got = _by_dst(M.copy_ops_layer(tree, str(tmp_path / "chain"), [_op(tmp_path / "chain", ["/c255", "/other"], "/dst/")]))
    assert got["/dst/c0"]["src"] == str(tmp_path / "chain" / "c0")
    os.symlink("c255", tmp_path / "chain" / "c256")            # ... the 256th is one too many
    with pytest.raises(M.MiError) as ei:
        M.copy_ops_layer(tree, str(tmp_path / "chain"), [_op(tmp_path / "chain", ["/c256", "/other"], "/dst/")])
    assert "too many links" in str(ei.value)
