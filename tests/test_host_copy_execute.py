"""CPU tests of mi_copy_op_execute: CopyOperation.Execute (lib/snapshot/copy_op.go:83-147) over fileio.Copier
(lib/fileio/copy.go) -- the on-disk copy of a COPY/ADD step when the build modifies the file system.

Replayed: TestExecuteCopyOperation's six shapes (copy_op_test.go:73-250) and lib/fileio/copy_test.go (dangling symlink,
target missing / empty / overwritten, special bits, directory onto a missing and onto an existing target, a symlink inside,
the source containing the destination); then the four owner rules of the Copier's header comment, and: after the copy the
scan layer of the root holds what the copy-op layer of the same step holds (TestAddLayersEqual's intent, end to end)."""
import os
import stat

import pytest

import makisu_amd as M

pytestmark = pytest.mark.skipif(os.geteuid() != 0, reason="the copier chowns: needs root, like the reference's")
HELLO, HELLO2 = b"hello", b"hello2"


def _op(src_root, srcs, dst, uid=0, gid=0):
    return {"src_root": str(src_root), "srcs": list(srcs), "dst": dst, "uid": uid, "gid": gid}


def test_execute_copy_operation_replayed(tmp_path):
    src, work = tmp_path / "src", tmp_path / "work"
    (src / "test1" / "test2").mkdir(parents=True)
    work.mkdir()
    (src / "test.txt").write_bytes(HELLO)
    (src / "test2.txt").write_bytes(HELLO2)
    (src / "test1" / "test2" / "test3.txt").write_bytes(HELLO)
    (src / "test1" / "test4").mkdir()
    (src / "test1" / "test4" / "test5.txt").write_bytes(HELLO2)
    # absolute file to absolute file
    dst = M.copy_op_resolve(1, "", str(work / "a" / "test2" / "test.txt"))
    M.copy_op_execute(_op(src, ["/test.txt"], dst, 1234, 5678), chown=True)
    assert (work / "a" / "test2" / "test.txt").read_bytes() == HELLO
    st = os.lstat(work / "a" / "test2" / "test.txt")
    assert (st.st_uid, st.st_gid) == (1234, 5678)
    assert os.lstat(work / "a").st_uid == 0 and os.lstat(work / "a" / "test2").st_uid == 1234   # ancestors root, the dst dir --chown
    # absolute file to relative file
    M.copy_op_execute(_op(src, ["/test.txt"], M.copy_op_resolve(1, str(work / "b"), "test2/test.txt")), chown=True)
    assert (work / "b" / "test2" / "test.txt").read_bytes() == HELLO
    # absolute files to absolute dir / to relative dir "."
    M.copy_op_execute(_op(src, ["/test.txt", "/test2.txt"], M.copy_op_resolve(2, str(work / "c"), "test2/")), chown=True)
    assert (work / "c" / "test2" / "test.txt").read_bytes() == HELLO and (work / "c" / "test2" / "test2.txt").read_bytes() == HELLO2
    M.copy_op_execute(_op(src, ["/test.txt", "/test2.txt"], M.copy_op_resolve(2, str(work / "d" / "test2"), ".")), chown=True)
    assert (work / "d" / "test2" / "test.txt").read_bytes() == HELLO and (work / "d" / "test2" / "test2.txt").read_bytes() == HELLO2
    # absolute dirs to relative dir: CONTENTS of the directories
    M.copy_op_execute(_op(src, ["/test1/test2", "/test1/test4"], M.copy_op_resolve(2, str(work / "e"), "dst/")), chown=True)
    assert (work / "e" / "dst" / "test3.txt").read_bytes() == HELLO and (work / "e" / "dst" / "test5.txt").read_bytes() == HELLO2
    # absolute dir and file to relative dir
    M.copy_op_execute(_op(src, ["/test1/test2", "/test2.txt"], M.copy_op_resolve(2, str(work / "f"), "dst/")), chown=True)
    assert (work / "f" / "dst" / "test3.txt").read_bytes() == HELLO and (work / "f" / "dst" / "test2.txt").read_bytes() == HELLO2
    with pytest.raises(M.MiError) as ei:
        M.copy_op_execute(_op(src, ["/test.txt"], str(work / "x")), chown=True, preserve_owner=True)
    assert "both chown and archive are true" in str(ei.value)
    with pytest.raises(M.MiError) as ei:
        M.copy_op_execute(_op(src, ["/nope"], str(work / "x")))
    assert ei.value.code == -5


def test_fileio_copy_file_cases_replayed(tmp_path):
    s, t = tmp_path / "s", tmp_path / "t"
    s.mkdir()
    t.mkdir()
    # TestCopyFileDanglingSymlink: the link is copied as a link, over what was there.  (Below a directory: as the SOURCE of
    # an op a dangling link already fails Execute's evalSymlinks, in the reference as here.)
    (s / "ld").mkdir()
    os.symlink("/nonexistent", s / "ld" / "link")
    (t / "ld").mkdir()
    (t / "ld" / "link").write_bytes(b"was a file")
    M.copy_op_execute(_op("/", [str(s / "ld")], str(t / "ld") + "/"), internal=True)
    assert os.readlink(t / "ld" / "link") == "/nonexistent"
    with pytest.raises(M.MiError) as ei:
        M.copy_op_execute(_op("/", [str(s / "ld" / "link")], str(t / "x")), internal=True)
    assert "eval symlinks for" in str(ei.value)
    (s / "f").write_bytes(b"Testing COPY")                                   # TestCopyFileTargetNotExist
    M.copy_op_execute(_op("/", [str(s / "f")], str(t / "new")), internal=True)
    assert (t / "new").read_bytes() == b"Testing COPY"
    os.chmod(s / "f", 0o777 | stat.S_ISUID)                                  # TestCopyFileSetSpecialBit
    M.copy_op_execute(_op("/", [str(s / "f")], str(t / "suid")), internal=True)
    assert os.stat(t / "suid").st_mode & 0o7777 == 0o777 | stat.S_ISUID
    (t / "empty").write_bytes(b"")                                           # TestCopyFileTargetEmpty
    M.copy_op_execute(_op("/", [str(s / "f")], str(t / "empty")), internal=True)
    assert (t / "empty").read_bytes() == b"Testing COPY"
    (t / "longer").write_bytes(b"Testing COPY target, which is longer")       # TestCopyFileTargetOverwrite: truncated
    os.chmod(t / "longer", 0o400)
    M.copy_op_execute(_op("/", [str(s / "f")], str(t / "longer")), internal=True)
    assert (t / "longer").read_bytes() == b"Testing COPY"
    os.mkfifo(s / "pipe")                                                    # a special file is skipped, silently
    M.copy_op_execute(_op("/", [str(s / "pipe")], str(t / "pipe")), internal=True)
    assert not os.path.lexists(t / "pipe")


def test_fileio_copy_directory_cases_replayed(tmp_path):
    s = tmp_path / "source"
    (s / "sub1").mkdir(parents=True)
    (s / "sub2").mkdir()
    (s / "sub1" / "one").write_bytes(b"Test source file one")
    (s / "two").write_bytes(b"Test source file two")
    os.symlink(str(s / "sub1"), s / "link")                                  # TestCopyDirectoryIncludingSymlink
    os.chmod(s / "sub1", 0o750)
    t1 = tmp_path / "t1"                                                     # TestCopyDirectoryTargetNotExist
    M.copy_op_execute(_op("/", [str(s)], str(t1) + "/"), internal=True)
    assert (t1 / "sub1" / "one").read_bytes() == b"Test source file one" and (t1 / "two").read_bytes() == b"Test source file two"
    assert os.path.isdir(t1 / "sub2") and os.readlink(t1 / "link") == str(s / "sub1")
    assert os.lstat(t1 / "sub1").st_mode & 0o7777 == 0o750 and os.lstat(t1).st_mode & 0o7777 == 0o755
    t2 = tmp_path / "t2"                                                     # TestCopyDirectoryTargetExists
    t2.mkdir()
    os.chmod(t2, 0o700)
    (t2 / "mine").write_bytes(b"Test target file one")
    M.copy_op_execute(_op("/", [str(s)], str(t2) + "/"), internal=True)
    assert (t2 / "mine").read_bytes() == b"Test target file one" and (t2 / "two").read_bytes() == b"Test source file two"
    assert os.lstat(t2).st_mode & 0o7777 == 0o700                            # an existing target keeps its permissions
    inner = s / "target-inside"                                              # TestCopyDirectoryInfiniteLoop
    inner.mkdir()
    M.copy_op_execute(_op("/", [str(s)], str(inner) + "/"), internal=True)
    assert (inner / "sub1" / "one").read_bytes() == b"Test source file one" and (inner / "two").exists()
    assert not os.path.lexists(inner / "target-inside")                      # "TargetDir was not recreated."
    # blacklist: a blacklisted entry below the source is left out -- unless the sources are a previous stage's
    t3, t4 = tmp_path / "t3", tmp_path / "t4"
    M.copy_op_execute(_op("/", [str(s)], str(t3) + "/"), blacklist=[str(s / "sub2"), str(inner)])
    assert not os.path.lexists(t3 / "sub2") and (t3 / "sub1" / "one").exists()
    M.copy_op_execute(_op("/", [str(s)], str(t4) + "/"), internal=True, blacklist=[str(s / "sub2")])
    assert os.path.isdir(t4 / "sub2")


def test_owner_rules(tmp_path):
    """the four scenarios of lib/fileio/copy.go:41-66"""
    s = tmp_path / "ctx"
    (s / "d").mkdir(parents=True)
    (s / "d" / "f").write_bytes(b"x")
    for p in (s, s / "d", s / "d" / "f"):
        os.chown(p, 111, 222)
    owner = lambda p: (os.lstat(p).st_uid, os.lstat(p).st_gid)               # noqa: E731
    M.copy_op_execute(_op("/", [str(s)], str(tmp_path / "plain" / "x") + "/"))                         # from context, no flags
    assert [owner(tmp_path / "plain" / "x" / q) for q in ("", "d", "d/f")] == [(0, 0)] * 3
    M.copy_op_execute(_op("/", [str(s)], str(tmp_path / "from" / "x") + "/"), internal=True)           # --from
    assert [owner(tmp_path / "from" / "x" / q) for q in ("", "d", "d/f")] == [(0, 0), (111, 222), (111, 222)]
    M.copy_op_execute(_op("/", [str(s)], str(tmp_path / "chown" / "x") + "/", 7, 8), chown=True, internal=True)
    assert [owner(tmp_path / "chown" / "x" / q) for q in ("", "d", "d/f")] == [(7, 8)] * 3 and owner(tmp_path / "chown") == (0, 0)
    M.copy_op_execute(_op("/", [str(s)], str(tmp_path / "arch" / "x") + "/"), internal=True, preserve_owner=True)   # --from --archive
    assert [owner(tmp_path / "arch" / "x" / q) for q in ("", "d", "d/f")] == [(111, 222)] * 3
    (tmp_path / "there").mkdir()                                                                      # an existing dst dir keeps its owner
    os.chown(tmp_path / "there", 55, 66)
    M.copy_op_execute(_op("/", [str(s)], str(tmp_path / "there") + "/", 7, 8), chown=True)
    assert owner(tmp_path / "there") == (55, 66) and owner(tmp_path / "there" / "d" / "f") == (7, 8)


def test_after_the_copy_the_scan_finds_what_the_copy_layer_holds(tmp_path):
    """a COPY step with --modifyfs both ways: its layer from the copy op (no disk needed) and, after Execute, from a scan"""
    root, ctx = tmp_path / "root", tmp_path / "ctx"
    (root / "app").mkdir(parents=True)
    (ctx / "src" / "sub").mkdir(parents=True)
    (ctx / "src" / "a").write_bytes(b"AAAA")
    (ctx / "src" / "sub" / "b").write_bytes(b"BB")
    os.symlink("a", ctx / "src" / "l")
    op = _op(ctx, ["src"], "/app/data/")
    with M.MemFS(str(root)) as by_copy, M.MemFS(str(root)) as by_scan:
        by_copy.scan()
        by_scan.scan()
        la = by_copy.add_layer_by_copy_ops([op])
        M.copy_op_execute(dict(op, dst=str(root) + op["dst"]))              # the build root is "/" in a real build
        lb = by_scan.scan()
        key = lambda e: (e["relpath"], e["kind"], e["mode"], e["size"], e["link_target"], e["uid"], e["gid"])   # noqa: E731
        assert [key(e) for e in la] == [key(e) for e in lb]
        assert [e["relpath"] for e in la] == ["app", "app/data", "app/data/a", "app/data/l", "app/data/sub", "app/data/sub/b"]


def test_checkpoint_moves_copy_from_sources_aside(tmp_path):
    """MemFS.Checkpoint (mem_fs.go:132-185, called by build_stage.go:342-345 at the end of a stage some later stage copies
    from): sources, given absolute or relative to the root or as patterns, land below new_root with the layout they have
    below the root; owners are kept, a created target directory takes the source's."""
    root, sandbox = tmp_path / "root", tmp_path / "sandbox"
    (root / "out" / "bin").mkdir(parents=True)
    (root / "out" / "bin" / "tool").write_bytes(b"ELF")
    (root / "out" / "conf-a.yaml").write_bytes(b"a")
    (root / "out" / "conf-b.yaml").write_bytes(b"b")
    (root / "skip").mkdir()
    (root / "skip" / "x").write_bytes(b"x")
    for p in (root / "out", root / "out" / "bin", root / "out" / "bin" / "tool"):
        os.chown(p, 42, 43)
    with M.MemFS(str(root), blacklist=[str(root / "skip")]) as fs:
        fs.checkpoint(str(sandbox), [str(root / "out" / "bin"), "out/conf-a.yaml", str(root / "out" / "conf-*.yaml"), "skip"])
        assert (sandbox / "out" / "bin" / "tool").read_bytes() == b"ELF"
        assert (sandbox / "out" / "conf-a.yaml").read_bytes() == b"a" and (sandbox / "out" / "conf-b.yaml").read_bytes() == b"b"
        assert not os.path.lexists(sandbox / "skip")                          # blacklisted: silently left out
        o = lambda p: (os.lstat(p).st_uid, os.lstat(p).st_gid)                # noqa: E731
        assert o(sandbox / "out" / "bin") == (42, 43) and o(sandbox / "out" / "bin" / "tool") == (42, 43)
        assert o(sandbox / "out") == (0, 0)                                   # an ancestor: 0755 root:root
        with pytest.raises(M.MiError) as ei:
            fs.checkpoint(str(sandbox), ["/etc/hostname"])                    # outside the root
        assert "trim src" in str(ei.value)
