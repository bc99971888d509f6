"""CPU tests of mi_memfs_untar: MemFS.UpdateFromTarReader with untar = true -- untarOneItem, tario.ApplyHeader, hard links
last, parent mtimes put back (lib/snapshot/mem_fs.go:165-255, 571-718; lib/tario/apply.go:23-47).

TestUntarFromPath (mem_fs_test.go:31-117) replayed, the single rules one by one, and a round trip on generated trees: the
layers a scan WRITES (layer writer), read back (tar reader) and untarred onto an empty root, reproduce the tree -- and the
next scan of that root finds nothing to add.  Runs as root here, as the reference's untar has to (chown)."""
import io
import os
import tarfile

import pytest
from hypothesis import given, settings

import makisu_amd as M
from test_host_diff_properties import tree_pairs
from test_host_memfs import _materialize

pytestmark = pytest.mark.skipif(os.geteuid() != 0, reason="untar chowns: needs root, like the reference's")


def _tar(path, members):
    """members: (name, kind, payload, kw): kind d/f/l(sym)/h(hard); payload = bytes | link target"""
    with tarfile.open(path, "w", format=tarfile.GNU_FORMAT) as tf:
        for name, kind, payload, kw in members:
            ti = tarfile.TarInfo(name)
            ti.mtime = kw.get("mtime", 1_500_000_000)
            ti.uid, ti.gid = kw.get("uid", 0), kw.get("gid", 0)
            ti.mode = kw.get("mode", 0o755 if kind == "d" else 0o644)
            if kind == "d":
                ti.type = tarfile.DIRTYPE
                tf.addfile(ti)
            elif kind == "l":
                ti.type, ti.linkname = tarfile.SYMTYPE, payload
                tf.addfile(ti)
            elif kind == "h":
                ti.type, ti.linkname = tarfile.LNKTYPE, payload
                tf.addfile(ti)
            else:
                ti.size = len(payload)
                tf.addfile(ti, io.BytesIO(payload))


def test_untar_from_path_replayed(tmp_path):
    root = tmp_path / "root"
    root.mkdir()
    a1 = str(tmp_path / "archive1.tar")
    _tar(a1, [("test.txt", "f", b"TEST", {"mode": 0o677}), ("test1", "d", None, {}), ("test2", "d", None, {}),
              ("test1/test1.txt", "f", b"TEST1", {"mode": 0o677}), ("test2.txt", "h", "test1/test1.txt", {"mode": 0o677}),
              ("target.txt", "f", b"TARGET", {"mode": 0o677}), ("mydir", "l", "/target.txt", {})])
    # "Files already existing under the memfs root."
    (root / "test1").mkdir()
    (root / "test1" / "test1.txt").write_bytes(b"TEST1")
    (root / "mydir").mkdir()
    with M.MemFS(str(root)) as fs:
        assert fs.update_from_tar(a1, untar=True) == 7
        assert (root / "test.txt").read_bytes() == b"TEST" and (root / "test1" / "test1.txt").read_bytes() == b"TEST1"
        assert os.path.islink(root / "mydir") and not os.path.isdir(os.readlink(root / "mydir"))
        assert os.readlink(root / "mydir") == str(root / "target.txt")            # an absolute target is re-rooted
        assert (root / "mydir").read_bytes() == b"TARGET"
        st = os.lstat(root / "test.txt")
        assert (st.st_mode & 0o7777, int(st.st_mtime), st.st_uid) == (0o677, 1_500_000_000, 0)
        assert os.lstat(root / "test2.txt").st_ino == os.lstat(root / "test1" / "test1.txt").st_ino   # the hard link, made last
        # the tree IS the disk -- but for the hard link: the tree holds a TypeLink header, the walk sees a regular file with
        # two names, and IsSimilarHeader never equates the two (compare.go:38-42): the first scan after a FROM re-adds
        # every hard-linked file of the base image as a regular file, in the reference as here
        assert [e["relpath"] for e in fs.scan()] == ["test2.txt"] and fs.scan() == []
        # "Whiteout files already existing in the memfs."
        a2 = str(tmp_path / "archive2.tar")
        _tar(a2, [(".wh.test.txt", "d", None, {}), (".wh.test1", "d", None, {})])
        assert fs.update_from_tar(a2, untar=True) == 2
        assert not os.path.lexists(root / "test.txt") and not os.path.lexists(root / "test1")
        assert sorted(e["relpath"] for e in fs.entries()) == ["mydir", "target.txt", "test2", "test2.txt"]
        assert fs.scan() == []


def test_untar_one_item_rules(tmp_path):
    """similar -> untouched; directory on directory -> updated in place, children stay; anything else -> removed and made
    again; parents keep their mtime; a missing parent directory is an error ("stat parent dir")"""
    root = tmp_path / "root"
    (root / "etc" / "keep").mkdir(parents=True)
    (root / "etc" / "keep" / "inner").write_bytes(b"stays")
    (root / "etc" / "same").write_bytes(b"same")
    os.utime(root / "etc" / "same", (1000, 1000))
    os.chmod(root / "etc" / "same", 0o644)
    (root / "etc" / "becomes-dir").write_bytes(b"file")
    (root / "etc" / "becomes-file").mkdir()
    (root / "etc" / "becomes-file" / "gone").write_bytes(b"x")
    os.utime(root / "etc", (777, 777))
    ino_same = os.lstat(root / "etc" / "same").st_ino
    t = str(tmp_path / "l.tar")
    _tar(t, [("etc/same", "f", b"same", {"mtime": 1000}),                          # similar header: not rewritten
             ("etc/keep", "d", None, {"mode": 0o700, "mtime": 2000, "uid": 12, "gid": 34}),
             ("etc/becomes-dir", "d", None, {"mtime": 3000}), ("etc/becomes-dir/f", "f", b"new", {"mtime": 3001}),
             ("etc/becomes-file", "f", b"now a file", {"mtime": 4000})])
    with M.MemFS(str(root)) as fs:
        fs.update_from_tar(t, untar=True)
        assert os.lstat(root / "etc" / "same").st_ino == ino_same
        st = os.lstat(root / "etc" / "keep")
        assert (st.st_mode & 0o7777, int(st.st_mtime), st.st_uid, st.st_gid) == (0o700, 2000, 12, 34)
        assert (root / "etc" / "keep" / "inner").read_bytes() == b"stays"
        assert (root / "etc" / "becomes-dir" / "f").read_bytes() == b"new" and int(os.lstat(root / "etc" / "becomes-dir").st_mtime) == 3000
        assert (root / "etc" / "becomes-file").read_bytes() == b"now a file"
        assert int(os.lstat(root / "etc").st_mtime) == 777                        # put back after the children changed
        bad = str(tmp_path / "bad.tar")
        _tar(bad, [("no/such/parent/f", "f", b"x", {})])
        with pytest.raises(M.MiError) as ei:
            fs.update_from_tar(bad, untar=True)
        assert ei.value.code == -5 and "stat parent dir of " + str(root / "no/such/parent/f") in str(ei.value)
        assert fs.update_from_tar(t, untar=True) == 0                             # the handle stays usable; nothing new


def _write_layer(path, layer):
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        with M.Layer(out_fd=fd, gzip_level=M.GZIP_OFF) as lw:
            for e in layer:
                lw.add(e, e["src"] if e["kind"] == M.KIND_FILE and e["src"] else None)
            lw.finish()
    finally:
        os.close(fd)


def _state(root):
    # (a symlink's mtime is not restored -- untarSymlink only chowns, mem_fs.go:672-684 -- which is why isSimilarSymlink
    # looks at the target alone)
    return [(e["relpath"], e["kind"], e["mode"], e["size"], e["mtime_sec"] if e["kind"] != M.KIND_SYMLINK else None,
             e["link_target"], e["uid"], e["gid"]) for e in M.tree_walk(root, root, (), M.TREE_SCAN, full=True)[1:]]


@settings(max_examples=60, deadline=None, derandomize=True, database=None)
@given(tree_pairs())
def test_layers_written_by_a_scan_untar_to_the_same_tree(tmp_path_factory, pair):
    before, after = pair
    base = tmp_path_factory.mktemp("rt")
    dir_a, dir_b, dir_r = str(base / "a"), str(base / "b"), str(base / "r")
    for d in (dir_a, dir_b, dir_r):
        os.mkdir(d)
    _materialize(dir_a, before)
    _materialize(dir_b, after)
    t1, t2 = str(base / "1.tar"), str(base / "2.tar")
    with M.MemFS(dir_a) as fs:
        _write_layer(t1, fs.scan())                                               # everything of A
    with M.MemFS(dir_b) as fs:
        fs.update_from_entries(before)                                            # the tree holds A, the disk B
        _write_layer(t2, fs.scan())                                               # B - A: content, ancestors, whiteouts
    with M.MemFS(dir_r) as fs:
        fs.update_from_tar(t1, untar=True)
        assert _state(dir_r) == _state(dir_a) and fs.scan() == []
        fs.update_from_tar(t2, untar=True)
        assert _state(dir_r) == _state(dir_b)
        assert fs.scan() == []
