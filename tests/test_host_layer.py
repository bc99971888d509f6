"""CPU tests of the layer writer (mi_layer_*, csrc/mi_layer.hip): tar framing + the two stream
digests of step.tarAndGzipDiffs / commitLayer (lib/builder/step/common.go:35-111), and the cache
entry codec (lib/cache/cache_manager.go:239-252).

Pins available here: the empty layer (1024 zero bytes -> 5f70bf18..., the constant the reference
holds at lib/docker/image/const_darwin.go:18), read-back of every field through python tarfile and
GNU tar (the reference's own write_test.go round-trips through the tar command the same way), and
agreement with the oracle's independent ustar header writer -- and, since round 3, the Go-written
layer tar the reference holds (testdata/files/busybox/393ccd5c.../layer.tar, 390 USTAR headers): every
header block and the whole stream are reproduced byte for byte (test_go_written_layer_*).  That pins
the USTAR path of the writer to Go's archive/tar bytes; the PAX path (long / non-ASCII names, large
ids) and tar.FileInfoHeader's Go >= 1.9 permission-only Mode stay unpinned (no Go toolchain here).
"""
import gzip
import hashlib
import io
import os
import shutil
import subprocess
import sys
import tarfile

import pytest

import makisu_amd as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EMPTY_TAR_TRAILER_DIGEST = "5f70bf18a086007016e948b04aed3b82103a36bea41755b6cddfaf10ace3c6ef"


def _build_tree(root):
    os.makedirs(root / "bin")
    os.makedirs(root / "etc" / "deep" / "er")
    os.makedirs(root / "empty-dir")
    (root / "bin" / "tool").write_bytes(os.urandom(300000))
    (root / "bin" / "tool").chmod(0o4755)                     # setuid survives (FileInfoHeader c_ISUID)
    (root / "etc" / "passwd").write_bytes(b"root:x:0:0\n")
    (root / "etc" / "empty").write_bytes(b"")
    (root / "etc" / "exact512").write_bytes(b"z" * 512)
    (root / "etc" / "deep" / "er" / "x.conf").write_bytes(b"k=v\n" * 1000)
    os.symlink("passwd", root / "etc" / "alias")
    os.symlink(str(root / "etc" / "passwd"), root / "etc" / "abs-alias")   # the scan walk root-trims it to /etc/passwd
    long_dir = root / ("d" * 60) / ("e" * 60)
    os.makedirs(long_dir)
    (long_dir / ("f" * 30)).write_bytes(b"split me")          # 152 chars: USTAR prefix/name split
    (long_dir / ("g" * 120)).write_bytes(b"pax path")         # base name > 100: PAX "path" record
    (root / "café.txt").write_bytes(b"non-ascii name")   # PAX "path" record
    os.utime(root / "etc" / "passwd", (1_500_000_000.75, 1_500_000_000.75))


def _write_layer(tmp_path, root, gzip_level, blacklist=()):
    ents = M.tree_walk(str(root), None, blacklist, M.TREE_SCAN, full=True)
    ents = [e for e in ents if e["relpath"] not in (".", "")]
    order = M.commit_order([e["relpath"] for e in ents])
    out = tmp_path / ("layer.%d" % gzip_level)
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        with M.Layer(out_fd=fd, gzip_level=gzip_level) as layer:
            for k in order:
                e = ents[k]
                layer.add(e, os.path.join(str(root), e["relpath"]) if e["kind"] == M.KIND_FILE else None)
            pair = layer.finish()
    finally:
        os.close(fd)
    return ents, [ents[k] for k in order], pair, out.read_bytes()


def test_empty_layer_is_the_go_tar_trailer():
    with M.Layer(gzip_level=M.GZIP_OFF) as layer:
        pair = layer.finish()
    assert pair["tar_bytes"] == 1024 and pair["n_entries"] == 0
    assert pair["tar_digest"] == "sha256:" + EMPTY_TAR_TRAILER_DIGEST
    assert hashlib.sha256(bytes(1024)).hexdigest() == EMPTY_TAR_TRAILER_DIGEST
    with M.Layer() as layer:                                   # with the gzip leg: same tar digest
        pair = layer.finish()
    assert pair["tar_digest"].hex() == EMPTY_TAR_TRAILER_DIGEST and pair["gzip_bytes"] > 0


@pytest.mark.parametrize("level", [M.GZIP_OFF, 0, 1, M.GZIP_DEFAULT, 9])
def test_layer_reads_back_and_digests_match(tmp_path, level):
    root = tmp_path / "rootfs"
    root.mkdir()
    _build_tree(root)
    ents, ordered, pair, blob = _write_layer(tmp_path, root, level)
    if level == M.GZIP_OFF:
        tar_bytes = blob
        assert pair["gzip_digest"] is None
    else:
        assert hashlib.sha256(blob).hexdigest() == pair["gzip_digest"].hex()      # gzipDigester
        assert len(blob) == pair["gzip_bytes"]                                    # GzipDescriptor.Size
        tar_bytes = gzip.decompress(blob)
    assert hashlib.sha256(tar_bytes).hexdigest() == pair["tar_digest"].hex()      # tarDigester
    assert len(tar_bytes) == pair["tar_bytes"] and len(tar_bytes) % 512 == 0
    assert tar_bytes[-1024:] == bytes(1024)
    assert pair["n_entries"] == len(ents)
    with tarfile.open(fileobj=io.BytesIO(tar_bytes)) as tf:
        members = tf.getmembers()
        assert len(members) == len(ordered)
        for m, e in zip(members, ordered):                    # commit order, field for field
            want = e["relpath"] + ("/" if e["kind"] == M.KIND_DIR else "")
            assert m.name.rstrip("/") == e["relpath"] and not m.name.startswith("/")
            assert (m.isdir(), m.isreg(), m.issym()) == (e["kind"] == 0, e["kind"] == 1, e["kind"] == 2), want
            assert m.mode == e["mode"] & 0o7777
            assert (m.uid, m.gid) == (e["uid"], e["gid"])
            assert m.mtime == e["mtime_sec"]                  # whole seconds (write.go:62)
            assert m.uname == "" and m.gname == ""            # mem_layer.go:161-162
            if e["kind"] == M.KIND_FILE:
                assert m.size == e["size"]
                assert tf.extractfile(m).read() == (root / e["relpath"]).read_bytes()
            else:
                assert m.size == 0
            if e["kind"] == M.KIND_SYMLINK:
                assert m.linkname == e["link_target"]
    names = [e["relpath"] for e in ordered]
    assert names == sorted(names, key=lambda p: ("/" + p))    # sort.Strings over absolute paths


def test_directory_names_carry_a_trailing_slash_and_formats(tmp_path):
    d = {"relpath": "/usr/lib", "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": 7}
    h = M.layer_header_bytes(d)
    assert len(h) == 512 and h[:8] == b"usr/lib/" and h[156:157] == b"5"
    assert h[257:265] == b"ustar\x0000"
    assert h[100:108] == b"0000755\x00" and h[124:136] == b"00000000000\x00" and h[136:148] == b"00000000007\x00"
    assert h[329:337] == b"0000000\x00" and h[337:345] == b"0000000\x00"      # Devmajor/Devminor via templateV7Plus
    assert h[154:156] == b"\x00 "                              # checksum: 6 digits, NUL, space
    assert int(h[148:154], 8) == sum(h[:148]) + 8 * 32 + sum(h[156:])
    # prefix split exactly like splitUSTARPath
    name = "p" * 90 + "/" + "s" * 60
    h = M.layer_header_bytes({"relpath": name, "kind": M.KIND_FILE, "mode": 0o644, "size": 3})
    assert len(h) == 512 and h[:60] == b"s" * 60 and h[60] == 0 and h[345:345 + 90] == b"p" * 90
    # unsplittable long name -> PAX record file first
    name = "q" * 130
    h = M.layer_header_bytes({"relpath": name, "kind": M.KIND_FILE, "mode": 0o600, "size": 5})
    assert len(h) == 1536 and h[156:157] == b"x" and h[:13] == b"PaxHeaders.0/"
    rec = b"%d path=%s\n" % (len(name) + 10, name.encode())
    assert h[512:512 + len(rec)] == rec and int(h[124:135], 8) == len(rec)
    assert h[1024:1024 + 100] == name.encode()[:100] and h[1024 + 156:1024 + 157] == b"0"
    # a name cut at the field's end right behind a "/" gets a NUL where the trailing slashes begin (formatString: "Some
    # buggy readers treat regular files with a trailing slash in the V7 path field as a directory"); one byte only
    name = "d" * 99 + "/" + "f" * 120
    h = M.layer_header_bytes({"relpath": name, "kind": M.KIND_FILE, "mode": 0o600, "size": 5})
    assert len(h) == 1536 and h[1024:1024 + 100] == b"d" * 99 + b"\x00"
    name = "d" * 97 + "///" + "f" * 120
    h = M.layer_header_bytes({"relpath": name, "kind": M.KIND_FILE, "mode": 0o600, "size": 5})
    assert h[1024:1024 + 100] == b"d" * 97 + b"\x00//"
    # ... and only a string that was CUT: a directory name of exactly 100 bytes keeps its slash
    h = M.layer_header_bytes({"relpath": "e" * 99, "kind": M.KIND_DIR, "mode": 0o40755})
    assert len(h) == 512 and h[:100] == b"e" * 99 + b"/"
    link = "t" * 99 + "/" + "u" * 30                      # the link field is written by the same formatter
    h = M.layer_header_bytes({"relpath": "l", "kind": M.KIND_SYMLINK, "mode": 0o120777, "link_target": link})
    assert len(h) == 1536 and b" linkpath=" + link.encode() + b"\n" in h[512:1024]
    assert h[1024 + 157:1024 + 257] == b"t" * 99 + b"\x00"
    # large uid -> PAX uid record, field zeroed
    h = M.layer_header_bytes({"relpath": "f", "kind": M.KIND_FILE, "mode": 0o600, "size": 1, "uid": 3000000})
    assert len(h) == 1536 and b" uid=3000000\n" in h[512:1024] and h[1024 + 108:1024 + 116] == b"0000000\x00"


def test_headers_agree_with_the_oracles_independent_writer(oracle):
    """Two restatements of the ustar layout written separately (product C++, oracle C) must agree
    wherever the oracle's simpler writer applies: regular files, uid/gid 0, ASCII names."""
    for name in ("a", "dir/file.txt", "x" * 100, "p" * 80 + "/" + "s" * 90, "a/" * 70 + "end"):
        for size, mtime, mode in ((0, 0, 0o644), (12345, 1_600_000_000, 0o755), (8 << 30 - 1, 1, 0o600)):
            mine = M.layer_header_bytes({"relpath": name, "kind": M.KIND_FILE, "mode": mode, "size": size,
                                         "mtime_sec": mtime})
            try:
                ref = oracle.tar_header(name, size, mtime, mode)
            except ValueError:
                assert len(mine) > 512                         # the oracle gives up where Go switches to PAX
                continue
            assert mine == ref, name


def test_gnu_tar_extracts_the_layer(tmp_path):
    if not shutil.which("tar"):
        pytest.skip("no tar binary")
    root = tmp_path / "rootfs"
    root.mkdir()
    _build_tree(root)
    _, ordered, pair, blob = _write_layer(tmp_path, root, M.GZIP_DEFAULT)
    (tmp_path / "layer.tgz").write_bytes(blob)
    out = tmp_path / "extracted"
    out.mkdir()
    subprocess.check_call(["tar", "-xzf", str(tmp_path / "layer.tgz"), "-C", str(out), "--no-same-owner"])
    for e in ordered:
        p = out / e["relpath"]
        if e["kind"] == M.KIND_FILE:
            assert p.read_bytes() == (root / e["relpath"]).read_bytes()
            assert int(os.lstat(p).st_mtime) == e["mtime_sec"]
        elif e["kind"] == M.KIND_SYMLINK:
            assert os.readlink(p) == e["link_target"]
        else:
            assert p.is_dir()


def test_whiteouts_hardlinks_and_errors(tmp_path):
    src = tmp_path / "data"
    src.write_bytes(b"0123456789")
    fd = os.open(tmp_path / "l.tar", os.O_WRONLY | os.O_CREAT, 0o644)
    with M.Layer(out_fd=fd, gzip_level=M.GZIP_OFF) as layer:
        layer.add({"relpath": "/opt", "kind": M.KIND_DIR, "mode": 0o755, "mtime_sec": 5})
        layer.add_whiteout("/opt/gone")
        layer.add({"relpath": "opt/data", "kind": M.KIND_FILE, "mode": 0o640, "size": 10, "uid": 7, "gid": 8}, str(src))
        layer.add({"relpath": "opt/hard", "kind": M.KIND_HARDLINK, "mode": 0o640, "link_target": "opt/data"})
        with pytest.raises(M.MiError) as ei:
            layer.add_whiteout("/opt/.wh.already")             # mem_layer.go:216-218
        assert "whiteout prefix" in str(ei.value)
    os.close(fd)
    # a failed layer stays failed; build it again without the bad call
    fd = os.open(tmp_path / "l.tar", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    with M.Layer(out_fd=fd, gzip_level=M.GZIP_OFF) as layer:
        layer.add({"relpath": "/opt", "kind": M.KIND_DIR, "mode": 0o755, "mtime_sec": 5})
        layer.add_whiteout("/opt/gone")
        layer.add({"relpath": "opt/data", "kind": M.KIND_FILE, "mode": 0o640, "size": 10, "uid": 7, "gid": 8}, str(src))
        layer.add({"relpath": "opt/hard", "kind": M.KIND_HARDLINK, "mode": 0o640, "link_target": "opt/data"})
        pair = layer.finish()
    os.close(fd)
    raw = (tmp_path / "l.tar").read_bytes()
    assert hashlib.sha256(raw).hexdigest() == pair["tar_digest"].hex()
    with tarfile.open(tmp_path / "l.tar") as tf:
        m = {x.name: x for x in tf.getmembers()}
        wh = m["opt/.wh.gone"]
        assert wh.isreg() and wh.size == 0 and wh.mode == 0 and wh.mtime == 0 and wh.uid == 0   # zero header
        assert m["opt/hard"].islnk() and m["opt/hard"].linkname == "opt/data"
        assert (m["opt/data"].uid, m["opt/data"].gid) == (7, 8)
    # CopyN semantics: a source shorter than the header's size is an error, a longer one is cut
    with M.Layer(gzip_level=M.GZIP_OFF) as layer:
        with pytest.raises(M.MiError) as ei:
            layer.add({"relpath": "short", "kind": M.KIND_FILE, "mode": 0o600, "size": 11}, str(src))
        assert ei.value.code == -5 and "EOF" in str(ei.value)
    with M.Layer(gzip_level=M.GZIP_OFF) as layer:
        layer.add({"relpath": "cut", "kind": M.KIND_FILE, "mode": 0o600, "size": 4}, str(src))
        pair = layer.finish()
        assert pair["tar_bytes"] == 512 + 512 + 1024
    with M.Layer(gzip_level=M.GZIP_OFF) as layer:
        with pytest.raises(M.MiError) as ei:
            layer.add({"relpath": "dev/null", "kind": 4, "mode": 0o666})         # write.go:49-51
        assert "unsupported type" in str(ei.value)
        # a failed layer stays failed: nothing more goes in, and it does not finish
        with pytest.raises(M.MiError) as ei:
            layer.add({"relpath": "fine", "kind": M.KIND_DIR, "mode": 0o40755})
        assert "finished or failed" in str(ei.value)
        with pytest.raises(M.MiError):
            layer.finish()
    with M.Layer(gzip_level=M.GZIP_OFF) as layer:
        with pytest.raises(M.MiError) as ei:
            layer.add({"relpath": "nope", "kind": M.KIND_FILE, "mode": 0o600, "size": 1}, str(tmp_path / "missing"))
        assert ei.value.code == -5 and "open src file" in str(ei.value)


def test_big_file_spans_many_blocks(tmp_path):
    big = tmp_path / "big"
    data = os.urandom(5 * (1 << 20) + 777)
    big.write_bytes(data)
    fd = os.open(tmp_path / "b.tgz", os.O_WRONLY | os.O_CREAT, 0o644)
    with M.Layer(out_fd=fd, gzip_level=1) as layer:
        layer.add({"relpath": "big", "kind": M.KIND_FILE, "mode": 0o644, "size": len(data)}, str(big))
        pair = layer.finish()
    os.close(fd)
    tar_bytes = gzip.decompress((tmp_path / "b.tgz").read_bytes())
    assert hashlib.sha256(tar_bytes).hexdigest() == pair["tar_digest"].hex()
    assert tar_bytes[512:512 + len(data)] == data


def test_cache_entry_codec():
    """createEntry / parseEntry / the key prefix (lib/cache/cache_manager.go:34-35,239-252)."""
    assert M.cache_key("abc123") == "makisu_builder_cache_abc123"
    t, g = hashlib.sha256(b"tar").digest(), hashlib.sha256(b"gz").digest()
    entry = M.cache_create_entry(t, g)
    assert entry == "%s,%s" % (t.hex(), g.hex())
    td, gd = M.cache_parse_entry(entry)
    assert td == "sha256:" + t.hex() and gd == "sha256:" + g.hex() and td.hex() == t.hex()
    assert M.cache_create_entry() == "MAKISU_CACHE_EMPTY"      # createEntry(nil)
    assert M.cache_parse_entry("MAKISU_CACHE_EMPTY") is None   # PullCache -> (nil, nil)
    for bad in ("", "nocomma", t.hex(), t.hex() + "," + "zz" * 32, t.hex()[:-1] + "," + g.hex()):
        with pytest.raises(ValueError):
            M.cache_parse_entry(bad)
    up = M.cache_parse_entry(t.hex().upper() + "," + g.hex().upper())     # hex is hex: "ABCDEF" as "abcdef"
    assert (up[0].hex(), up[1].hex()) == (t.hex(), g.hex())
    # the layer writer's pair feeds the codec directly
    with M.Layer() as layer:
        pair = layer.finish()
        with pytest.raises(M.MiError):                         # a finished layer is finished
            layer.finish()
        with pytest.raises(M.MiError):
            layer.add({"relpath": "late", "kind": M.KIND_DIR, "mode": 0o40755})
    e = M.cache_create_entry(pair["tar_sha256"], pair["gzip_sha256"])
    assert M.cache_parse_entry(e) == (pair["tar_digest"], pair["gzip_digest"])


def test_ustar_headers_equal_python_tarfile():
    """A third independent ustar writer: Python's tarfile (USTAR_FORMAT) produces the same header as
    the product's framer for names that need no prefix split (the two split long names at different
    slashes, both legally) -- except devmajor/devminor, which Go's templateV7Plus formats as
    "0000000\0" for EVERY entry while Python 3.10 leaves them empty for non-devices; the test puts
    Go's form there and recomputes the checksum.  With the oracle's writer that makes three
    implementations agreeing on every other byte; Go's archive/tar itself stays unpinned."""
    cases = [
        ({"relpath": "etc/passwd", "kind": M.KIND_FILE, "mode": 0o100644, "size": 1234, "mtime_sec": 1_600_000_000,
          "uid": 0, "gid": 0}, tarfile.REGTYPE),
        ({"relpath": "usr/local/bin/tool", "kind": M.KIND_FILE, "mode": 0o104755, "size": 0, "mtime_sec": 1,
          "uid": 1000, "gid": 100}, tarfile.REGTYPE),
        ({"relpath": "var/lib", "kind": M.KIND_DIR, "mode": 0o40750, "mtime_sec": 86400, "uid": 7, "gid": 8}, tarfile.DIRTYPE),
        ({"relpath": "bin/sh", "kind": M.KIND_SYMLINK, "mode": 0o120777, "link_target": "/bin/busybox", "mtime_sec": 5},
         tarfile.SYMTYPE),
        ({"relpath": "bin/ln2", "kind": M.KIND_HARDLINK, "mode": 0o644, "link_target": "bin/busybox", "mtime_sec": 5},
         tarfile.LNKTYPE),
        ({"relpath": "x" * 100, "kind": M.KIND_FILE, "mode": 0o600, "size": 8 * 1024 ** 3 - 1, "mtime_sec": 2 ** 31},
         tarfile.REGTYPE),
    ]
    for e, typ in cases:
        ti = tarfile.TarInfo(e["relpath"])
        ti.type = typ
        ti.mode = e["mode"] & 0o7777
        ti.size = e.get("size", 0)
        ti.mtime = e["mtime_sec"]
        ti.uid, ti.gid = e.get("uid", 0), e.get("gid", 0)
        ti.uname = ti.gname = ""
        ti.linkname = e.get("link_target") or ""
        want = bytearray(ti.tobuf(format=tarfile.USTAR_FORMAT, encoding="utf-8", errors="surrogateescape"))
        want[329:337] = want[337:345] = b"0000000\x00"
        want[148:156] = b" " * 8
        want[148:156] = b"%06o\x00 " % sum(want)
        assert M.layer_header_bytes(e) == bytes(want), e["relpath"]


def test_parallel_gzip_is_one_member_and_independent_of_the_thread_count(tmp_path, monkeypatch):
    """The gzip leg deflates 1 MiB blocks of the tar on a thread pool (the reference uses pgzip,
    lib/tario/gzip.go:31-47).  The blob must be ONE gzip member that zlib reads back to exactly the
    tar, at every level, with compressible, incompressible and empty content -- and its bytes must
    not depend on how many threads compressed it."""
    import zlib
    files = {"zeros": bytes(3 * (1 << 20) + 5), "rand": os.urandom(2 * (1 << 20) + 123), "empty": b"",
             "text": b"layer layer layer\n" * 200000, "tiny": b"x"}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    blobs = {}
    for threads in ("1", "3", "16"):
        monkeypatch.setenv("MI_GZIP_THREADS", threads)
        for level in (1, M.GZIP_DEFAULT, 9):
            out = tmp_path / ("l%s_%d.tgz" % (threads, level))
            fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            with M.Layer(out_fd=fd, gzip_level=level) as layer:
                for name in sorted(files):
                    layer.add({"relpath": name, "kind": M.KIND_FILE, "mode": 0o644, "size": len(files[name])},
                              str(tmp_path / name))
                pair = layer.finish()
            os.close(fd)
            blob = out.read_bytes()
            assert hashlib.sha256(blob).hexdigest() == pair["gzip_digest"].hex() and len(blob) == pair["gzip_bytes"]
            # the member's header as compress/gzip (and pgzip) write it: deflate, no flags, no mtime, XFL 2 for the best
            # compression / 4 for the best speed / else 0, OS 255 "unknown"
            assert blob[:10] == bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, {9: 2, 1: 4}.get(level, 0), 255])
            d = zlib.decompressobj(16 + 15)                       # gzip framing, exactly one member
            tar_bytes = d.decompress(blob) + d.flush()
            assert d.eof and d.unused_data == b""
            assert len(tar_bytes) == pair["tar_bytes"]
            assert hashlib.sha256(tar_bytes).hexdigest() == pair["tar_digest"].hex()
            assert len(blob) < len(tar_bytes) // 2                # the zeros and the text did compress
            blobs.setdefault(level, set()).add(blob)
    assert all(len(v) == 1 for v in blobs.values())               # same bytes with 1, 3 and 16 threads


def test_chunk_root_helper_equals_the_oracle_definition():
    """mi_chunk_root (host): flat below 65 digests, fan-out-64 tree above -- the oracle's
    mi_ref_chunk_root on the same lists, including the sizes where a level appears."""
    import numpy as np
    from oracle import mi_oracle as O
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 63, 64, 65, 128, 129, 4095, 4096, 4097, 70000):
        dg = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        assert M.chunk_root(dg) == bytes(O.chunk_root(dg)), n


# ---- the Go-written layer of the reference's fixtures ------------------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GO_LAYER_TAR_DIGEST = "4ac76077f2c741c856a2419dfdb0804b18e48d2e1a9ce9c6a3f0605a2078caba"


def test_go_written_layer_every_header_block_is_reproduced(tmp_path, go_layer_tar):
    """(i) each of the 390 entries mi_tar_entries reads from the Go-written tar, handed back to the
    writer with MI_LAYER_MODE_WITH_TYPE (the Mode field keeps the file-type bits: FileInfoHeader up to
    Go 1.8, which is what wrote this fixture), gives exactly the 512 bytes Go wrote: names with the directory slash,
    "ustar\0" + "00", empty uname/gname, "0000000\0" dev fields, the "%06o\0 " checksum, hard links
    with size 0.  Without the flag (mode = an st_mode through FileInfoHeader's Go >= 1.9 rule) the
    ONLY bytes that differ are the type bits of the mode field and the checksum."""
    raw, members = go_layer_tar
    p = tmp_path / "layer.tar"
    p.write_bytes(raw)
    ents = M.tar_entries(str(p))
    assert len(ents) == len(members) == 390
    kinds = {"5": M.KIND_DIR, "0": M.KIND_FILE, "1": M.KIND_HARDLINK, "2": M.KIND_SYMLINK}
    n_by_kind = {}
    for e, m in zip(ents, members):
        assert e["kind"] == kinds[m["type"]] and e["relpath"].rstrip("/") == m["name"].rstrip("/")
        n_by_kind[m["type"]] = n_by_kind.get(m["type"], 0) + 1
        want = raw[m["header_offset"]:m["header_offset"] + 512]
        assert hashlib.sha256(want).hexdigest() == m["header_sha256"]
        got = M.layer_header_bytes(e, mode_with_type=True)
        assert got == want, (m["name"], [i for i in range(512) if got[i] != want[i]][:8])
        if e["kind"] == M.KIND_FILE:
            assert e["data_offset"] == m["data_offset"] and e["size"] == m["size"]
        cooked = M.layer_header_bytes(e)                      # FileInfoHeader's permission-only Mode
        diff = [i for i in range(512) if cooked[i] != want[i]]
        assert diff and all(100 <= i < 108 or 148 <= i < 156 for i in diff), (m["name"], diff)
        assert cooked[100:108] == b"%07o\x00" % (int(m["mode_field"].rstrip("\x00"), 8) & 0o7777)
    assert n_by_kind == {"1": 372, "5": 12, "0": 6}


def test_go_written_layer_is_reframed_byte_for_byte(tmp_path, go_layer_tar):
    """(ii) the whole layer through mi_layer_begin / add / finish -- members extracted to a directory,
    entries (hard links included) in stream order -- is the Go-written stream again: 1 308 672 bytes,
    TarDigest 4ac76077...caba, and through the gzip leg a blob that inflates to it."""
    raw, members = go_layer_tar
    p = tmp_path / "layer.tar"
    p.write_bytes(raw)
    ents = M.tar_entries(str(p))
    src = tmp_path / "members"
    src.mkdir()
    paths = {}
    for i, (e, m) in enumerate(zip(ents, members)):
        if e["kind"] == M.KIND_FILE:
            data = raw[m["data_offset"]:m["data_offset"] + m["size"]]
            assert hashlib.sha256(data).hexdigest() == m["data_sha256"]
            paths[i] = src / ("m%03d" % i)
            paths[i].write_bytes(data)
    for level in (M.GZIP_OFF, M.GZIP_DEFAULT):
        out = tmp_path / ("out.%d" % level)
        fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            with M.Layer(out_fd=fd, gzip_level=level, mode_with_type=True) as layer:
                for i, e in enumerate(ents):
                    layer.add(e, str(paths[i]) if i in paths else None)
                pair = layer.finish()
        finally:
            os.close(fd)
        assert pair["tar_bytes"] == 1308672 and pair["n_entries"] == 390
        assert pair["tar_digest"] == "sha256:" + GO_LAYER_TAR_DIGEST
        blob = out.read_bytes()
        if level == M.GZIP_OFF:
            assert blob == raw
        else:
            assert gzip.decompress(blob) == raw
            assert pair["gzip_digest"].hex() == hashlib.sha256(blob).hexdigest()
    # and the reader's own digest of the stream (mi_tar_inflate over the reference's blob) agrees
    import base64
    import json
    blob_path = tmp_path / "blob"
    blob_path.write_bytes(base64.b64decode(
        json.load(open(os.path.join(GOLDEN, "sha256_reference_fixtures.json")))["vectors"][0]["file_b64"]))
    inf = M.tar_inflate(str(blob_path))
    assert inf["tar_digest"] == "sha256:" + GO_LAYER_TAR_DIGEST and inf["tar_bytes"] == 1308672


def test_pax_extended_header_block_byte_for_byte():
    """writeRawFile's block for the 'x' member laid out by hand (archive/tar writer.go: name = toASCII(path.Join(dir,
    "PaxHeaders.0", file)) cut to 100 bytes, mode / uid / gid / mtime 0 as zero-padded octal, the records' length as the
    size, typeflag 'x', magic "ustar\\0" + "00", NO uname / gname / device fields, checksum 6 digits + NUL + space) and the
    main header that follows it (templateV7Plus with toASCII strings; numbers that do not fit written as zero)."""
    def fin(b):
        b[148:156] = b" " * 8
        b[148:156] = b"%06o\x00 " % sum(b)
        return bytes(b)

    name = "caf\u00e9/" + "n" * 120                      # non-ASCII AND too long: a path record, no prefix split
    e = {"relpath": name, "kind": M.KIND_FILE, "mode": 0o100640, "size": 5, "uid": 3000000, "gid": 12, "mtime_sec": 77}
    h = M.layer_header_bytes(e)
    raw_name = name.encode()
    recs = b"".join(sorted([b"%d path=%s\n" % (len(raw_name) + 10, raw_name), b"15 uid=3000000\n"]))   # (3 digits + " path=" + "\n")
    x = bytearray(512)
    xname = ("caf/PaxHeaders.0/" + "n" * 120).encode()[:100]
    x[0:len(xname)] = xname
    x[100:108] = x[108:116] = x[116:124] = b"0000000\x00"
    x[124:136] = b"%011o\x00" % len(recs)
    x[136:148] = b"00000000000\x00"
    x[156:157] = b"x"
    x[257:265] = b"ustar\x0000"
    assert h[:512] == fin(x)
    assert h[512:1024] == recs + bytes(512 - len(recs))
    m = bytearray(512)
    ascii_name = ("caf/" + "n" * 120).encode()
    m[0:100] = ascii_name[:100]
    m[100:108] = b"0000640\x00"
    m[108:116] = b"0000000\x00"                            # the uid does not fit: zero, the record carries it
    m[116:124] = b"0000014\x00"
    m[124:136] = b"00000000005\x00"
    m[136:148] = b"00000000115\x00"
    m[156:157] = b"0"
    m[257:265] = b"ustar\x0000"
    m[329:337] = m[337:345] = b"0000000\x00"
    assert h[1024:] == fin(m) and len(h) == 1536
    # a lone byte 0x80 is not ASCII either (isASCII: c >= 0x80)
    h = M.layer_header_bytes({"relpath": "a\udc80b", "kind": M.KIND_FILE, "mode": 0o600, "size": 0})
    assert len(h) == 1536 and h[512:512 + 12] == b"12 path=a\x80b\n" and h[1024:1027] == b"ab\x00"
    # a name of exactly 100 bytes is not split, slashes or not (splitUSTARPath: length <= nameSize)
    name = "d" * 49 + "/" + "f" * 50
    h = M.layer_header_bytes({"relpath": name, "kind": M.KIND_FILE, "mode": 0o600, "size": 0})
    assert len(h) == 512 and h[:100] == name.encode() and h[345:500] == bytes(155)


def test_pax_name_is_path_cleaned_like_go():
    """the PaxHeaders.0 name is path.Join(dir, "PaxHeaders.0", file) -- path.Clean'd."""
    long = "g" * 120
    for name, want in ((("a//b/./" + long), "a/b/PaxHeaders.0/" + long), (("a/../b/" + long), "b/PaxHeaders.0/" + long),
                       (long, "PaxHeaders.0/" + long)):
        hb = M.layer_header_bytes({"relpath": name, "kind": M.KIND_FILE, "mode": 0o100644, "size": 1})
        assert len(hb) >= 1536 and hb[156:157] == b"x"
        assert hb[:100].rstrip(b"\0") == want.encode()[:100].rstrip(b"/")


def test_cache_parse_entry_str_accepts_what_the_reference_accepts():
    """parseEntry (cache_manager.go:239-245) splits at the first comma and validates nothing."""
    assert M.cache_parse_entry_str("abc,def") == ("sha256:abc", "sha256:def")
    assert M.cache_parse_entry_str("a,b,c") == ("sha256:a", "sha256:b,c")
    assert M.cache_parse_entry_str(",") == ("sha256:", "sha256:")
    with pytest.raises(ValueError):
        M.cache_parse_entry_str("no comma")


@pytest.mark.parametrize("level", [M.GZIP_OFF, M.GZIP_DEFAULT])
def test_a_failing_sink_fails_the_layer(tmp_path, level):
    """lib/stream/multi_writer_test.go:45-57 (TestMultiWriterFailure): one sink of the tee cannot write --
    the device is full, or the descriptor is gone -- and the writer reports it ("failed to write: ...",
    multi_writer.go:62-64) from mi_layer_add or, at the latest, mi_layer_finish; never a digest pair."""
    src = tmp_path / "payload"
    src.write_bytes(os.urandom(5 << 20))
    entry = {"relpath": "payload", "kind": M.KIND_FILE, "mode": 0o100644, "size": 5 << 20, "mtime_sec": 1}
    for make_fd, what in ((lambda: os.open("/dev/full", os.O_WRONLY), "No space left"), (None, "Bad file descriptor")):
        if make_fd is None:
            fd = os.open(tmp_path / "gone", os.O_WRONLY | os.O_CREAT, 0o644)
            os.close(fd)                                       # the writer gets a descriptor that is closed
        else:
            fd = make_fd()
        try:
            with M.Layer(out_fd=fd, gzip_level=level) as layer:
                with pytest.raises(M.MiError) as ei:
                    for _ in range(4):                         # enough blocks for the sink to have tried
                        layer.add(entry, str(src))
                    layer.finish()
                assert ei.value.code == -5 and "failed to write" in str(ei.value) and what in str(ei.value)
                with pytest.raises(M.MiError):                 # and the layer stays failed
                    layer.finish()
        finally:
            if make_fd is not None:
                os.close(fd)


def test_mem_layer_test_go_cases_replayed(tmp_path):
    """lib/snapshot/mem_layer_test.go: TestCreateHeader (:27-90: a directory's header name ends in "/", a regular file's
    and a symlink's do not, the types), TestAddHeader (:92-149: a destination whose base name has the whiteout prefix is
    filed as a whiteout -- committed header-only, a zero header carrying only the name) and TestAddWhiteout (:151-186:
    "<dir>/.wh.<base>", and a path that already has the prefix is refused)."""
    src = tmp_path / "test"
    src.write_bytes(b"content that a whiteout must not carry")
    # TestCreateHeader: dst "/tmp/testDest"
    h = M.layer_header_bytes({"relpath": "/tmp/testDest", "kind": M.KIND_DIR, "mode": 0o40700})
    assert h[:13] == b"tmp/testDest/" and h[13] == 0 and h[156:157] == b"5"
    h = M.layer_header_bytes({"relpath": "/tmp/testDest", "kind": M.KIND_FILE, "mode": 0o100600, "size": 0})
    assert h[:12] == b"tmp/testDest" and h[12] == 0 and h[156:157] == b"0"
    h = M.layer_header_bytes({"relpath": "/tmp/testDest", "kind": M.KIND_SYMLINK, "mode": 0o120777, "link_target": str(src)})
    assert h[:12] == b"tmp/testDest" and h[12] == 0 and h[156:157] == b"2"
    # TestAddHeader/Whiteout: dst "/tmp/.wh.testDest" -> whiteout of "/tmp/testDest", hdr.Name "tmp/.wh.testDest"
    zero = M.layer_header_bytes({"relpath": "/tmp/.wh.testDest", "kind": M.KIND_FILE, "mode": 0o100644, "size": 38,
                                 "uid": 5, "gid": 6, "mtime_sec": 77})
    assert len(zero) == 512 and zero[:16] == b"tmp/.wh.testDest" and zero[16] == 0 and zero[156:157] == b"0"
    assert zero[100:108] == b"0000000\x00" and zero[108:116] == b"0000000\x00" and zero[116:124] == b"0000000\x00"
    assert zero[124:136] == b"00000000000\x00" and zero[136:148] == b"00000000000\x00"     # size 0, mtime 0: a zero tar.Header
    fd = os.open(tmp_path / "l.tar", os.O_WRONLY | os.O_CREAT, 0o644)
    with M.Layer(out_fd=fd, gzip_level=M.GZIP_OFF) as layer:
        layer.add({"relpath": "/tmp/.wh.testDest", "kind": M.KIND_FILE, "mode": 0o100644, "size": 38, "uid": 5, "gid": 6,
                   "mtime_sec": 77}, str(src))
        layer.add_whiteout("/tmp/testDest2")                                   # TestAddWhiteout/RegularFile
        with pytest.raises(M.MiError):
            layer.add_whiteout("/tmp/.wh.testDest")                            # TestAddWhiteout/RejectWhiteout
    os.close(fd)
    fd = os.open(tmp_path / "l.tar", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    with M.Layer(out_fd=fd, gzip_level=M.GZIP_OFF) as layer:
        layer.add({"relpath": "/tmp/.wh.testDest", "kind": M.KIND_FILE, "mode": 0o100644, "size": 38, "uid": 5, "gid": 6,
                   "mtime_sec": 77}, str(src))
        layer.add_whiteout("/tmp/testDest2")
        pair = layer.finish()
    os.close(fd)
    raw = (tmp_path / "l.tar").read_bytes()
    assert len(raw) == 512 + 512 + 1024 and raw[:512] == zero and pair["n_entries"] == 2      # header-only, both
    assert raw[512:512 + 17] == b"tmp/.wh.testDest2" and raw[512 + 17] == 0
    # the same zero header whichever way the whiteout was asked for
    fd = os.open(tmp_path / "m.tar", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    with M.Layer(out_fd=fd, gzip_level=M.GZIP_OFF) as layer:
        layer.add_whiteout("/tmp/testDest")
        layer.finish()
    os.close(fd)
    assert (tmp_path / "m.tar").read_bytes()[:512] == zero


def _commit_scan_layer(tmp_path, root, before, tag):
    """AddLayerByScan + commitLayer with the host entry points: walk, diff against the previous walk, write the changed
    entries, their carried ancestors and one whiteout per deleted subtree in commit order through the layer writer (gzip
    on); returns (this walk, the member names read back from the gzip blob as '/'-rooted names, the DigestPair)."""
    after = M.tree_walk(str(root), mode=M.TREE_SCAN, full=True)
    flags, wh = M.snapshot_diff(before, after)
    items = [("entry", e) for e, f in zip(after, flags) if f != M.DIFF_SAME and e["relpath"] not in (".", "")]
    items += [("whiteout", e) for e, w in zip(before, wh) if w]

    def key(it):
        e = it[1]
        return e["relpath"] if it[0] == "entry" else e["relpath"]       # rangeFiles sorts by the map key = the path
    items.sort(key=key)
    out = tmp_path / ("layer_%s.tar.gz" % tag)
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        with M.Layer(out_fd=fd, gzip_level=M.GZIP_DEFAULT) as layer:
            for what, e in items:
                if what == "whiteout":
                    layer.add_whiteout("/" + e["relpath"].lstrip("/"))
                else:
                    layer.add(e, os.path.join(str(root), e["relpath"]) if e["kind"] == M.KIND_FILE else None)
            pair = layer.finish()
    finally:
        os.close(fd)
    blob = out.read_bytes()
    assert hashlib.sha256(blob).digest() == pair["gzip_sha256"] and pair["gzip_bytes"] == len(blob)
    raw = gzip.decompress(blob)
    assert hashlib.sha256(raw).digest() == pair["tar_sha256"] and pair["tar_bytes"] == len(raw)
    with tarfile.open(fileobj=io.BytesIO(raw)) as tf:
        names = ["/" + m.name.lstrip("/") + ("/" if m.isdir() else "") for m in tf.getmembers()]
    return after, names, pair


def test_commit_diffs_sequence_like_the_reference(tmp_path):
    """lib/builder/step/common_test.go:94-152 (TestCommitDiffs): four RUN steps on one build context and what the gzipped
    layer tar of each must hold -- here the steps' effects are applied with python and every layer goes walk ->
    snapshot diff -> commit order -> tar framing -> two stream digests -> gzip, all through the C ABI.  Also
    TestTarAndGzipDiffsEmpty / AddedFile (:52-92)."""
    root = tmp_path / "ctx"
    os.makedirs(root)
    s0 = M.tree_walk(str(root), mode=M.TREE_SCAN, full=True)
    # TestTarAndGzipDiffsEmpty: nothing written -> no members (the blob is the gzip of the 1024-byte trailer)
    s0b, names, pair = _commit_scan_layer(tmp_path, root, s0, "empty")
    assert names == [] and pair["tar_bytes"] == 1024 and pair["tar_digest"].hex() == EMPTY_TAR_TRAILER_DIGEST
    # "touch file1 && touch file2"   (== TestTarAndGzipDiffsAddedFile for the first of them)
    (root / "file1").write_bytes(b"")
    (root / "file2").write_bytes(b"")
    s1, names, _ = _commit_scan_layer(tmp_path, root, s0b, "a")
    assert sorted(names) == ["/file1", "/file2"]
    # "mkdir dir1 && rm file1"
    os.makedirs(root / "dir1")
    os.unlink(root / "file1")
    s2, names, _ = _commit_scan_layer(tmp_path, root, s1, "b")
    assert sorted(names) == ["/.wh.file1", "/dir1/"]
    # "rm -rf dir1"
    shutil.rmtree(root / "dir1")
    s3, names, _ = _commit_scan_layer(tmp_path, root, s2, "c")
    assert names == ["/.wh.dir1"]
    # "ls ./": no files were tarred
    s4, names, pair = _commit_scan_layer(tmp_path, root, s3, "d")
    assert names == [] and pair["tar_digest"].hex() == EMPTY_TAR_TRAILER_DIGEST


def test_write_entry_cases_through_the_tar_command(tmp_path):
    """lib/tario/write_test.go:29-201 (TestWriteEntry: WriteDirectory, WriteHardLink, WriteSymlink, WriteRegularFile), the
    reference's own way of checking its writer: untar with the `tar` command and look at what lands on disk -- a 0777
    directory keeps its permission bits, a hard link (Typeflag TypeLink, Size 0, Linkname relative) shares the inode of
    its target and reads its bytes, a symlink keeps its target, a regular file its bytes and mode."""
    if not shutil.which("tar"):
        pytest.skip("no tar binary")
    src = tmp_path / "src"
    src.mkdir()
    (src / "test").write_bytes(b"test data")
    os.chmod(src / "test", 0o777)
    fd = os.open(tmp_path / "t.tar", os.O_WRONLY | os.O_CREAT, 0o644)
    with M.Layer(out_fd=fd, gzip_level=M.GZIP_OFF) as layer:
        layer.add({"relpath": "/d/dir777", "kind": M.KIND_DIR, "mode": 0o40777, "mtime_sec": 1_500_000_000})
        layer.add({"relpath": "/d/test", "kind": M.KIND_FILE, "mode": 0o100777, "size": 9, "mtime_sec": 1_500_000_001},
                  str(src / "test"))
        layer.add({"relpath": "/d/link", "kind": M.KIND_HARDLINK, "mode": 0o100777, "link_target": "d/test"})
        layer.add({"relpath": "/d/sym", "kind": M.KIND_SYMLINK, "mode": 0o120777, "link_target": "test"})
        layer.finish()
    os.close(fd)
    out = tmp_path / "out"
    out.mkdir()
    subprocess.check_call(["tar", "-xf", str(tmp_path / "t.tar"), "-C", str(out), "--no-same-owner", "-p"])
    assert os.lstat(out / "d" / "dir777").st_mode & 0o777 == 0o777
    st_t, st_l = os.lstat(out / "d" / "test"), os.lstat(out / "d" / "link")
    assert st_l.st_mode & 0o777 == 0o777 and (out / "d" / "link").read_bytes() == b"test data"
    assert st_t.st_ino == st_l.st_ino and st_l.st_nlink > 1
    assert os.readlink(out / "d" / "sym") == "test" and (out / "d" / "sym").read_bytes() == b"test data"
    assert int(st_t.st_mtime) == 1_500_000_001


def test_plain_c_commit_layer(tmp_path):
    """tests/cabi/layer_driver.c: the whole of AddLayerByScan + commitLayer from plain C against the header -- two walks,
    the snapshot diff, commit order, tar framing, both stream digests and the gzip leg -- gives the blob and the DigestPair
    the python harness gets from the same trees."""
    exe = str(tmp_path / "layer_driver")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cabi", "layer_driver.c"), "-o", exe,
                           "-L", os.path.join(root, "makisu_amd"), "-lmakisu_mi",
                           "-Wl,-rpath," + os.path.join(root, "makisu_amd")])
    a, b = tmp_path / "before", tmp_path / "after"
    a.mkdir()
    _build_tree(a)
    shutil.copytree(a, b, symlinks=True)
    os.unlink(b / "etc" / "abs-alias")                                    # an absolute target must lie under the walked root
    os.symlink(str(b / "etc" / "passwd"), b / "etc" / "abs-alias")        # (TrimRoot fails the scan otherwise, as in the reference)
    shutil.rmtree(b / "etc" / "deep")                                     # one whiteout for the subtree
    os.unlink(b / "bin" / "tool")
    (b / "etc" / "passwd").write_bytes(b"root:x:0:0\nmore\n")
    (b / "new-dir").mkdir()
    (b / "new-dir" / "f").write_bytes(os.urandom(70000))
    out = subprocess.run([exe, str(a), str(b), str(tmp_path / "c.tar.gz")], check=True, capture_output=True, text=True).stdout
    lines = dict(l.split(" ", 1) for l in out.splitlines())
    blob = (tmp_path / "c.tar.gz").read_bytes()
    raw = gzip.decompress(blob)
    t_hex, t_bytes = lines["T"].split()
    g_hex, g_bytes = lines["G"].split()
    assert hashlib.sha256(raw).hexdigest() == t_hex and int(t_bytes) == len(raw)
    assert hashlib.sha256(blob).hexdigest() == g_hex and int(g_bytes) == len(blob)
    with tarfile.open(fileobj=io.BytesIO(raw)) as tf:
        names = [m.name + ("/" if m.isdir() else "") for m in tf.getmembers()]
    assert "etc/.wh.deep" in names and "bin/.wh.tool" in names and "new-dir/f" in names and "etc/passwd" in names
    assert "etc/deep/er/x.conf" not in names
    assert int(lines["N"]) == len(names)
    # the python harness on the same pair of trees: the same bytes
    before = M.tree_walk(str(a), mode=M.TREE_SCAN, full=True)
    _, names_py, pair = _commit_scan_layer(tmp_path, b, before, "py")
    assert pair["tar_digest"].hex() == t_hex and pair["gzip_digest"].hex() == g_hex


def test_compression_level_names_like_the_reference(tmp_path):
    """lib/tario/gzip_test.go:23-27 TestSetCompressionLevelFail + the map of gzip.go:28-33 (the `--compression` flag): four
    names and an error that names the offender; "no" is still a gzip member (pgzip.NoCompression: stored blocks, longer than
    the tar), not a bare tar; every named level inflates to the same tar and TarDigest."""
    with pytest.raises(ValueError, match="invalid compression level invalid"):
        M.compression_level("invalid")
    for bad in ("", "Default", "9", None):
        with pytest.raises(ValueError, match="invalid compression level"):
            M.compression_level(bad)
    assert {n: M.compression_level(n) for n in ("no", "speed", "size", "default")} == \
        {"no": 0, "speed": 1, "size": 9, "default": M.GZIP_DEFAULT}
    root = tmp_path / "rootfs"
    root.mkdir()
    _build_tree(root)
    (root / "zeros.bin").write_bytes(bytes(300_000))
    seen = {}
    for name in ("no", "speed", "size", "default"):
        _, _, pair, blob = _write_layer(tmp_path, root, M.compression_level(name))
        assert blob[:3] == b"\x1f\x8b\x08"
        seen[name] = (pair["tar_digest"].hex(), len(blob), hashlib.sha256(gzip.decompress(blob)).hexdigest(), pair["tar_bytes"])
    assert len({v[0] for v in seen.values()}) == 1 and all(v[0] == v[2] for v in seen.values())
    assert seen["no"][1] > seen["no"][3] > seen["speed"][1] + 290_000 and seen["speed"][1] >= seen["size"][1]


def test_digest_hex_parsing_like_the_reference():
    """lib/docker/image/digest_test.go:28-35 TestDigestHexParsing: Hex() is what follows "sha256:"."""
    d = M.Digest("sha256:123abc123")
    assert d.hex() == "123abc123" and d.hex() != M.Digest.from_raw(hashlib.sha256(b"").digest()).hex()
    assert M.Digest.from_raw(hashlib.sha256(b"").digest()) == "sha256:e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"


PROBE_CHILD = r"""
import gzip, hashlib, os, sys
sys.path.insert(0, %(root)r)
import makisu_amd as M
root, out = sys.argv[1], sys.argv[2]
ents = [e for e in M.tree_walk(root, None, (), M.TREE_SCAN, full=True) if e["relpath"] not in (".", "")]
order = M.commit_order([e["relpath"] for e in ents])
fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
with M.Layer(out_fd=fd, gzip_level=M.GZIP_DEFAULT) as layer:
    for k in order:
        e = ents[k]
        layer.add(e, os.path.join(root, e["relpath"]) if e["kind"] == M.KIND_FILE else None)
    pair = layer.finish()
os.close(fd)
blob = open(out, "rb").read()
assert hashlib.sha256(blob).hexdigest() == pair["gzip_digest"].hex() and len(blob) == pair["gzip_bytes"]
assert hashlib.sha256(gzip.decompress(blob)).hexdigest() == pair["tar_digest"].hex()
print(pair["tar_digest"].hex(), pair["tar_bytes"], len(blob))
"""


def test_blocks_that_will_not_compress_are_stored(tmp_path):
    """The gzip leg (common.go:44-52, lib/tario/gzip.go:46-48) does not search incompressible blocks for matches: three 8 KiB
    samples of a 1 MiB block through the fastest level, and a block none of which shrinks goes out as stored deflate blocks
    of 65 535 bytes -- one valid member either way (python's gzip inflates it to the tar the TarDigest names), the same tar
    and digest with MI_GZIP_PROBE=0, a blob that does not depend on the thread count, and what compresses still does:
    zeros and a repeated 4 KiB pattern (period inside deflate's window) next to the random file stay small."""
    import numpy as np
    root = tmp_path / "rootfs"
    root.mkdir()
    rng = np.random.default_rng(7)
    (root / "a_random.bin").write_bytes(rng.integers(0, 256, 5 << 20, dtype=np.uint8).tobytes())
    (root / "b_zeros.bin").write_bytes(bytes(3 << 20))
    (root / "c_pattern.bin").write_bytes(rng.integers(0, 256, 4096, dtype=np.uint8).tobytes() * 768)
    (root / "d_half.bin").write_bytes(rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes() + bytes(1 << 20))
    full = b"\x00\xff\xff\x00\x00"                                   # a stored block of 65 535 bytes: BFINAL 0, LEN, ~LEN

    def run(tag, **env):
        out = tmp_path / ("blob." + tag)
        r = subprocess.run([sys.executable, "-c", PROBE_CHILD % {"root": ROOT}, str(root), str(out)],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        tar_digest, tar_bytes, n = r.stdout.split()
        return tar_digest, int(tar_bytes), out.read_bytes()

    def stored_runs(blob):                                           # two full stored blocks back to back, anywhere
        at, n = blob.find(full), 0
        while at >= 0:
            n += blob[at + 65540:at + 65545] == full
            at = blob.find(full, at + 1)
        return n

    d_on, tar_bytes, on = run("on")
    d_off, _, off = run("off", MI_GZIP_PROBE="0")
    d_one, _, one = run("one", MI_GZIP_THREADS="1")
    d_many, _, many = run("many", MI_GZIP_THREADS="7")
    assert d_on == d_off == d_one == d_many and on == one == many
    assert stored_runs(on) >= 4 * 15 and stored_runs(off) == 0       # the random file's blocks but its first (the tar header shrinks)
    assert on != off and abs(len(on) - len(off)) < tar_bytes // 1000
    # 6 MiB of the 13 will not compress; the rest (zeros, the pattern, the zero half) takes next to nothing
    assert (6 << 20) < len(on) < (6 << 20) + (300 << 10)
    # a block too short to sample (under 64 KiB) is deflated as ever: a small layer of text shrinks, one of random bytes cannot
    small = tmp_path / "small"
    small.mkdir()
    (small / "notes.txt").write_bytes(b"the quick brown fox jumps over the lazy dog\n" * 500)
    _, _, pair, blob = _write_layer(tmp_path, small, M.GZIP_DEFAULT)
    assert len(blob) < pair["tar_bytes"] // 10 and gzip.decompress(blob)[512:512 + 9] == b"the quick"
    (small / "notes.txt").write_bytes(rng.integers(0, 256, 22000, dtype=np.uint8).tobytes())
    _, _, pair, blob = _write_layer(tmp_path, small, M.GZIP_DEFAULT)
    assert pair["tar_bytes"] - 3000 < len(blob) < pair["tar_bytes"] + 200 and hashlib.sha256(gzip.decompress(blob)).hexdigest() == pair["tar_digest"].hex()
