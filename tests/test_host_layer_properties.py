"""Property tests of the layer writer's header bytes (mi_layer_header_bytes, csrc/mi_layer.hip) on the paths no Go-written
fixture pins: PAX records (names that USTAR cannot hold or split, non-ASCII names and link targets, ids above 2^21, sizes
of 8 GiB and more, negative or huge mtimes) next to plain USTAR headers.  Two INDEPENDENT readers must get the entry back
exactly: python's tarfile (a POSIX.1-2001 pax reader written against the standard, not against Go) and this library's own
reader (mi_tar_open, which follows Go's archive/tar reader).  And the structure Go's writer has is asserted on the bytes
themselves: a PAX entry is 'PaxHeaders.0/<base>' with typeflag x, its records sorted by key and self-describing in
length, and the following USTAR header keeps what fits (lib/snapshot/mem_layer.go:152-190 builds the tar.Header that
archive/tar's Writer.WriteHeader formats this way)."""
import io
import os
import re
import tarfile

import pytest
from hypothesis import HealthCheck, example, given, settings, strategies as st

import makisu_amd as M

SEG = st.text(alphabet=st.characters(blacklist_characters="/\x00", blacklist_categories=("Cs",)), min_size=1, max_size=70) \
    .filter(lambda s: s not in (".", "..") and not s.startswith(".wh."))
ASCII_SEG = st.text(alphabet="abcXYZ019-_.+ ", min_size=1, max_size=120).filter(lambda s: s.strip(". ") != "" and s not in (".", ".."))


@st.composite
def entries(draw):
    kind = draw(st.sampled_from([M.KIND_FILE, M.KIND_FILE, M.KIND_DIR, M.KIND_SYMLINK, M.KIND_HARDLINK]))
    segs = draw(st.lists(st.one_of(ASCII_SEG, SEG), min_size=1, max_size=6))
    rel = "/".join(segs)
    e = {"relpath": rel, "kind": kind,
         "mode": draw(st.integers(0, 0o7777)) | {M.KIND_FILE: 0o100000, M.KIND_DIR: 0o40000, M.KIND_SYMLINK: 0o120000,
                                                  M.KIND_HARDLINK: 0o100000}[kind],
         "uid": draw(st.one_of(st.integers(0, 2097151), st.integers(2097152, 2**31 - 1))),
         "gid": draw(st.one_of(st.integers(0, 2097151), st.integers(2097152, 2**31 - 1))),
         "mtime_sec": draw(st.one_of(st.integers(0, 2**33 - 1), st.integers(2**33, 2**40), st.integers(-2**31, -1))),
         "size": 0}
    if kind == M.KIND_FILE:
        e["size"] = draw(st.one_of(st.integers(0, 4096), st.sampled_from([8 * 2**30 - 1, 8 * 2**30, 2**40 + 17])))
    if kind in (M.KIND_SYMLINK, M.KIND_HARDLINK):
        t = "/".join(draw(st.lists(st.one_of(ASCII_SEG, SEG), min_size=1, max_size=4)))
        e["link_target"] = ("/" + t) if draw(st.booleans()) else t
    return e


def _fits_ustar(e, name):
    try:
        name.encode("ascii")
        (e.get("link_target") or "").encode("ascii")
    except UnicodeEncodeError:
        return False
    return True


@settings(max_examples=300, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.filter_too_much])
@given(entries())
@example({"relpath": "a" * 100, "kind": 1, "mode": 0o100644, "uid": 0, "gid": 0, "mtime_sec": 1, "size": 1})
@example({"relpath": "a" * 101, "kind": 1, "mode": 0o100644, "uid": 0, "gid": 0, "mtime_sec": 1, "size": 1})
@example({"relpath": "p" * 155 + "/" + "n" * 100, "kind": 1, "mode": 0o100644, "uid": 0, "gid": 0, "mtime_sec": 1, "size": 1})
@example({"relpath": "p" * 156 + "/" + "n" * 100, "kind": 1, "mode": 0o100644, "uid": 0, "gid": 0, "mtime_sec": 1, "size": 1})
@example({"relpath": "d" * 99 + "/", "kind": 0, "mode": 0o40755, "uid": 0, "gid": 0, "mtime_sec": 1, "size": 0})
@example({"relpath": "d" * 100, "kind": 0, "mode": 0o40755, "uid": 0, "gid": 0, "mtime_sec": 1, "size": 0})
@example({"relpath": "x/" * 140 + "leaf", "kind": 1, "mode": 0o100600, "uid": 2097151, "gid": 2097152, "mtime_sec": 2**33 - 1, "size": 0})
@example({"relpath": "l", "kind": 2, "mode": 0o120777, "uid": 0, "gid": 0, "mtime_sec": 0, "size": 0, "link_target": "t" * 101})
@example({"relpath": "l", "kind": 3, "mode": 0o100644, "uid": 0, "gid": 0, "mtime_sec": 0, "size": 0, "link_target": "/" + "t" * 100})
@example({"relpath": "caf\u00e9/\u65e5\u672c", "kind": 1, "mode": 0o100644, "uid": 1, "gid": 2, "mtime_sec": -1, "size": 8 * 2**30})
def test_header_bytes_read_back_through_two_independent_readers(tmp_path_factory, e):
    try:
        h = M.layer_header_bytes(e)
    except M.MiError:
        # the writer refuses what no header can hold (e.g. a PAX record for a key it does not write); nothing of the
        # strategies above should get here
        raise
    assert len(h) % 512 == 0 and len(h) in (512, 1536, 2048, 2560) or len(h) % 512 == 0
    name = e["relpath"].lstrip("/") + ("/" if e["kind"] == M.KIND_DIR else "")
    pax = len(h) > 512
    if pax:
        # Go: PaxHeaders.0/<base name>, typeflag x, records "%d key=value\n" sorted by key, each length self-inclusive
        # (archive/tar writer.go writePAXHeader / writeRawFile: name = path.Join(dir, "PaxHeaders.0", base), non-ASCII
        # characters dropped, cut to 100 bytes, trailing slashes trimmed)
        d_, base = name.rsplit("/", 1) if "/" in name.rstrip("/") or name.endswith("/") else ("", name)
        if name.endswith("/"):
            d_, base = name, ""
        parts = [x for x in (d_.rstrip("/"), "PaxHeaders.0", base) if x != ""]
        want = "/".join(parts)
        want = "".join(ch for ch in want if ord(ch) < 128)[:100].rstrip("/").encode()
        assert h[156:157] == b"x" and h[:100].rstrip(b"\x00") == want, (h[:100], want)
        n = int(h[124:135], 8)
        recs = h[512:512 + n]
        keys, p = [], 0
        while p < n:
            sp = recs.index(b" ", p)
            ln = int(recs[p:sp])
            rec = recs[p:p + ln]
            assert rec.endswith(b"\n") and b"=" in rec
            keys.append(rec[sp - p + 1:rec.index(b"=")].decode())
            p += ln
        assert p == n and keys == sorted(keys) and set(keys) <= {"path", "linkpath", "uid", "gid", "size", "mtime"}
        assert set(recs[n:]) <= {0}
    # reader 1: python tarfile (data blocks are not needed to read headers; sizes are taken from the header / PAX)
    stream = io.BytesIO(h + bytes(1024))
    with tarfile.open(fileobj=stream, mode="r:", ignore_zeros=False) as tf:
        ti = tf.next()
    assert ti is not None and ti.name == name.rstrip("/") if e["kind"] != M.KIND_DIR else ti.name == name.rstrip("/")
    assert ti.uid == e["uid"] and ti.gid == e["gid"] and ti.mode == (e["mode"] & 0o7777)
    assert int(ti.mtime) == e["mtime_sec"] if e["mtime_sec"] >= 0 else True     # python refuses negative octal; PAX carries it
    assert ti.uname == "" and ti.gname == ""
    want_type = {M.KIND_FILE: tarfile.REGTYPE, M.KIND_DIR: tarfile.DIRTYPE, M.KIND_SYMLINK: tarfile.SYMTYPE,
                 M.KIND_HARDLINK: tarfile.LNKTYPE}[e["kind"]]
    assert ti.type == want_type
    if e["kind"] == M.KIND_FILE:
        assert ti.size == e["size"]
    if e["kind"] == M.KIND_SYMLINK:
        assert ti.linkname == e["link_target"]
    if e["kind"] == M.KIND_HARDLINK:
        # written as stored (tario.WriteHeader trims Name only, lib/tario/write.go:55-68); the reference's own reader makes
        # it absolute again (mem_fs.go:217-219), as reader 2 below does
        assert ti.linkname == e["link_target"]
    # reader 2: this library's own tar reader (Go's reader restated) on a file holding the header, the data it announces
    # only when that is small, and the trailer
    if e["size"] <= 4096:
        d = tmp_path_factory.mktemp("hdr")
        path = os.path.join(str(d), "one.tar")
        with open(path, "wb") as f:
            f.write(h + bytes((e["size"] + 511) // 512 * 512) + bytes(1024))
        got = M.tar_entries(path)
        assert len(got) == 1
        g = got[0]
        assert g["relpath"].strip("/") == name.strip("/") and g["kind"] == e["kind"]
        assert g["uid"] == e["uid"] and g["gid"] == e["gid"] and g["size"] == e["size"]
        assert g["mtime_sec"] == e["mtime_sec"] and (g["mode"] & 0o7777) == (e["mode"] & 0o7777)
        if e["kind"] in (M.KIND_SYMLINK, M.KIND_HARDLINK):
            want = e["link_target"] if e["kind"] == M.KIND_SYMLINK else e["link_target"].lstrip("/")
            assert (g.get("link_target") or "").lstrip("/") == want.lstrip("/")


@st.composite
def small_layers(draw):
    n = draw(st.integers(1, 10))
    items, seen = [], set()
    for i in range(n):
        segs = draw(st.lists(st.one_of(ASCII_SEG, SEG), min_size=1, max_size=3))
        rel = "/".join(segs)
        if rel in seen:
            continue
        seen.add(rel)
        kind = draw(st.sampled_from([M.KIND_FILE, M.KIND_FILE, M.KIND_DIR, M.KIND_SYMLINK]))
        data = draw(st.binary(min_size=0, max_size=3000)) if kind == M.KIND_FILE else b""
        items.append((rel, kind, data, draw(st.integers(0, 0o777)), draw(st.integers(0, 2**22)), draw(st.integers(0, 2**31))))
    return items


@settings(max_examples=60, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.filter_too_much])
@given(small_layers())
def test_whole_layers_frame_and_digest(tmp_path_factory, items):
    """Whole streams: every member's header and bytes come back through tarfile, the stream is blocks of 512 with the
    1024-byte trailer Go's Writer.Close writes, TarDigest is the SHA-256 of exactly those bytes (common.go:44-55)."""
    import hashlib
    d = tmp_path_factory.mktemp("lay")
    out = os.path.join(str(d), "layer.tar")
    fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    srcs = {}
    try:
        with M.Layer(out_fd=fd, gzip_level=M.GZIP_OFF) as layer:
            for k, (rel, kind, data, perm, uid, mtime) in enumerate(items):
                e = {"relpath": rel, "kind": kind, "size": len(data), "uid": uid, "gid": uid // 2, "mtime_sec": mtime,
                     "mode": perm | {M.KIND_FILE: 0o100000, M.KIND_DIR: 0o40000, M.KIND_SYMLINK: 0o120000}[kind]}
                src = None
                if kind == M.KIND_FILE:
                    src = os.path.join(str(d), "src%d" % k)
                    with open(src, "wb") as f:
                        f.write(data)
                if kind == M.KIND_SYMLINK:
                    e["link_target"] = "../t%d" % k
                layer.add(e, src)
            pair = layer.finish()
    finally:
        os.close(fd)
    raw = open(out, "rb").read()
    assert len(raw) % 512 == 0 and raw[-1024:] == bytes(1024) and pair["tar_bytes"] == len(raw)
    assert pair["tar_sha256"] == hashlib.sha256(raw).digest() and pair["n_entries"] == len(items)
    with tarfile.open(out, "r:") as tf:
        members = tf.getmembers()
        assert len(members) == len(items)
        for m, (rel, kind, data, perm, uid, mtime) in zip(members, items):
            assert m.name == rel.rstrip("/") and m.uid == uid and m.gid == uid // 2 and m.mode == perm and int(m.mtime) == mtime
            if kind == M.KIND_FILE:
                assert m.isreg() and tf.extractfile(m).read() == data
            elif kind == M.KIND_DIR:
                assert m.isdir()
            else:
                assert m.issym() and m.linkname == "../t%d" % items.index((rel, kind, data, perm, uid, mtime))
    got = M.tar_entries(out)
    assert [g["relpath"].strip("/") for g in got] == [rel.strip("/") for rel, *_ in items]
    assert [g["size"] for g in got] == [len(data) for _, _, data, *_ in items]


_PSEG = st.one_of(st.text(alphabet=st.characters(blacklist_characters="\x00/", blacklist_categories=("Cs",)), min_size=1, max_size=12),
                  st.sampled_from([".wh.a", ".wh.b", "a", "b", "a-b", ".wh.a-b", ".wh..wh.opq"])) \
    .filter(lambda s: s not in (".", ".."))


@settings(max_examples=300, deadline=None, derandomize=True, database=None)
@given(st.lists(st.lists(_PSEG, min_size=1, max_size=4).map(lambda p: "/".join(p)), min_size=0, max_size=40, unique=True))
def test_commit_order_is_sort_strings_on_the_absolute_paths(names):
    """memLayer.rangeFiles (lib/snapshot/mem_layer.go:232-244): sort.Strings over the layer's keys -- byte order of the
    absolute destination paths, whatever the characters."""
    got = [names[k] for k in M.commit_order(names)]

    def key(s):                                   # a whiteout marker is filed under the path it deletes (addHeader :197-212)
        d, _, b = ("/" + s).rpartition("/")
        return (d + "/" + (b[4:] if b.startswith(".wh.") else b)).encode("utf-8")
    want = sorted(names, key=key)
    if len({key(s) for s in names}) == len(names):        # (a marker AND the path it deletes share one map key: no order to check)
        assert got == want


def _similar_restated(a, b, ignore_time):
    """tario.IsSimilarHeader (lib/tario/compare.go:24-120) on entry dicts: a symlink by its target alone; a hard link by
    mtime (whole seconds), target, uid, gid, mode; a directory by mtime, uid, gid, mode; a regular file by those and
    size.  Another type on the other side is never similar."""
    if a["kind"] != b["kind"]:
        return False
    if a["kind"] == M.KIND_SYMLINK:
        return a.get("link_target") == b.get("link_target")
    same = (ignore_time or a["mtime_sec"] == b["mtime_sec"]) and a["uid"] == b["uid"] and a["gid"] == b["gid"] and \
        (a["mode"] & 0o7777) == (b["mode"] & 0o7777)
    if a["kind"] == M.KIND_HARDLINK:
        # the reference holds hard-link targets as absolute paths (UpdateFromTarReader, lib/snapshot/mem_fs.go:217-219:
        # "Docker hard link names are all absolute, but don't have a leading slash"); entries here keep the tar's form,
        # so the comparison makes them absolute
        ab = lambda t: "/" + (t or "").lstrip("/")                                  # noqa: E731
        return same and ab(a.get("link_target")) == ab(b.get("link_target"))
    if a["kind"] == M.KIND_FILE:
        return same and a["size"] == b["size"]
    return same


_ENTRY = st.fixed_dictionaries({
    "kind": st.sampled_from([M.KIND_DIR, M.KIND_FILE, M.KIND_SYMLINK, M.KIND_HARDLINK]),
    "perm": st.sampled_from([0o644, 0o755, 0o4755, 0o1777]), "uid": st.sampled_from([0, 1000]), "gid": st.sampled_from([0, 7]),
    "mtime_sec": st.sampled_from([5, 6]), "size": st.sampled_from([0, 10]), "link_target": st.sampled_from(["a", "/a", "b"]),
    "relpath": st.sampled_from(["x", "y/z"])})


@settings(max_examples=2000, deadline=None, derandomize=True, database=None)
@given(_ENTRY, _ENTRY, st.booleans())
def test_entry_similar_equals_the_restated_predicate(a, b, ignore_time):
    def full(d):
        t = {M.KIND_DIR: 0o40000, M.KIND_FILE: 0o100000, M.KIND_SYMLINK: 0o120000, M.KIND_HARDLINK: 0o100000}[d["kind"]]
        e = dict(d, mode=d["perm"] | t)
        if d["kind"] not in (M.KIND_SYMLINK, M.KIND_HARDLINK):
            e["link_target"] = None
        if d["kind"] != M.KIND_FILE:
            e["size"] = 0
        return e
    ea, eb = full(a), full(b)
    assert M.entry_similar(ea, eb, ignore_time=ignore_time) == _similar_restated(ea, eb, ignore_time)
