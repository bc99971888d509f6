"""Property test of the layer-tar READER (mi_tar_open / mi_tar_entries, csrc/mi_tar.hip -- Go's archive/tar reader as
MemFS.UpdateFromTarReader uses it, lib/snapshot/mem_fs.go:165-255): archives written by python's tarfile in all three
formats it knows (USTAR, GNU with its long-name / long-link members, PAX with extended headers) from generated member
lists -- long and non-ASCII names and link targets, ids beyond the octal fields, sub-second and pre-1970 mtimes -- must
list exactly as tarfile itself lists them, regular files with the right data offsets."""
import io
import os
import tarfile

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from test_host_tar import _compare_with_tarfile

SEG = st.text(alphabet=st.characters(blacklist_characters="/\x00\n", blacklist_categories=("Cs",)), min_size=1, max_size=60) \
    .filter(lambda s: s not in (".", "..") and not s.startswith(".wh."))
# sub-second mtimes whose str() is plain "digits.digits": tarfile writes str(float) into the pax record, and an exponent
# form ("1e-05") is no time to archive/tar's parsePAXTime -- the reference fails such an archive (test_host_tar.py)
FRAC = st.integers(0, 2 * 10**12).map(lambda v: v / 1000)
ASEG = st.text(alphabet="abcdXYZ0189-_.+", min_size=1, max_size=110).filter(lambda s: s.strip(".") != "" and not s.startswith(".wh."))


@st.composite
def members(draw, ascii_only):
    seg = ASEG if ascii_only else st.one_of(ASEG, SEG)
    out, seen = [], set()
    for _ in range(draw(st.integers(1, 8))):
        name = "/".join(draw(st.lists(seg, min_size=1, max_size=5)))
        if name in seen:
            continue
        seen.add(name)
        kind = draw(st.sampled_from(["file", "file", "dir", "sym", "hard"]))
        out.append({"name": name, "kind": kind,
                    "data": draw(st.binary(min_size=0, max_size=2000)) if kind == "file" else b"",
                    "mode": draw(st.integers(0, 0o7777)),
                    "uid": draw(st.one_of(st.integers(0, 2097151), st.integers(2097152, 2**31 - 1))) if not ascii_only
                    else draw(st.integers(0, 2097151)),
                    "mtime": draw(st.one_of(st.integers(0, 2**33 - 1), FRAC) if ascii_only else
                                  st.one_of(st.integers(0, 2**36), FRAC, st.integers(-2**31, -1))),
                    "link": "/".join(draw(st.lists(seg, min_size=1, max_size=4)))})
    return out


def _write(path, fmt, ms):
    with tarfile.open(path, "w", format=fmt) as tf:
        for m in ms:
            ti = tarfile.TarInfo(m["name"])
            ti.mode, ti.uid, ti.gid, ti.mtime = m["mode"], m["uid"], m["uid"] // 3, m["mtime"]
            if m["kind"] == "file":
                ti.size = len(m["data"])
                tf.addfile(ti, io.BytesIO(m["data"]))
                continue
            ti.type = {"dir": tarfile.DIRTYPE, "sym": tarfile.SYMTYPE, "hard": tarfile.LNKTYPE}[m["kind"]]
            if m["kind"] in ("sym", "hard"):
                ti.linkname = m["link"]
            tf.addfile(ti)


@settings(max_examples=120, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.filter_too_much,
                                                                   HealthCheck.function_scoped_fixture])
@given(ms=members(ascii_only=False), fmt=st.sampled_from([tarfile.PAX_FORMAT, tarfile.GNU_FORMAT]))
def test_reader_lists_pax_and_gnu_archives_like_tarfile(tmp_path_factory, ms, fmt):
    path = os.path.join(str(tmp_path_factory.mktemp("t")), "a.tar")
    _write(path, fmt, ms)
    _compare_with_tarfile(path)


@settings(max_examples=60, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.filter_too_much,
                                                                  HealthCheck.function_scoped_fixture])
@given(ms=members(ascii_only=True))
def test_reader_lists_ustar_archives_like_tarfile(tmp_path_factory, ms):
    path = os.path.join(str(tmp_path_factory.mktemp("t")), "a.tar")
    try:
        _write(path, tarfile.USTAR_FORMAT, ms)
    except ValueError:                      # a name or link USTAR cannot hold: tarfile refuses to write it
        return
    _compare_with_tarfile(path)
