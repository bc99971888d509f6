"""The layer-tar reader on damaged input: a base layer comes from a registry, so mi_tar_open / mi_tar_entries /
mi_tar_inflate parse bytes nobody here wrote.  Valid archives (ustar, GNU long names, PAX records, a gzip blob) with bytes
flipped, runs overwritten, the tail cut off or garbage appended must come back as entries or as an error -- never as a
crash, a hang or an out-of-bounds read (tools/asan_host_tests.sh runs this file against the ASan + UBSan build)."""
import gzip
import io
import os
import tarfile

from hypothesis import given, settings, strategies as st

import makisu_amd as M


def _archive(fmt, big_ids=True, long_names=True):
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w", format=fmt) as tf:
        for name, kind, payload in [("etc", "d", None), ("etc/passwd", "f", b"root:x:0:0\n" * 40), ("etc/link", "l", "passwd"),
                                    ("d/" + "n" * (120 if long_names else 60) + "/" + "m" * (90 if long_names else 30), "f", b"long name"),
                                    ("etc/hard", "h", "etc/passwd"),
                                    ("bin/üñî", "f", b"non-ascii" * 100), ("empty", "f", b"")]:
            ti = tarfile.TarInfo(name)
            ti.mtime, ti.mode, ti.uid, ti.gid = 1_600_000_000, 0o644, (1 << 22) if big_ids else 1000, 7
            if kind == "d":
                ti.type, ti.mode = tarfile.DIRTYPE, 0o755
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
    return buf.getvalue()


SEEDS = [_archive(tarfile.USTAR_FORMAT, False, False), _archive(tarfile.GNU_FORMAT), _archive(tarfile.PAX_FORMAT)]
SEEDS.append(gzip.compress(SEEDS[2], mtime=0))

_MUTATION = st.one_of(
    st.tuples(st.just("flip"), st.integers(0, 1 << 20), st.integers(1, 255)),
    st.tuples(st.just("fill"), st.integers(0, 1 << 20), st.integers(1, 700), st.sampled_from([0, 0xFF, 0x30, 0x37, 0x20])),
    st.tuples(st.just("cut"), st.integers(0, 1 << 20)),
    st.tuples(st.just("grow"), st.integers(1, 2000), st.sampled_from([0, 0x41])))


@settings(max_examples=1500, deadline=None, derandomize=True, database=None)
@given(st.integers(0, len(SEEDS) - 1), st.lists(_MUTATION, min_size=1, max_size=4))
def test_damaged_archives_are_entries_or_errors(tmp_path_factory, which, mutations):
    data = bytearray(SEEDS[which])
    for m in mutations:
        if m[0] == "flip":
            data[m[1] % len(data)] ^= m[2]
        elif m[0] == "fill":
            at = m[1] % len(data)
            data[at:at + m[2]] = bytes([m[3]]) * len(data[at:at + m[2]])
        elif m[0] == "cut":
            del data[max(1, m[1] % len(data)):]
        else:
            data += bytes([m[2]]) * m[1]
    p = str(tmp_path_factory.mktemp("fuzz") / "a.tar")
    with open(p, "wb") as f:
        f.write(data)
    try:
        ents = M.tar_entries(p)
    except M.MiError:
        ents = None
    if ents is not None:
        size = len(data) if which < 3 else None
        for e in ents:
            assert e["size"] >= 0 and e["kind"] in range(0, 8)
            if size is not None and e["kind"] == M.KIND_FILE and e["size"]:
                assert 0 <= e["data_offset"] and e["data_offset"] + e["size"] <= size       # a listed range lies in the file
    try:
        M.tar_inflate(p, p + ".out")
    except M.MiError:
        pass
