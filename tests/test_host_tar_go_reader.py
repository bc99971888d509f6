"""The layer-tar reader where Go's archive/tar (the reader MemFS.UpdateFromTarReader drives, lib/snapshot/mem_fs.go:165-255;
Go 1.14 per the reference's Makefile:34) and POSIX / python's tarfile DISAGREE: what ends an archive, what a numeric field
may look like, which pax records count, what a global header is.  python cannot write most of these archives, so the
blocks are laid out by hand; every expectation cites the rule of reader.go / strconv.go it restates (the Go source is not
under /root/reference: restated from the published algorithm, see csrc/mi_tar.hip's header)."""
import gzip
import stat

import pytest

import makisu_amd as M


def block(name=b"f", size=0, typ=b"0", mode=0o644, uid=0, gid=0, mtime=0, link=b"", magic=b"ustar\x0000", prefix=b"",
          raw=None):
    """one 512-byte header; raw = {offset: bytes} overrides fields after everything else is laid out (checksum last)"""
    b = bytearray(512)
    b[0:len(name)] = name
    b[100:108] = b"%07o\x00" % mode
    b[108:116] = b"%07o\x00" % uid
    b[116:124] = b"%07o\x00" % gid
    b[124:136] = b"%011o\x00" % size
    b[136:148] = b"%011o\x00" % mtime
    b[156:157] = typ
    b[157:157 + len(link)] = link
    b[257:257 + len(magic)] = magic
    b[345:345 + len(prefix)] = prefix
    for at, val in (raw or {}).items():
        b[at:at + len(val)] = val
    b[148:156] = b" " * 8
    b[148:156] = b"%06o\x00 " % sum(b)
    return bytes(b)


def pad(data):
    return data + bytes(-len(data) % 512)


def pax(records, typ=b"x", name=b"PaxHeaders/f"):
    body = b"".join(records)
    return block(name, size=len(body), typ=typ) + pad(body)


def rec(key, value):
    """a well-formed pax record"""
    kv = b" " + key + b"=" + value + b"\n"
    n = len(kv) + 1
    while len(b"%d" % n) + len(kv) != n:
        n = len(b"%d" % n) + len(kv)
    return b"%d" % n + kv


END = bytes(1024)


def entries(tmp_path, raw, gz=False, name="a.tar"):
    p = str(tmp_path / name)
    with open(p, "wb") as f:
        f.write(gzip.compress(raw) if gz else raw)
    return M.tar_entries(p)


def refused(tmp_path, raw, gz=False):
    with pytest.raises(M.MiError):
        entries(tmp_path, raw, gz)


@pytest.mark.parametrize("gz", [False, True])
def test_what_ends_an_archive(tmp_path, gz):
    one = block(b"a", size=3) + pad(b"abc")
    two = one + block(b"b", size=600) + pad(b"x" * 600)
    # readHeader: "EOF is okay here; exactly 0 bytes read" -- no zero block at all
    assert [e["relpath"] for e in entries(tmp_path, two, gz)] == ["a", "b"]
    # "EOF is okay here; exactly 1 block of zeros read"
    assert [e["relpath"] for e in entries(tmp_path, two + bytes(512), gz)] == ["a", "b"]
    # "normal EOF; exactly 2 block of zeros read" -- and what follows them is never looked at
    assert [e["relpath"] for e in entries(tmp_path, two + END + b"garbage" * 100, gz)] == ["a", "b"]
    # "Zero block and then non-zero block": ErrHeader
    refused(tmp_path, one + bytes(512) + block(b"b"), gz)
    # io.ReadFull on a last block of 1..511 bytes: io.ErrUnexpectedEOF -- as the header, and after one zero block
    refused(tmp_path, two + b"\x00" * 100, gz)
    refused(tmp_path, two + b"junk", gz)
    refused(tmp_path, two + bytes(512) + bytes(100), gz)
    # the last member's padding is cut short (tryReadFull: io.EOF before the padding is complete ends the archive) ...
    cut_pad = one + block(b"b", size=600) + b"x" * 600
    assert [e["relpath"] for e in entries(tmp_path, cut_pad, gz)] == ["a", "b"]
    assert [e["relpath"] for e in entries(tmp_path, cut_pad + bytes(200), gz)] == ["a", "b"]
    # ... its DATA cut short does not (discard: io.ErrUnexpectedEOF)
    refused(tmp_path, one + block(b"b", size=600) + b"x" * 599, gz)


def test_numeric_fields_as_parse_numeric_reads_them(tmp_path):
    def one(raw, **kw):
        return entries(tmp_path, block(b"f", raw=raw, **kw) + END)[0]
    # "we need to skip leading NULs. Fields may also be padded with spaces or NULs": trimmed on BOTH sides
    assert one({108: b"\x00\x00 644 \x00"})["uid"] == 0o644
    assert one({108: b"  12\x00\x00\x00\x00"})["uid"] == 0o12
    # parseString cuts at the first NUL: what follows it is not looked at
    assert one({108: b"12\x00 34\x00\x00"})["uid"] == 0o12
    assert one({108: b"\x00" * 8})["uid"] == 0 and one({108: b" " * 8})["uid"] == 0
    # strconv.ParseUint(_, 8, 64): a non-octal digit or an inner space is an error
    refused(tmp_path, block(b"f", raw={108: b"0000128\x00"}) + END)
    refused(tmp_path, block(b"f", raw={108: b"12 34\x00\x00\x00"}) + END)
    refused(tmp_path, block(b"f", raw={136: b"1e3\x00\x00\x00\x00\x00\x00\x00\x00\x00"}) + END)
    # base-256: two's complement, at most 63 bits of magnitude
    assert one({108: b"\x80\x00\x00\x00\x00\x20\x00\x01"})["uid"] == 0x200001
    assert one({136: b"\xff" * 11 + b"\xfe"})["mtime_sec"] == -2
    assert one({124: b"\x80" + bytes(6) + b"\x01\x00\x00\x00\x00", 156: b"2"})["kind"] == 2   # (a symlink: no data to find)
    refused(tmp_path, block(b"f", raw={136: b"\x80\x00\x00\x00\x80" + bytes(7)}) + END)        # 2**63
    refused(tmp_path, block(b"f", raw={136: b"\x80\x00\x00\x01" + bytes(8)}) + END)            # beyond 64 bits
    # a negative size is an error for a member that could carry data, not for a header-only type (handleRegularFile)
    refused(tmp_path, block(b"f", raw={124: b"\xff" * 12}) + END)
    assert one({124: b"\xff" * 12, 156: b"5"})["kind"] == 0
    # devmajor / devminor are parsed -- and can fail the header -- in every format but V7
    refused(tmp_path, block(b"f", raw={329: b"zz\x00\x00\x00\x00\x00\x00"}) + END)
    refused(tmp_path, block(b"f", magic=b"ustar  \x00", raw={337: b"9\x00\x00\x00\x00\x00\x00\x00"}) + END)
    assert one({329: b"zz\x00\x00\x00\x00\x00\x00"}, magic=b"")["relpath"] == "f"


def test_pax_records_as_merge_pax_applies_them(tmp_path):
    def member(records, **kw):
        kw.setdefault("uid", 7)
        kw.setdefault("mtime", 100)
        return entries(tmp_path, pax(records) + block(b"name-in-header", **kw) + END)
    e = member([rec(b"path", b"long/name"), rec(b"uid", b"123456789"), rec(b"gid", b"5"), rec(b"mtime", b"1494882420.75")])
    assert len(e) == 1 and e[0]["relpath"] == "long/name" and e[0]["uid"] == 123456789 and e[0]["gid"] == 5
    assert e[0]["mtime_sec"] == 1494882420
    # "if v == "" { continue // Keep the original USTAR value }"
    e = member([rec(b"path", b""), rec(b"uid", b""), rec(b"mtime", b"")])
    assert e[0]["relpath"] == "name-in-header" and e[0]["uid"] == 7 and e[0]["mtime_sec"] == 100
    # a later record of the same key replaces the earlier one
    assert member([rec(b"uid", b"1"), rec(b"uid", b"2")])[0]["uid"] == 2
    # parsePAXTime: whole seconds round DOWN (a fraction of a negative time moves it away from zero); the fraction is cut
    # to nine digits first
    for text, want in ((b"-1.5", -2), (b"-0.5", -1), (b"-0.0", 0), (b"5.9999999999", 5), (b"-5.0000000001", -5),
                       (b"-5.000000001", -6), (b"7.", 7), (b"+8", 8), (b"1234567890.999999999999", 1234567890)):
        assert member([rec(b"mtime", text)])[0]["mtime_sec"] == want, text
    # ... and anything that is not [-]digits[.digits] fails the archive, for atime and ctime too
    for key, text in ((b"mtime", b"1e3"), (b"mtime", b"1.5e3"), (b"mtime", b"."), (b"mtime", b"12.3x"), (b"mtime", b" 12"),
                      (b"mtime", b"0x10"), (b"atime", b"yesterday"), (b"ctime", b"1.-5"),
                      (b"uid", b"12a"), (b"uid", b"1.0"), (b"gid", b"-"), (b"size", b"ten"), (b"size", b"-1"),
                      (b"uid", b"9223372036854775808")):
        refused(tmp_path, pax([rec(key, text)]) + block(b"f") + END)
    assert member([rec(b"uid", b"9223372036854775807")])[0]["uid"] == 0xffffffff     # fits int64; the entry holds 32 bits
    for huge in (b"99999999999999999999999", b"18446744073709551616", b"-9223372036854775809"):
        refused(tmp_path, pax([rec(b"uid", huge)]) + block(b"f") + END)
        refused(tmp_path, pax([rec(b"mtime", huge + b".5")]) + block(b"f") + END)
    assert member([rec(b"mtime", b"-9223372036854775808")])[0]["mtime_sec"] == -2**63
    assert member([b"5 a=\n"])[0]["relpath"] == "name-in-header"                     # the shortest record there is
    # parsePAXRecord: n >= 5, n within the body, a newline where n says, a key, no NUL where text is expected
    for bad in (b"4 a=\n", b"3 =\n", b"99 path=x\n", b"10 path=xy\n", b"9 path=x\n\n", b"6 =ab\n", b"7 pathx\n", b"x path=a\n",
                b"11 path=a\x00b\n", b"10 a\x00b=cd\n", b"-5 a=\n", b"10 path=xyz"):
        refused(tmp_path, pax([bad]) + block(b"f") + END)
    assert member([rec(b"comment", b"a\x00b")])[0]["relpath"] == "name-in-header"       # a NUL in another key's VALUE is fine
    # an extended header whose body fills its blocks exactly has no padding; through a gzip stream the same archive
    # lists the same
    for total in (512, 1024):
        filler = rec(b"comment", b"c" * 400)
        body = [rec(b"path", b"exact/fit")] + [filler] * (total // 512)
        short = total - sum(len(r) for r in body)
        body.append(rec(b"comment", b"c" * (short - len(rec(b"comment", b"")))))
        if sum(len(r) for r in body) != total:             # (the length's own digits moved it by one)
            body[-1] = rec(b"comment", b"c" * (short - len(rec(b"comment", b"")) - (sum(len(r) for r in body) - total)))
        assert sum(len(r) for r in body) == total
        raw = pax(body) + block(b"f", size=2) + pad(b"ok") + block(b"next") + END
        for gz in (False, True):
            e = entries(tmp_path, raw, gz)
            assert [(x["relpath"], x["size"]) for x in e] == [("exact/fit", 2), ("next", 0)]
            assert e[0]["data_offset"] == 512 + total + 512
    longname = block(b"././@LongLink", size=512, typ=b"L", magic=b"ustar  \x00") + b"L" * 511 + b"\x00"
    for gz in (False, True):
        assert [x["relpath"] for x in entries(tmp_path, longname + block(b"short") + block(b"next") + END, gz)] == ["L" * 511, "next"]
    # size: the member's data area follows the record, not the header field
    e = entries(tmp_path, pax([rec(b"size", b"700")]) + block(b"big", size=1) + pad(b"y" * 700) + block(b"next") + END)
    assert [(x["relpath"], x["size"]) for x in e] == [("big", 700), ("next", 0)]
    # of two 'x' headers before one member the LAST one counts ("paxHdrs, err = parsePAX(tr)": assigned, not merged)
    e = entries(tmp_path, pax([rec(b"path", b"first")]) + pax([rec(b"uid", b"55")]) + block(b"in-header") + END)
    assert e[0]["relpath"] == "in-header" and e[0]["uid"] == 55


def test_a_global_header_is_a_member_and_touches_nobody(tmp_path):
    g = pax([rec(b"uid", b"77"), rec(b"mtime", b"5"), rec(b"comment", b"0123abcd")], typ=b"g", name=b"pax_global_header")
    e = entries(tmp_path, g + block(b"a", uid=1, mtime=9) + block(b"b", uid=2, mtime=9) + END)
    assert [(x["relpath"], x["kind"], x["uid"], x["mtime_sec"]) for x in e] == \
        [("pax_global_header", 4, 0, 0), ("a", 1, 1, 9), ("b", 1, 2, 9)]
    assert e[0]["file_index"] == -1 and [x["file_index"] for x in e[1:]] == [0, 1]
    # mergePAX runs on the global header itself: a path record names it
    e = entries(tmp_path, pax([rec(b"path", b"elsewhere")], typ=b"g", name=b"g") + block(b"a") + END)
    assert [x["relpath"] for x in e] == ["elsewhere", "a"]
    # an 'x' (or a GNU long name) in front of a 'g' is lost with it: both live for one call of Next
    e = entries(tmp_path, pax([rec(b"path", b"renamed")]) + g + block(b"a") + END)
    assert [x["relpath"] for x in e] == ["pax_global_header", "a"]
    longname = block(b"././@LongLink", size=9, typ=b"L", magic=b"ustar  \x00") + pad(b"long-one\x00")
    e = entries(tmp_path, longname + g + block(b"a") + END)
    assert [x["relpath"] for x in e] == ["pax_global_header", "a"]
    # its records are parsed all the same
    refused(tmp_path, pax([b"4 a=\n"], typ=b"g") + block(b"a") + END)
    # the tree merge leaves such a member out (kind 4), like the devices and fifos of a layer
    root = tmp_path / "root"
    root.mkdir()
    with M.MemFS(str(root)) as fs:
        assert fs.update_from_entries(e) == 1
        assert [x["relpath"] for x in fs.entries() if x["relpath"] not in ("", ".")] == ["a"]


def test_gnu_long_names_against_pax_paths(tmp_path):
    longname = block(b"././@LongLink", size=12, typ=b"L", magic=b"ustar  \x00") + pad(b"gnu/long/one\x00")
    longlink = block(b"././@LongLink", size=9, typ=b"K", magic=b"ustar  \x00") + pad(b"gnu-link\x00")
    x = pax([rec(b"path", b"pax/path"), rec(b"linkpath", b"pax-link")])
    for head in (longname + longlink + x, x + longname + longlink):          # "if gnuLongName != "" { hdr.Name = gnuLongName }" after mergePAX
        e = entries(tmp_path, head + block(b"short", typ=b"2", link=b"short-link") + END)
        assert e[0]["relpath"] == "gnu/long/one" and e[0]["link_target"] == "gnu-link"
    # an EMPTY long name changes nothing
    empty = block(b"././@LongLink", size=1, typ=b"L", magic=b"ustar  \x00") + pad(b"\x00")
    assert entries(tmp_path, empty + block(b"short") + END)[0]["relpath"] == "short"
    assert entries(tmp_path, empty + x + block(b"short") + END)[0]["relpath"] == "pax/path"


def test_header_only_types_have_no_data_area(tmp_path):
    # isHeaderOnlyType: link, symlink, char, block, dir, fifo -- whatever their size field says
    for typ, kind, fmt in ((b"1", 3, stat.S_IFREG), (b"2", 2, stat.S_IFLNK), (b"3", 4, stat.S_IFCHR), (b"4", 4, stat.S_IFBLK),
                           (b"5", 0, stat.S_IFDIR), (b"6", 4, stat.S_IFIFO)):
        e = entries(tmp_path, block(b"h", size=1024, typ=typ, link=b"t") + block(b"next", size=2) + pad(b"ok") + END)
        assert [(x["relpath"], x["kind"]) for x in e] == [("h", kind), ("next", 1)], typ
        assert e[1]["data_offset"] == 1024 and stat.S_IFMT(e[0]["mode"]) == fmt and e[0]["mode"] & 0o7777 == 0o644
    # "Legacy archives use trailing slash for directories": typeflag NUL + a name ending in "/" is a directory -- decided on
    # the FINAL name, and then header-only too
    e = entries(tmp_path, block(b"old/", size=512, typ=b"\x00", magic=b"") + block(b"next") + END)
    assert [(x["relpath"], x["kind"]) for x in e] == [("old", 0), ("next", 1)]
    e = entries(tmp_path, pax([rec(b"path", b"made/a/dir/")]) + block(b"plain", size=512, typ=b"\x00") + block(b"next") + END)
    assert [(x["relpath"], x["kind"]) for x in e] == [("made/a/dir", 0), ("next", 1)]
    # any other unknown type keeps its data area
    e = entries(tmp_path, block(b"odd", size=600, typ=b"Z") + pad(b"z" * 600) + block(b"next") + END)
    assert [(x["relpath"], x["kind"]) for x in e] == [("odd", 4), ("next", 1)]
    refused(tmp_path, block(b"odd", size=600, typ=b"Z") + b"z" * 100)


def test_field_widths(tmp_path):
    # a name and a link target of exactly 100 bytes have no NUL: the field's width ends them
    e = entries(tmp_path, block(b"n" * 100, typ=b"2", link=b"t" * 100, magic=b"") + END)[0]
    assert e["relpath"] == "n" * 100 and e["link_target"] == "t" * 100
    # a checksum of six significant octal digits (many high bytes in the block)
    raw = block(b"\xff" * 100, typ=b"2", link=b"\xfe" * 100, magic=b"ustar  \x00")
    assert int(raw[148:154], 8) >= 0o100000
    assert len(entries(tmp_path, raw + END)) == 1
    bad = bytearray(raw)
    bad[148:149] = b"0"                                     # the leading digit counts
    refused(tmp_path, bytes(bad) + END)


def test_prefix_field_by_format(tmp_path):
    # ustar: 155 bytes at 345; star (magic ustar\0 + the trailer "tar\0"): 131 bytes, the times follow; old GNU: none
    assert entries(tmp_path, block(b"n", prefix=b"p" * 155) + END)[0]["relpath"] == "p" * 155 + "/n"
    star = block(b"n", prefix=b"q" * 131, raw={476: b"00000000017\x00", 488: b"00000000017\x00", 508: b"tar\x00"})
    assert entries(tmp_path, star + END)[0]["relpath"] == "q" * 131 + "/n"
    refused(tmp_path, block(b"n", raw={476: b"notanumber!\x00", 508: b"tar\x00"}) + END)
    assert entries(tmp_path, block(b"n", magic=b"ustar  \x00", prefix=b"ignored") + END)[0]["relpath"] == "n"
    assert entries(tmp_path, block(b"n", magic=b"", prefix=b"ignored") + END)[0]["relpath"] == "n"
