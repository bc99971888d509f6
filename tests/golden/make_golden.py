#!/usr/bin/env python3
"""Generates tests/golden/*.json from the reference's own fixtures (run in the container that
has /root/reference; the GPU box does not).  The fixture BYTES are stored base64 inside JSON so
no reference file is copied under its own name; expected digests are the constants the
reference's tests assert, cited next to each entry.

  python tests/golden/make_golden.py
"""
import base64
import hashlib
import json
import os
import zlib

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def b64(path):
    return base64.b64encode(open(path, "rb").read()).decode()


def main():
    sha = {
        "comment": "SHA-256 known answers pinned by uber/makisu's own tests and fixtures",
        "vectors": [
            {"name": "alpine_layer_blob", "file_b64": b64(REF + "/testdata/files/alpine/test_layer.tar"),
             "sha256": "393ccd5c4dd90344c9d725125e13f636ce0087c62f5ca89050faaacbb9e3ed5b",
             "pinned_by": "lib/utils/testutil/constants.go:28 SampleLayerTarDigest; "
                          "lib/registry/client_test.go:45-62 via saveLayer client.go:616-633"},
            {"name": "alpine_image_config", "file_b64": b64(REF + "/testdata/files/alpine/test_image_config"),
             "sha256": "a052f56e596097698ac74bb4b03607f2dd6bc026751878ff5d57a74bb043f098",
             "pinned_by": "lib/utils/testutil/constants.go:25 SampleImageConfigDigest; "
                          "lib/registry/pull_fixture.go:110"},
            {"name": "gnu_empty_tar_10240_zeros", "zeros": 10240,
             "sha256": "84ff92691f909a05b224e1c56abb4864f01b4f8e3c854e4bb4c7baf1d3f6d652",
             "pinned_by": "lib/docker/image/const_linux.go:18 DigestEmptyTar; digest_test.go:37-56"},
            {"name": "bsd_empty_tar_1024_zeros", "zeros": 1024,
             "sha256": "5f70bf18a086007016e948b04aed3b82103a36bea41755b6cddfaf10ace3c6ef",
             "pinned_by": "lib/docker/image/const_darwin.go:18 (= Go tar.Writer trailer)"},
            {"name": "empty", "zeros": 0,
             "sha256": "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855",
             "pinned_by": "doc comment lib/docker/image/digest.go:25"},
        ]}
    for v in sha["vectors"]:
        data = base64.b64decode(v["file_b64"]) if "file_b64" in v else bytes(v["zeros"])
        assert hashlib.sha256(data).hexdigest() == v["sha256"], v["name"]
    json.dump(sha, open(os.path.join(HERE, "sha256_reference_fixtures.json"), "w"), indent=1)

    # C1 plumbing input: testdata/build-context (28 files), with independent answers
    ctx = REF + "/testdata/build-context"
    entries = []
    for dirpath, dirnames, filenames in os.walk(ctx):
        dirnames.sort()
        for fn in sorted(filenames):
            p = os.path.join(dirpath, fn)
            rel = os.path.relpath(p, ctx)
            if os.path.islink(p):
                entries.append({"path": rel, "symlink": os.readlink(p)})
                continue
            data = open(p, "rb").read()
            entries.append({"path": rel, "b64": base64.b64encode(data).decode(),
                            "sha256": hashlib.sha256(data).hexdigest(),
                            "crc32": zlib.crc32(data), "size": len(data)})
    entries.sort(key=lambda e: e["path"])
    json.dump({"comment": "uber/makisu testdata/build-context (BASELINE.json configs[0]); answers "
                          "from hashlib/zlib, independent of the oracle", "entries": entries},
              open(os.path.join(HERE, "build_context_c1.json"), "w"), indent=1)
    # The Go-written layer tar the reference holds: testdata/files/busybox/393ccd5c.../layer.tar
    # (= gunzip of the alpine blob above).  Written by Go's archive/tar (docker save): USTAR magic
    # "ustar\x0000", empty uname/gname, "0000000\0" dev fields, checksum "%06o\0 ", Mode values with
    # the old file-type bits.  Per member: where its header sits, the SHA-256 of that 512-byte block
    # and of its data -- the framer (mi_layer_*) has to reproduce every one of them.
    import tarfile
    lp = REF + "/testdata/files/busybox/393ccd5c4dd90344c9d725125e13f636ce0087c62f5ca89050faaacbb9e3ed5b/layer.tar"
    raw = open(lp, "rb").read()
    assert zlib.decompress(base64.b64decode(sha["vectors"][0]["file_b64"]), 31) == raw
    members = []
    with tarfile.open(lp) as tf:
        for m in tf.getmembers():
            members.append({"name": m.name, "type": m.type.decode(), "header_offset": m.offset,
                            "header_sha256": hashlib.sha256(raw[m.offset:m.offset + 512]).hexdigest(),
                            "mode_field": raw[m.offset + 100:m.offset + 108].decode("latin1"),
                            "size": m.size, "linkname": m.linkname,
                            "data_offset": m.offset_data if m.isreg() else 0,
                            "data_sha256": hashlib.sha256(raw[m.offset_data:m.offset_data + m.size]).hexdigest()
                            if m.isreg() else None})
    json.dump({"comment": "headers of the reference's Go-written layer tar (testdata/files/busybox/393ccd5c.../layer.tar, "
                          "the gunzip of sha256_reference_fixtures.json[alpine_layer_blob]); listed with python tarfile",
               "tar_bytes": len(raw), "tar_sha256": hashlib.sha256(raw).hexdigest(),
               "pinned_by": "computed from the fixture; the blob's digest is lib/utils/testutil/constants.go:28",
               "members": members},
              open(os.path.join(HERE, "go_layer_tar_members.json"), "w"), indent=0)
    print("wrote", len(sha["vectors"]), "sha vectors,", len(entries), "context entries and", len(members), "tar members")


if __name__ == "__main__":
    main()
