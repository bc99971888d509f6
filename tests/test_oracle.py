"""CPU tests of the oracle itself: pinned against the reference's golden vectors (SHA-256),
zlib (CRC32), an independent pure-Python statement of the Gear spec, and its own committed
known-answer file.  Gear cut points are UNPINNED w.r.t. the reference (it has no CDC)."""
import base64
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
SEED = 0x4D414B49
M64 = (1 << 64) - 1


def _fixtures():
    return json.load(open(os.path.join(GOLD, "sha256_reference_fixtures.json")))["vectors"]


@pytest.mark.parametrize("shani", [False, True])
def test_sha256_reference_fixtures(oracle, shani):
    for v in _fixtures():
        data = base64.b64decode(v["file_b64"]) if "file_b64" in v else bytes(v["zeros"])
        assert oracle.sha256(data, shani).hex() == v["sha256"], v["name"]


def test_sha256_nist_vectors(oracle):
    kats = {b"abc": "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad",
            b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq":
                "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1",
            b"a" * 1000000: "cdc76e5c9914fb9281a1c7e284d73e67f1809a48a497200e046d39ccc7112cd0"}
    for msg, want in kats.items():
        assert oracle.sha256(msg, False).hex() == want
        assert oracle.sha256(msg, True).hex() == want


def test_sha256_every_padding_boundary(oracle):
    rng = np.random.default_rng(3)
    for n in list(range(0, 200)) + [255, 256, 257, 1000, 4095, 4096, 4097, 65536, 100003]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        want = hashlib.sha256(d).digest()
        assert oracle.sha256(d, False) == want
        assert oracle.sha256(d, True) == want


def test_reference_live_fixtures_if_present(oracle):
    # in the build container the reference tree is mounted: hash its fixtures in place too
    ref = "/root/reference/testdata/files"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not mounted (GPU box)")
    blob = open(ref + "/alpine/test_layer.tar", "rb").read()
    assert oracle.sha256(blob).hex() == "393ccd5c4dd90344c9d725125e13f636ce0087c62f5ca89050faaacbb9e3ed5b"
    layer = open(ref + "/busybox/393ccd5c4dd90344c9d725125e13f636ce0087c62f5ca89050faaacbb9e3ed5b/layer.tar", "rb").read()
    assert zlib.decompress(blob, 31) == layer          # the layer is the gunzip of the blob
    assert oracle.sha256(layer).hex() == "4ac76077f2c741c856a2419dfdb0804b18e48d2e1a9ce9c6a3f0605a2078caba"


def test_crc32_matches_zlib_and_combine(oracle):
    rng = np.random.default_rng(5)
    for n in [0, 1, 7, 8, 9, 63, 64, 1000, 65536, 99991]:
        a = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        b = rng.integers(0, 256, (n * 13) % 5000, dtype=np.uint8).tobytes()
        assert oracle.crc32(a) == zlib.crc32(a)
        assert oracle.crc32(b, oracle.crc32(a)) == zlib.crc32(a + b)      # running form
        assert oracle.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)


def test_c1_build_context(oracle):
    """BASELINE.json configs[0]: the reference's testdata/build-context on the CPU path --
    per-file SHA-256 / CRC32 against hashlib/zlib answers, sorted commit order
    (lib/snapshot/mem_layer.go:232-244), and the running CRC of checksumPathContents
    (lib/builder/step/add_copy_step.go:194-238) rebuilt with crc32_combine."""
    ctx = json.load(open(os.path.join(GOLD, "build_context_c1.json")))["entries"]
    assert [e["path"] for e in ctx] == sorted(e["path"] for e in ctx)
    files = [e for e in ctx if "b64" in e]
    assert len(ctx) == 28 and sum(e["size"] for e in files) == 10355
    blobs = [base64.b64decode(e["b64"]) for e in files]
    data = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    sizes = [len(b) for b in blobs]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    p = oracle.CdcParams(SEED, 13, 2048, 65536)
    rf, rc = oracle.scan_batch(data, offs, sizes, p)
    for e, row in zip(files, rf):
        assert row["file_sha256"].tobytes().hex() == e["sha256"], e["path"]
        assert int(row["crc32"]) == e["crc32"], e["path"]
    # running checksum the way checksumPathContents feeds it: relpath then bytes / link target
    running = zlib.crc32(b"seedCOPY . /x")
    combined = running
    for e in ctx:
        name = e["path"].encode()
        payload = base64.b64decode(e["b64"]) if "b64" in e else e["symlink"].encode()
        running = zlib.crc32(payload, zlib.crc32(name, running))
        combined = oracle.crc32_combine(combined, oracle.crc32(name), len(name))
        combined = oracle.crc32_combine(combined, oracle.crc32(payload), len(payload))
    assert combined == running
    assert "%x" % running == format(running, "x")       # cacheID formatting: unpadded hex (:119)


# ---- Gear: independent pure-Python statement of the spec (small inputs only) -------------
def _splitmix_table(seed):
    out, s = [], seed
    for _ in range(256):
        s = (s + 0x9E3779B97F4A7C15) & M64
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        out.append(z ^ (z >> 31))
    return out


def _py_cdc(data, seed, mask_bits, mn, mx):
    g = _splitmix_table(seed)
    cands = []
    for i in range(len(data)):
        h = 0
        for k in range(min(i + 1, 64)):               # literal window sum, no rolling
            h = (h + (g[data[i - k]] << k)) & M64
        if mask_bits == 0 or (h >> (64 - mask_bits)) == 0:
            cands.append(i + 1)
    ends, last = [], 0
    for e in cands:
        while e - last > mx:
            last += mx
            ends.append(last)
        if e - last >= mn:
            last = e
            ends.append(e)
    while len(data) - last > mx:
        last += mx
        ends.append(last)
    if len(data) > last:
        ends.append(len(data))
    return cands, ends


def test_gear_table_and_window_definition(oracle):
    assert [int(x) for x in oracle.gear_table(SEED)] == _splitmix_table(SEED)
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()
    for mask_bits, mn, mx in [(4, 64, 256), (6, 64, 1024), (0, 64, 128), (8, 100, 300)]:
        cands, ends = _py_cdc(data, SEED, mask_bits, mn, mx)
        assert [int(x) for x in oracle.gear_candidates(data, SEED, mask_bits)] == cands
        p = oracle.CdcParams(SEED, mask_bits, mn, mx)
        assert [int(x) for x in oracle.cdc_two_phase(data, p)] == ends
        assert [int(x) for x in oracle.cdc_classic(data, p)] == ends


def test_gear_halo_equals_continuous_scan(oracle):
    data = oracle.synth_fill(SEED, 1, 0, 50000).tobytes()
    whole = oracle.gear_candidates(data, SEED, 8)
    for split in (1, 63, 64, 65, 20000):
        a = oracle.gear_candidates(data[:split], SEED, 8)
        b = oracle.gear_candidates(data[split:], SEED, 8, halo=data[max(0, split - 63):split])
        assert np.array_equal(np.concatenate([a, b + split]), whole), split


@pytest.mark.parametrize("mask_bits,mn,mx", [(13, 2048, 65536), (10, 1024, 8192), (6, 64, 4096),
                                             (0, 64, 64), (32, 2048, 65536), (16, 4096, 262144)])
def test_two_phase_equals_classic(oracle, mask_bits, mn, mx):
    p = oracle.CdcParams(SEED, mask_bits, mn, mx)
    for cid, n in [(1, 1 << 20), (2, 65536), (3, 1), (4, mn), (5, mx + 1), (6, 0)]:
        data = oracle.synth_fill(SEED, cid, 0, n)
        a, b = oracle.cdc_two_phase(data, p), oracle.cdc_classic(data, p)
        assert np.array_equal(a, b)
        if n:
            lens = np.diff(np.concatenate([[0], a]))
            assert a[-1] == n and lens.max() <= mx and (lens[:-1] >= mn).all()
    for data in (bytes(300000), b"ab" * 100000, oracle.synth_fill(SEED, 9, 0, 4096).tobytes() * 50):
        assert np.array_equal(oracle.cdc_two_phase(data, p), oracle.cdc_classic(data, p))


def test_gear_known_answers(oracle):
    """Committed KATs (made by tests/golden/make_gear_kat.py from this same oracle): they
    freeze the spec so an accidental change of table / window / selection shows up."""
    kat = json.load(open(os.path.join(GOLD, "gear_cdc_kat.json")))
    for v in kat["vectors"]:
        p = oracle.CdcParams(v["gear_seed"], v["mask_bits"], v["min_size"], v["max_size"])
        data = oracle.synth_fill(v["data_seed"], v["content_id"], 0, v["size"])
        ends = oracle.cdc_two_phase(data, p)
        assert [int(x) for x in ends] == v["ends"], v["name"]
        files, chunks = oracle.scan_batch(data, [0], [v["size"]], p)
        assert files["chunk_root"][0].tobytes().hex() == v["chunk_root"], v["name"]
        assert hashlib.sha256(data.tobytes()).hexdigest() == v["sha256"]   # generator pinned too


def test_synth_fill_offsets(oracle):
    whole = oracle.synth_fill(7, 3, 0, 1000)
    for off, n in [(0, 1), (1, 7), (5, 100), (8, 8), (993, 7)]:
        assert np.array_equal(oracle.synth_fill(7, 3, off, n), whole[off:off + n])
    assert not np.array_equal(oracle.synth_fill(7, 4, 0, 64), whole[:64])


def test_scan_batch_threads_and_dedup(oracle):
    p = oracle.CdcParams(SEED, 13, 2048, 65536)
    n = 64
    cids = [i % 40 for i in range(n)]                 # files 40.. repeat files 0..23
    data = np.concatenate([oracle.synth_fill(SEED, c, 0, 65536) for c in cids])
    offs, sizes = np.arange(n) * 65536, [65536] * n
    f1, c1 = oracle.scan_batch(data, offs, sizes, p, True, 1)
    f4, c4 = oracle.scan_batch(data, offs, sizes, p, False, 4)
    assert np.array_equal(c1, c4) and np.array_equal(f1, f4)
    for row in c1:
        blob = data[int(offs[row["file_index"]] + row["offset"]):][:int(row["length"])]
        assert row["sha256"].tobytes() == hashlib.sha256(blob.tobytes()).digest()
    dup = c1["dup_of"]
    assert (dup[c1["file_index"] < 40] == -1).all()
    assert (dup[c1["file_index"] >= 40] >= 0).all()
    d2, uniq = oracle.dedup(c1["sha256"])
    assert np.array_equal(d2, dup) and uniq == (dup < 0).sum()
    for f in range(n):
        rows = c1[c1["file_index"] == f]
        assert f1["chunk_root"][f].tobytes() == hashlib.sha256(rows["sha256"].tobytes()).digest()


def test_layer_scan_is_sha256_of_a_readable_tar(oracle, tmp_path):
    """The reference-shaped scanner: the digest equals hashlib over the same stream, and the
    stream is a tar archive python's tarfile can list (Go archive/tar byte parity is UNPINNED)."""
    import io
    import tarfile
    blobs = [b"hello\n", bytes(range(256)) * 3, b"", b"x" * 1000]
    data = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    sizes = [len(b) for b in blobs]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    n, digest = oracle.layer_scan(data, offs, sizes)
    stream = bytearray()
    for i, b in enumerate(blobs):
        stream += oracle.tar_header("f%08d" % i, len(b))
        stream += b + bytes((512 - len(b) % 512) % 512)
    stream += bytes(1024)
    assert n == len(stream) and digest == hashlib.sha256(stream).digest()
    with tarfile.open(fileobj=io.BytesIO(bytes(stream))) as tf:
        members = tf.getmembers()
        assert [m.name for m in members] == ["f%08d" % i for i in range(4)]
        assert [m.size for m in members] == sizes
        assert tf.extractfile(members[1]).read() == blobs[1]


def test_threaded_dedup_equals_serial(oracle):
    """mi_ref_dedup_mt (bucketed, n threads) against the single-bucket-sort form and a python dict."""
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (3000, 32), dtype=np.uint8)
    rows = base[rng.integers(0, 3000, 20000)]
    rows[::7, 0] = 0                                   # crowd one bucket
    d1, u1 = oracle.dedup(rows)
    for t in (1, 2, 5, 16):
        dt, ut = oracle.dedup_mt(rows, t)
        assert ut == u1 and np.array_equal(dt, d1)
    seen, want = {}, np.empty(len(rows), dtype=np.int64)
    for i, r in enumerate(rows):
        k = r.tobytes()
        want[i] = seen.get(k, -1)
        seen.setdefault(k, i)
    assert np.array_equal(d1, want) and u1 == len(seen)


def test_scan_synthetic_equals_scan_batch(oracle):
    """Generating inside the workers (full-size configs) == scanning a host copy, any thread count."""
    seed = 0x4D414B49 + 2
    sizes = [65536, 1, 0, 300000, 70000, 65536, 2048, 100]
    cids = [5, 6, 7, 8, 5, 9, 10, 6]
    p = oracle.CdcParams(0x4D414B49, 13, 2048, 65536)
    data = np.concatenate([oracle.synth_fill(seed, c, 0, n) for c, n in zip(cids, sizes)])
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    rf, rc = oracle.scan_batch(data, offs, sizes, p, True, 1, 0)
    for t in (1, 4):
        sf, sc, nu = oracle.scan_synthetic(seed, cids, sizes, p, True, t, 0)
        assert np.array_equal(sf, rf) and np.array_equal(sc, rc)
        assert nu == int((rc["dup_of"] < 0).sum())
    ph = oracle.last_phase_seconds()
    assert set(ph) == {"scan_s", "gather_s", "dedup_s"} and ph["scan_s"] > 0


def test_synth_fill_unaligned_ranges(oracle):
    whole = oracle.synth_fill(7, 3, 0, 1000)
    for off, n in ((0, 0), (1, 7), (3, 13), (8, 64), (5, 900), (999, 1)):
        assert np.array_equal(oracle.synth_fill(7, 3, off, n), whole[off:off + n])


def test_chunk_root_is_a_fanout_64_tree(oracle):
    """The root definition (ours; the reference has no per-file content digest): flat SHA-256 over the
    concatenated digests up to 64 chunks, above that 64-wide node digests, repeated."""
    rng = np.random.default_rng(3)
    for n in (0, 1, 64, 65, 4096, 4097, 70000):
        d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        nodes = [d[i].tobytes() for i in range(n)]
        while len(nodes) > 64:
            nodes = [hashlib.sha256(b"".join(nodes[i:i + 64])).digest() for i in range(0, len(nodes), 64)]
        assert oracle.chunk_root(d) == hashlib.sha256(b"".join(nodes)).digest()
