"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Cut points: parity UNPINNED w.r.t. the reference (it has no CDC);
the oracle is this repo's restatement of its own Gear spec.  SHA-256: pinned by the
reference's fixtures (tests/test_oracle.py) and checked here on the GPU as well.
"""
import hashlib

import numpy as np
import pytest

try:
    # Several tests below use torch next to the engine.  PyTorch-ROCm wheels bundle their own HIP
    # runtime (SONAME libamdhip64.so.7); loaded first it also serves libmakisu_mi.so, loaded
    # second it would be a second runtime that finds no GPU -- so torch comes first in any
    # process that uses both (bench.py does the same).
    import torch  # noqa: F401
except ImportError:          # CPU-only collection without torch: the GPU tests are skipped anyway
    torch = None

pytestmark = pytest.mark.gpu

SEED = 0x4D414B49


def _params(oracle, eng):
    c = eng.cfg
    return oracle.CdcParams(c.gear_seed, c.mask_bits, c.min_size, c.max_size)


def _check_batch(oracle, eng, blobs, flags_sha=False):
    """Runs blobs through the engine and the oracle; asserts every column is identical."""
    with eng.batch(len(blobs), sum(len(b) for b in blobs)) as b:
        for i, blob in enumerate(blobs):
            b.add_bytes(blob, tag=1000 + i)
        b.run()
        files, chunks = b.files().copy(), b.chunks().copy()
        back = b.read_back().copy()
    data = np.frombuffer(b"".join(bytes(x) for x in blobs), dtype=np.uint8)
    assert np.array_equal(back, data), "staged bytes differ from what was added"
    sizes = np.array([len(x) for x in blobs], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64) if len(blobs) else sizes
    rf, rc = oracle.scan_batch(data if data.size else np.zeros(1, np.uint8), offs, sizes,
                               _params(oracle, eng))
    assert len(chunks) == len(rc)
    assert np.array_equal(files["n_chunks"], rf["n_chunks"])
    assert np.array_equal(files["first_chunk"], rf["first_chunk"])
    assert np.array_equal(files["size"], sizes)
    assert np.array_equal(files["user_tag"], 1000 + np.arange(len(blobs)))
    assert np.array_equal(chunks["file_index"], rc["file_index"])
    assert np.array_equal(chunks["offset"], rc["offset"]), "cut points differ"
    assert np.array_equal(chunks["length"], rc["length"]), "cut points differ"
    assert np.array_equal(chunks["sha256"], rc["sha256"]), "chunk digests differ"
    assert np.array_equal(files["chunk_root"], rf["chunk_root"]), "file roots differ"
    assert np.array_equal(chunks["dup_of"], rc["dup_of"]), "dedup marking differs"
    if flags_sha:
        assert np.array_equal(files["file_sha256"], rf["file_sha256"])
    return files, chunks


@pytest.fixture(scope="module")
def eng():
    import makisu_amd
    e = makisu_amd.Engine()
    yield e
    e.close()


def test_device_is_gfx950(eng):
    info = eng.device_info()
    assert "gfx950" in info["name"]
    assert info["n_cu"] >= 200


def test_sha256_many_kats(eng):
    # FIPS 180-4 / NIST examples + the reference's empty-tar constants
    # (lib/docker/image/const_linux.go:18, const_darwin.go:18, digest.go:25)
    blobs = [b"", b"abc", b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq",
             b"a" * 1000000, bytes(10240), bytes(1024)]
    blobs += [bytes(range(256)) * 3][:1] + [b"x" * n for n in (55, 56, 57, 63, 64, 65, 119, 120, 127, 128)]
    got = eng.sha256_many(blobs)
    for blob, d in zip(blobs, got):
        assert d == hashlib.sha256(blob).digest(), len(blob)
    assert got[4].hex() == "84ff92691f909a05b224e1c56abb4864f01b4f8e3c854e4bb4c7baf1d3f6d652"
    assert got[5].hex() == "5f70bf18a086007016e948b04aed3b82103a36bea41755b6cddfaf10ace3c6ef"
    assert got[0].hex() == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"


def test_sha256_many_random_lengths(eng):
    rng = np.random.default_rng(7)
    blobs = [rng.integers(0, 256, int(n), dtype=np.uint8).tobytes()
             for n in rng.integers(0, 5000, 700)]
    for blob, d in zip(blobs, eng.sha256_many(blobs)):
        assert d == hashlib.sha256(blob).digest()


def test_small_batch_random_files(oracle, eng):
    rng = np.random.default_rng(1)
    blobs = [oracle.synth_fill(SEED, i, 0, int(n)).tobytes()
             for i, n in enumerate([65536, 65536, 100000, 1, 63, 64, 65, 2047, 2048, 2049,
                                    4096, 131072, 65535, 65537, 300000])]
    _check_batch(oracle, eng, blobs)


def test_empty_and_ragged(oracle, eng):
    blobs = [b"", b"a", b"", bytes(5000), b"\xff" * 70000, b""]
    files, chunks = _check_batch(oracle, eng, blobs)
    assert files["n_chunks"][0] == 0 and files["n_chunks"][2] == 0
    # SHA-256 of zero chunk digests = SHA-256("")
    assert files["chunk_root"][0].tobytes() == hashlib.sha256(b"").digest()


def test_empty_batch(eng):
    with eng.batch() as b:
        b.run()
        assert b.counts() == (0, 0, 0)
        assert len(b.files()) == 0 and len(b.chunks()) == 0


def test_low_entropy_forced_cuts(oracle, eng):
    # all-zero, period-4 KiB and short-period content exercise max_size forced cuts
    period = oracle.synth_fill(SEED, 99, 0, 4096).tobytes()
    blobs = [bytes(500000), period * 100, b"ab" * 150000, b"\x01" * 65536]
    files, chunks = _check_batch(oracle, eng, blobs)
    assert chunks["length"].max() <= eng.cfg.max_size


def test_duplicates_are_marked(oracle, eng):
    a = oracle.synth_fill(SEED, 1, 0, 200000).tobytes()
    b_ = oracle.synth_fill(SEED, 2, 0, 70000).tobytes()
    files, chunks = _check_batch(oracle, eng, [a, b_, a, a[:150000], b_])
    assert (chunks["dup_of"] >= 0).sum() > 0
    first = chunks[chunks["dup_of"] < 0]
    assert len({x.tobytes() for x in first["sha256"]}) == len(first)


def test_multi_tile_file(oracle, eng):
    # > 64 KiB tiles: the cut carry across tiles and the 64-byte halo
    blobs = [oracle.synth_fill(SEED, 5, 0, 5 * 65536 + 1234).tobytes(),
             oracle.synth_fill(SEED, 6, 0, 3 * 65536).tobytes()]
    _check_batch(oracle, eng, blobs)


@pytest.mark.parametrize("mask_bits,min_size,max_size", [
    (0, 64, 64), (0, 64, 4096), (2, 64, 256), (6, 256, 1024), (10, 1024, 8192),
    (13, 2048, 65536), (16, 4096, 262144), (32, 2048, 65536)])
def test_param_sweep(oracle, mask_bits, min_size, max_size):
    import makisu_amd
    with makisu_amd.Engine(mask_bits=mask_bits, min_size=min_size, max_size=max_size) as e:
        blobs = [oracle.synth_fill(SEED, 40 + i, 0, n).tobytes()
                 for i, n in enumerate([70000, 5000, 200000, 64, 1])]
        blobs.append(bytes(30000))
        _check_batch(oracle, e, blobs)


def test_file_sha256_flag(oracle):
    import makisu_amd
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_SHA256) as e:
        blobs = [oracle.synth_fill(SEED, 70 + i, 0, n).tobytes() for i, n in enumerate([1, 70000, 0, 4097])]
        files, _ = _check_batch(oracle, e, blobs, flags_sha=True)
        for blob, row in zip(blobs, files):
            assert row["file_sha256"].tobytes() == hashlib.sha256(blob).digest()


def test_synthetic_matches_oracle_generator(oracle, eng):
    sizes = [65536, 1000, 65536, 7, 200000]
    cids = [3, 4, 3, 9, 11]
    with eng.batch() as b:
        b.add_synthetic(sizes, cids, seed=SEED)
        b.run()
        back = b.read_back().copy()
        files, chunks = b.files().copy(), b.chunks().copy()
    want = np.concatenate([oracle.synth_fill(SEED, c, 0, n) for c, n in zip(cids, sizes)])
    assert np.array_equal(back, want)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    rf, rc = oracle.scan_batch(want, offs, sizes, _params(oracle, eng))
    assert np.array_equal(chunks["sha256"], rc["sha256"])
    assert np.array_equal(chunks["dup_of"], rc["dup_of"])
    assert np.array_equal(files["chunk_root"], rf["chunk_root"])
    # file 2 is a copy of file 0: all of its chunks are duplicates
    f2 = chunks[chunks["file_index"] == 2]
    assert (f2["dup_of"] >= 0).all()


def test_many_small_files_vs_oracle(oracle, eng):
    # 2000 x 64 KiB of the C2 generator -- oracle runs in ~1 s
    n = 2000
    with eng.batch() as b:
        b.add_synthetic([65536] * n, None, seed=SEED)
        b.run()
        files, chunks = b.files().copy(), b.chunks().copy()
    data = np.concatenate([oracle.synth_fill(SEED, i, 0, 65536) for i in range(n)])
    rf, rc = oracle.scan_batch(data, np.arange(n) * 65536, [65536] * n, _params(oracle, eng))
    assert np.array_equal(chunks["offset"], rc["offset"])
    assert np.array_equal(chunks["length"], rc["length"])
    assert np.array_equal(chunks["sha256"], rc["sha256"])
    assert np.array_equal(files["chunk_root"], rf["chunk_root"])


def test_rerun_is_idempotent(eng):
    with eng.batch() as b:
        b.add_synthetic([65536] * 500, None, seed=SEED + 3)
        b.run()
        c1 = b.chunks().copy()
        b.rerun()
        c2 = b.chunks().copy()
    assert np.array_equal(c1, c2)


def test_add_path(oracle, eng, tmp_path):
    blobs = [oracle.synth_fill(SEED, 80 + i, 0, n).tobytes() for i, n in enumerate([100000, 0, 5])]
    with eng.batch() as b:
        for i, blob in enumerate(blobs):
            p = tmp_path / ("f%d" % i)
            p.write_bytes(blob)
            b.add_path(str(p), tag=i)
        b.run()
        chunks = b.chunks().copy()
    data = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    sizes = [len(x) for x in blobs]
    rf, rc = oracle.scan_batch(data, np.concatenate([[0], np.cumsum(sizes)[:-1]]), sizes, _params(oracle, eng))
    assert np.array_equal(chunks["sha256"], rc["sha256"])


def test_add_path_errors(eng, tmp_path):
    import makisu_amd
    with eng.batch() as b:
        with pytest.raises(makisu_amd.MiError) as ei:
            b.add_path(str(tmp_path / "missing"), size=10)
        assert ei.value.code == -5
        p = tmp_path / "short"
        p.write_bytes(b"abc")
        with pytest.raises(makisu_amd.MiError):
            b.add_path(str(p), size=10)      # shorter than the stat-time size
        b.add_path(str(p), size=2)           # CopyN semantics: only `size` bytes are read
        b.run()
        assert b.counts()[2] == 2


def test_reference_fixtures_on_gpu(eng):
    """The reference's own SHA-256 known answers (tests/golden, see make_golden.py for the
    file:line each is pinned by) hashed by the HIP kernel: whole blobs through mi_sha256_many,
    and again as whole-file digests of a batch (MI_FLAG_FILE_SHA256)."""
    import base64
    import json
    import os
    import makisu_amd
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                       "sha256_reference_fixtures.json")))["vectors"]
    blobs = [base64.b64decode(v["file_b64"]) if "file_b64" in v else bytes(v["zeros"]) for v in gold]
    for v, d in zip(gold, eng.sha256_many(blobs)):
        assert d.hex() == v["sha256"], v["name"]
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_SHA256) as e, e.batch() as b:
        for blob in blobs:
            b.add_bytes(blob)
        b.run()
        for v, row in zip(gold, b.files()):
            assert row["file_sha256"].tobytes().hex() == v["sha256"], v["name"]
            assert makisu_amd.Digest.from_raw(row["file_sha256"]) == "sha256:" + v["sha256"]


def test_long_strings_take_the_host_route_and_agree_with_the_gpu_route(tmp_path):
    """VERDICT r5 item 3 (`makisu push`: bin/makisu/cmd/push.go:207,230 -> lib/docker/image/digester.go:45-60).  The reference's
    two large fixtures -- the alpine layer blob (393ccd5c..., 675 797 B) and its Go-written layer.tar (4ac76077..., 1 308 672 B) --
    hold a launch for 50 / 97 ms on a GPU lane: alone they go to SHA-NI streams (the default), with MI_SHA_LONG_ON_GPU=1 to lanes;
    the digests are the reference's on both routes, through mi_sha256_many and through MI_FLAG_FILE_SHA256.  And the time: eight
    128 MiB blobs, 10 s on eight lanes, finish within 1.2 x 0.06 s x ceil(8 / threads) + the PCIe-free host copy"""
    import os
    import subprocess
    import sys
    code = r"""
import base64, hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, %r)
import makisu_amd as M
gold = json.load(open(os.path.join(%r, "golden", "sha256_reference_fixtures.json")))["vectors"]
import gzip
alpine = [v for v in gold if v["name"] == "alpine_layer_blob"][0]
blob = base64.b64decode(alpine["file_b64"])
layer_tar = gzip.decompress(blob)                                                      # testdata/files/busybox/393c.../layer.tar
big = [alpine, {"name": "go_written_layer_tar", "sha256": "4ac76077f2c741c856a2419dfdb0804b18e48d2e1a9ce9c6a3f0605a2078caba"}]
blobs = [blob, layer_tar]
assert (len(blob), len(layer_tar)) == (675797, 1308672)
with M.Engine(flags=M.FLAG_FILE_SHA256) as e:
    for v, d in zip(big, e.sha256_many(blobs)):
        assert d.hex() == v["sha256"], v["name"]
    small = [os.urandom(n) for n in (0, 1, 55, 64, 4096, 65536)] * 50                  # beside many short ones: both sides of one call
    got = e.sha256_many(blobs + small)
    assert [d.hex() for d in got[:2]] == [v["sha256"] for v in big]
    assert all(d == hashlib.sha256(s).digest() for d, s in zip(got[2:], small))
    with e.batch() as b:
        for blob in blobs + small:
            b.add_bytes(blob)
        b.run()
        rows = b.files()
        assert [rows["file_sha256"][i].tobytes().hex() for i in range(2)] == [v["sha256"] for v in big]
        assert all(rows["file_sha256"][2 + i].tobytes() == hashlib.sha256(s).digest() for i, s in enumerate(small))
    if os.environ.get("MI_SHA_LONG_ON_GPU") != "1":
        rng = np.random.default_rng(1)
        eight = [rng.integers(0, 256, 128 << 20, dtype=np.uint8).tobytes() for _ in range(8)]
        want = [hashlib.sha256(x).digest() for x in eight]
        data = np.frombuffer(b"".join(eight), dtype=np.uint8)
        import ctypes as C
        lens = np.full(8, 128 << 20, dtype=np.uint64); offs = (np.arange(8, dtype=np.uint64) * (128 << 20)); out = np.zeros((8, 32), dtype=np.uint8)
        u64p = C.POINTER(C.c_uint64)
        best, one_here = 1e9, 1e9
        for _ in range(3):                                                             # one blob = one SHA-NI stream on THIS box, now
            t0 = time.perf_counter()
            rc = e._lib.mi_sha256_many(e._h, data.ctypes.data, offs.ctypes.data_as(u64p), lens.ctypes.data_as(u64p), 1, out.ctypes.data)
            one_here = min(one_here, time.perf_counter() - t0)
            assert rc == 0 and out[0].tobytes() == want[0]
        for _ in range(5):
            t0 = time.perf_counter()
            rc = e._lib.mi_sha256_many(e._h, data.ctypes.data, offs.ctypes.data_as(u64p), lens.ctypes.data_as(u64p), 8, out.ctypes.data)
            best = min(best, time.perf_counter() - t0)
            assert rc == 0 and [out[i].tobytes() for i in range(8)] == want
        threads = min(len(os.sched_getaffinity(0)), 16)
        one = max(0.06, one_here)                                                      # 128 MiB on one SHA-NI core: 0.06 s, or what this box's core does today
        limit = 1.2 * one * -(-8 // threads) if threads >= 8 else 1.2 * one * 8 / threads + 0.05
        print("eight 128 MiB blobs: %%.3f s on %%d threads (one blob %%.3f s; limit %%.3f)" %% (best, threads, one_here, limit))
        assert best <= limit, (best, limit)
print("OK long_strings")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    for env in ({}, {"MI_SHA_LONG_ON_GPU": "1"}):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert p.returncode == 0 and "OK long_strings" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
        print("\n".join(ln for ln in p.stdout.splitlines() if ln.startswith("eight ")))        # pytest -s: the measured times


def test_c1_build_context_on_gpu(oracle):
    """BASELINE.json configs[0] input (testdata/build-context) through the GPU engine: per-file
    SHA-256 equals the hashlib answers recorded in the fixture, order is preserved."""
    import base64
    import json
    import os
    import makisu_amd
    ctx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "build_context_c1.json")))["entries"]
    files = [e for e in ctx if "b64" in e]
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_SHA256) as e, e.batch() as b:
        for i, ent in enumerate(files):
            b.add_bytes(base64.b64decode(ent["b64"]), tag=i)
        b.run()
        rows = b.files().copy()
    assert [int(t) for t in rows["user_tag"]] == list(range(len(files)))
    for ent, row in zip(files, rows):
        assert row["file_sha256"].tobytes().hex() == ent["sha256"], ent["path"]
        assert int(row["size"]) == ent["size"]


def test_full_c2_properties(eng):
    """BASELINE.json configs[1] at full size (100k x 64 KiB = 6.25 GiB): size-independent
    properties the oracle cannot check in seconds -- chunks tile every file exactly, sizes obey
    min/max, a second identical half dedups completely, roots depend only on content."""
    n = 100000
    cids = np.arange(n, dtype=np.uint64) % np.uint64(n // 2)       # second half repeats the first
    with eng.batch(n, n * 65536) as b:
        b.add_synthetic(np.full(n, 65536, dtype=np.uint64), cids, seed=SEED)
        b.run()
        files, chunks = b.files().copy(), b.chunks().copy()
    assert int(files["n_chunks"].sum()) == len(chunks)
    starts = files["first_chunk"].astype(np.int64)
    assert np.array_equal(starts, np.concatenate([[0], np.cumsum(files["n_chunks"])[:-1]]))
    ends = chunks["offset"] + chunks["length"]
    # within a file chunks are contiguous and end at the file size
    same_file = chunks["file_index"][1:] == chunks["file_index"][:-1]
    assert np.array_equal(chunks["offset"][1:][same_file], ends[:-1][same_file])
    assert (chunks["offset"][starts] == 0).all()
    last = starts + files["n_chunks"].astype(np.int64) - 1
    assert (ends[last] == 65536).all()
    assert chunks["length"].max() <= 65536
    not_last = np.ones(len(chunks), dtype=bool)
    not_last[last] = False
    assert chunks["length"][not_last].min() >= 2048
    # dedup: every chunk of the second half points into the first half, which is all-unique
    first_half = chunks["file_index"] < n // 2
    assert (chunks["dup_of"][first_half] == -1).all()
    assert (chunks["dup_of"][~first_half] >= 0).all()
    assert np.array_equal(chunks["sha256"][chunks["dup_of"][~first_half]], chunks["sha256"][~first_half])
    assert np.array_equal(files["chunk_root"][: n // 2], files["chunk_root"][n // 2:])
    # spot-check 64 random chunks against hashlib on bytes regenerated independently
    import hashlib
    from oracle import mi_oracle as O
    rng = np.random.default_rng(0)
    for i in rng.integers(0, len(chunks), 64):
        row = chunks[i]
        blob = O.synth_fill(SEED, int(cids[row["file_index"]]), int(row["offset"]), int(row["length"]))
        assert row["sha256"].tobytes() == hashlib.sha256(blob.tobytes()).digest()


def test_global_dedup_single_rank_nccl(eng, oracle):
    """The exchange path of makisu_amd.distributed on a real GPU (world_size 1 over RCCL):
    zero-copy digest view, all-gather, mi_dedup_mark on an external device array, rewrite."""
    import socket
    import torch
    import torch.distributed as dist
    from makisu_amd import distributed as mdist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        with eng.batch() as b:
            b.add_synthetic([65536] * 300, [i % 200 for i in range(300)], seed=SEED)
            b.run()
            local_dup = b.chunks()["dup_of"].copy()
            view = mdist.digests_tensor(b, torch.device("cuda", 0))
            assert np.array_equal(view.cpu().numpy(), b.chunks()["sha256"])
            n_total, n_unique, first, dup = mdist.global_dedup(eng, b, torch.device("cuda", 0))
            assert first == 0 and n_total == len(local_dup)
            assert np.array_equal(b.chunks()["dup_of"], local_dup)      # one rank: global == local
            assert n_unique == (local_dup < 0).sum()
            want, _ = oracle.dedup(b.chunks()["sha256"])
            assert np.array_equal(dup.cpu().numpy()[:n_total], want)
    finally:
        dist.destroy_process_group()


def _oracle_threads():
    import os
    return max(1, min(64, (os.cpu_count() or 1)))


def _compare_synth(oracle, eng, sizes, cids, seed=SEED):
    with eng.batch() as b:
        b.add_synthetic(sizes, cids, seed=seed)
        b.run()
        files, chunks = b.files().copy(), b.chunks().copy()
    data = np.concatenate([oracle.synth_fill(seed, c, 0, n) for c, n in zip(cids, sizes)])
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    rf, rc = oracle.scan_batch(data, offs, sizes, _params(oracle, eng), True, _oracle_threads(), 0)
    assert len(chunks) == len(rc)
    assert np.array_equal(chunks["file_index"], rc["file_index"])
    assert np.array_equal(chunks["offset"], rc["offset"]), "cut points differ"
    assert np.array_equal(chunks["length"], rc["length"]), "cut points differ"
    assert np.array_equal(chunks["sha256"], rc["sha256"]), "chunk digests differ"
    assert np.array_equal(files["chunk_root"], rf["chunk_root"])
    assert np.array_equal(chunks["dup_of"], rc["dup_of"])
    return files, chunks


def test_large_files_workgroup_path(oracle, eng):
    # files > 64 KiB take the one-workgroup-per-file kernel (4 tiles per step, cut carry
    # across tiles and steps); sizes straddle tile and step boundaries
    sizes = [65537, 4 * 65536, 4 * 65536 + 1, 5 * 1024 * 1024 + 77, 20 * 1024 * 1024, 65536, 3]
    _compare_synth(oracle, eng, sizes, list(range(100, 100 + len(sizes))))


def test_c3_scaled_128mib_files(oracle, eng):
    # BASELINE.json configs[2] shape (128 MiB files) at 6 files: ~0.75 GiB, oracle needs seconds
    sizes = [128 * 1024 * 1024] * 6
    files, chunks = _compare_synth(oracle, eng, sizes, [500, 501, 502, 500, 503, 501], seed=SEED + 1)
    assert (chunks["dup_of"][chunks["file_index"] == 3] >= 0).all()      # file 3 repeats file 0


def test_c5_zipf_mix_with_duplicates(oracle, eng):
    # BASELINE.json configs[4] shape, scaled: sizes 1 KiB..32 MiB on a log scale, 90 % of the
    # files are copies of the other 10 %; the unique chunk count must equal what the distinct
    # contents alone produce
    rng = np.random.default_rng(5)
    n = 400
    sizes = (2.0 ** rng.uniform(10, 25, n)).astype(np.int64)
    sizes[0] = 1 << 25
    distinct = n // 10
    cids = np.arange(n)
    src = rng.integers(0, distinct, n - distinct)
    cids[distinct:] = src
    sizes[distinct:] = sizes[src]
    files, chunks = _compare_synth(oracle, eng, [int(s) for s in sizes], [int(c) for c in cids], seed=SEED + 2)
    n_first = int(files["n_chunks"][:distinct].sum())
    uniq_in_distinct = len({x.tobytes() for x in chunks["sha256"][:n_first]})
    assert (chunks["dup_of"] < 0).sum() == uniq_in_distinct
    assert (chunks["dup_of"][n_first:] >= 0).all()


def test_host_staging_bigger_than_the_ring(oracle, eng):
    # host-fed bytes larger than both 64 MiB pinned staging buffers: flush/reuse of the ring
    blob = oracle.synth_fill(SEED, 900, 0, 150 * 1024 * 1024 + 13)
    small = [oracle.synth_fill(SEED, 901 + i, 0, n).tobytes() for i, n in enumerate([10, 70000, 0])]
    with eng.batch() as b:
        b.add_bytes(small[0], tag=1)
        b.add_bytes(blob, tag=2)
        b.add_bytes(small[1], tag=3)
        b.add_bytes(small[2], tag=4)
        b.run()
        back = b.read_back().copy()
        files, chunks = b.files().copy(), b.chunks().copy()
    want = np.concatenate([np.frombuffer(small[0], np.uint8), blob, np.frombuffer(small[1], np.uint8)])
    assert np.array_equal(back, want)
    sizes = [len(small[0]), blob.size, len(small[1]), 0]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    rf, rc = oracle.scan_batch(want, offs, sizes, _params(oracle, eng), True, _oracle_threads(), 0)
    assert np.array_equal(chunks["offset"], rc["offset"]) and np.array_equal(chunks["sha256"], rc["sha256"])
    assert np.array_equal(files["chunk_root"], rf["chunk_root"])
    assert [int(t) for t in files["user_tag"]] == [1, 2, 3, 4]


def test_two_batches_in_flight(oracle, eng):
    # mi_batch_submit / mi_batch_wait with two batches overlapping on one ctx
    a, b = eng.batch(), eng.batch()
    try:
        a.add_synthetic([65536] * 3000, list(range(3000)), seed=SEED)
        b.add_synthetic([65536] * 3000, list(range(3000, 6000)), seed=SEED)
        a.submit(); b.submit(); a.wait(); b.wait()
        ca1, cb1 = a.chunks().copy(), b.chunks().copy()
        for _ in range(3):
            a.submit(); b.submit(); a.wait(); b.wait()
        assert np.array_equal(a.chunks(), ca1) and np.array_equal(b.chunks(), cb1)
        assert not np.array_equal(ca1["sha256"][:100], cb1["sha256"][:100])
        data = np.concatenate([oracle.synth_fill(SEED, i, 0, 65536) for i in range(3000, 3040)])
        rf, rc = oracle.scan_batch(data, np.arange(40) * 65536, [65536] * 40, _params(oracle, eng), True, 1, 4)
        assert np.array_equal(cb1["sha256"][:len(rc)], rc["sha256"])
    finally:
        a.free(); b.free()


def test_file_crc32_matches_zlib(oracle):
    """MI_FLAG_FILE_CRC32: per-file CRC32-IEEE from the GPU (lane runs + GF(2) folding) against
    zlib, which is the same polynomial as Go's hash/crc32 IEEE the reference uses
    (lib/builder/step/add_copy_step.go:104)."""
    import zlib
    import makisu_amd
    sizes = [0, 1, 3, 4, 5, 127, 128, 129, 1023, 1024, 1025, 4096, 65535, 65536, 65537,
             3 * 65536, 3 * 65536 + 1000, 1 << 20, 5 * (1 << 20) + 3]
    blobs = [oracle.synth_fill(SEED, 300 + i, 0, n).tobytes() for i, n in enumerate(sizes)]
    blobs += [bytes(70000), b"\xff" * 1500, b"123456789"]
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_CRC32) as e, e.batch() as b:
        for blob in blobs:
            b.add_bytes(blob)
        b.run()
        rows = b.files().copy()
    for blob, row in zip(blobs, rows):
        assert int(row["crc32"]) == zlib.crc32(blob), len(blob)
    assert int(rows["crc32"][-1]) == 0xCBF43926            # the classic CRC-32 check value


def test_file_crc32_large_files_fold_in_parallel(oracle):
    """The per-file combine is a parallel fold over the file's 64 KiB tiles (one lane per tile, terms XORed per
    file): files whose tiles fill whole waves, straddle waves, and sit between one-tile files, against zlib."""
    import zlib
    import makisu_amd
    sizes = [5, 300 * (1 << 20) + 12345, 70000, 64 * 65536, 9, 65 * 65536, 65536, 127 * 65536 + 1, 0, 3, 1 << 27]
    cids = list(range(4100, 4100 + len(sizes)))
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_CRC32) as e, e.batch() as b:
        b.add_synthetic(sizes, cids, seed=SEED)
        b.run()
        rows = b.files().copy()
    for n, c, row in zip(sizes, cids, rows):
        assert int(row["crc32"]) == zlib.crc32(oracle.synth_fill(SEED, c, 0, n).tobytes()), n


def _walk_order(entries):
    """filepath.Walk order over a set of relative paths: lexical per directory, a directory
    before its children (Go path/filepath: Walk sorts names in each directory)."""
    tree = {}
    for e in entries:
        node = tree
        parts = e["path"].split("/")
        for d in parts[:-1]:
            node = node.setdefault(d, {})
        node[parts[-1]] = e
    out = []

    def rec(node, prefix):
        for name in sorted(node):
            v = node[name]
            rel = prefix + name
            if isinstance(v, dict) and "path" not in v:
                out.append({"path": rel, "dir": True})
                rec(v, rel + "/")
            else:
                out.append(v)
    rec(tree, "")
    return out


def test_context_checksum_c1_build_context():
    """The reference's COPY/ADD cache ID over testdata/build-context (BASELINE.json configs[0]):
    one running CRC32 over seed+directive+args, then per walked path relpath + bytes / link
    target (add_copy_step.go:102-238).  Expected value: zlib over the same byte stream, built
    independently here; the engine must give the same unpadded-hex cacheID."""
    import base64
    import json
    import os
    import zlib
    import makisu_amd
    ctx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "build_context_c1.json")))["entries"]
    # the fixture tree has no symlinks; add two so that branch (:221-227) is exercised too
    ctx = ctx + [{"path": "simple/zz-link", "symlink": "Dockerfile"}, {"path": "a-link", "symlink": "/etc/hosts"}]
    walk = _walk_order(ctx)
    prefix = b"deadbeef" + b"COPY" + b". /app/"
    running = zlib.crc32(prefix)
    entries, blobs = [], []
    for e in walk:
        running = zlib.crc32(e["path"].encode(), running)
        if e.get("dir"):
            entries.append((e["path"], None, -1))
        elif "symlink" in e:
            running = zlib.crc32(e["symlink"].encode(), running)
            entries.append((e["path"], e["symlink"], -1))
        else:
            data = base64.b64decode(e["b64"])
            running = zlib.crc32(data, running)
            entries.append((e["path"], None, len(blobs)))
            blobs.append(data)
    assert len(blobs) >= 20 and any(x[1] for x in entries) and any(x[2] == -1 and not x[1] for x in entries)
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_CRC32) as eng, eng.batch() as b:
        for blob in blobs:
            b.add_bytes(blob)
        b.run()
        assert b.context_checksum(prefix, entries) == "%x" % running
        # cache-ID relations the reference's own test asserts (copy_step_test.go:51-169):
        # same inputs -> same ID, different args -> different ID
        assert b.context_checksum(prefix, entries) == b.context_checksum(prefix, entries)
        assert b.context_checksum(b"deadbeefCOPY. /other/", entries) != "%x" % running
    # changing one byte of one file changes the ID
    blobs2 = list(blobs)
    blobs2[3] = blobs2[3][:-1] + bytes([blobs2[3][-1] ^ 1]) if blobs2[3] else b"x"
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_CRC32) as eng, eng.batch() as b:
        for blob in blobs2:
            b.add_bytes(blob)
        b.run()
        assert b.context_checksum(prefix, entries) != "%x" % running


def test_context_checksum_needs_the_flag(eng):
    import makisu_amd
    with eng.batch() as b:
        b.add_bytes(b"abc")
        b.run()
        with pytest.raises(makisu_amd.MiError) as ei:
            b.context_checksum(b"", [("a", None, 0)])
        assert ei.value.code == -6


def _go_walk(root):
    """path/filepath.Walk: lstat, visit, then children in bytewise name order; no symlink following."""
    import os
    import stat
    st = os.lstat(root)
    yield root, st
    if stat.S_ISDIR(st.st_mode):
        for name in sorted(os.listdir(root), key=os.fsencode):
            yield from _go_walk(os.path.join(root, name))


def _make_tree(base):
    import os
    os.makedirs(base / "a" / "deep" / "er")
    os.makedirs(base / "a-b")
    os.makedirs(base / "empty-dir")
    os.makedirs(base / "z" / ".wh..wh.plnk")
    (base / "a" / "x.txt").write_bytes(b"hello world\n" * 1000)
    (base / "a" / "deep" / "er" / "big.bin").write_bytes(bytes(range(256)) * 1500)      # 384000 B, multi tile
    (base / "a-b" / "y").write_bytes(b"")
    (base / "a.txt").write_bytes(b"sorted between a/ and a-b? Walk order decides\n")
    (base / "z" / "last").write_bytes(b"\x00" * 70000)
    (base / "z" / ".wh..wh.plnk" / "hidden").write_bytes(b"aufs metadata")
    (base / "z" / ".wh.deleted").write_bytes(b"")                                       # a plain whiteout marker
    os.symlink("x.txt", base / "a" / "link-to-x")
    os.symlink("/nonexistent/target", base / "dangling")
    os.mkfifo(base / "a" / "fifo")
    return base


def test_tree_walk_context_checksum(tmp_path):
    """mi_batch_add_tree(MI_TREE_CONTEXT) + mi_context_checksum_tree against an independent
    emulation of filepath.Walk + checksumPathContents (add_copy_step.go:153-238) with zlib."""
    import os
    import stat
    import zlib
    import makisu_amd
    root = str(_make_tree(tmp_path / "ctx"))
    prefix = b"seedADD. /dst/"
    running = zlib.crc32(prefix)
    want_rel = []
    for path, st in _go_walk(root):
        if stat.S_ISFIFO(st.st_mode) or stat.S_ISSOCK(st.st_mode) or stat.S_ISCHR(st.st_mode) or stat.S_ISBLK(st.st_mode):
            continue                                             # utils.IsSpecialFile -> skipped
        rel = os.path.relpath(path, root)
        want_rel.append(rel)
        running = zlib.crc32(rel.encode(), running)
        if stat.S_ISLNK(st.st_mode):
            running = zlib.crc32(os.readlink(path).encode(), running)
        elif stat.S_ISREG(st.st_mode):
            running = zlib.crc32(open(path, "rb").read(), running)
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_CRC32) as eng, eng.batch() as b:
        n = b.add_tree(root)
        ents = b.tree_entries(n)
        assert [e[0] for e in ents] == want_rel                 # Walk order, "." first, fifo skipped
        assert want_rel[0] == "." and "a/fifo" not in want_rel
        assert want_rel.index("a-b") > want_rel.index("a/x.txt")          # Walk order != sorted full paths
        assert dict((e[0], e[1]) for e in ents)["dangling"] == "/nonexistent/target"
        b.run()
        assert b.context_checksum_tree(prefix) == "%x" % running
        files = b.files()
        regular = [e for e in ents if e[4] == 1]
        assert [int(t) for t in files["user_tag"]] == [ents.index(e) for e in regular]
        assert [int(s) for s in files["size"]] == [e[3] for e in regular]


def test_tree_walk_scan_mode_skips(tmp_path):
    """MI_TREE_SCAN follows shouldSkip (lib/snapshot/utils.go:37-52): whiteout-META names and
    blacklisted subtrees are pruned, plain ".wh." whiteout markers are kept, special files skipped;
    relpaths are relative to rel_base."""
    import os
    import makisu_amd
    base = _make_tree(tmp_path / "root")
    os.unlink(base / "dangling")       # an absolute target outside rel_base fails the scan (createHeader's TrimRoot)
    with makisu_amd.Engine() as eng, eng.batch() as b:
        n = b.add_tree(str(base), rel_base=str(tmp_path), blacklist=[str(base / "a" / "deep")],
                       mode=makisu_amd.TREE_SCAN)
        rels = [e[0] for e in b.tree_entries(n)]
        b.run()
        assert b.counts()[0] == sum(1 for e in b.tree_entries(n) if e[4] == 1)
    assert rels[0] == "root"
    assert "root/z/.wh.deleted" in rels and "root/z/last" in rels
    assert not any(".wh..wh." in r for r in rels)
    assert not any(r.startswith("root/a/deep") for r in rels)      # blacklist prunes the subtree
    assert "root/a/fifo" not in rels and "root/a/link-to-x" in rels
    assert rels == sorted(rels, key=lambda r: [p.encode() for p in r.split("/")])   # lexical per level


def _two_rank_worker(rank, world, port, q):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import makisu_amd
    from makisu_amd import distributed as mdist
    torch.cuda.set_device(0)                       # the test box has one GPU: both ranks share it
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    n_files = 600
    mine = mdist.shard_round_robin(n_files, rank, world)            # C4: file_index mod world
    cids = [int(i) % 251 for i in mine]                             # content repeats across ranks
    sizes = [30000 + 7000 * (c % 9) for c in cids]
    with makisu_amd.Engine(device=0, flags=makisu_amd.FLAG_NO_DEDUP) as eng, eng.batch() as b:
        b.add_synthetic(sizes, cids, seed=SEED)
        b.run()
        n_total, n_unique, first, dup = mdist.global_dedup(eng, b, dev)
        chunks = b.chunks().copy()
    q.put((rank, int(n_total), int(n_unique), int(first), chunks["sha256"].tobytes(),
           chunks["dup_of"].tolist(), [int(c) for c in cids], sizes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_global_dedup_on_gpu(oracle):
    """N=2 through the real engine: two processes (sharing the one GPU of the test box, gloo as
    the transport) scan their round-robin shards, all-gather the digest sets, mark duplicates
    globally with the HIP kernel and rewrite dup_of with global rank-major indices; the result
    must equal the oracle's marking of the concatenated digest set."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (r0, nt0, nu0, f0, d0, dup0, cids0, sz0), (r1, nt1, nu1, f1, d1, dup1, cids1, sz1) = res
    assert nt0 == nt1 == len(dup0) + len(dup1) and nu0 == nu1
    assert f0 == 0 and f1 == len(dup0)
    all_digests = np.frombuffer(d0 + d1, dtype=np.uint8).reshape(-1, 32)
    want, uniq = oracle.dedup(all_digests)
    assert np.array_equal(np.array(dup0 + dup1), want) and uniq == nu0
    assert (np.array(dup1) >= 0).sum() > 0 and (np.array(dup1)[np.array(dup1) >= 0] < f1).any()  # cross-rank hits
    # and the digests themselves are what the oracle computes for rank 1's shard
    data = np.concatenate([oracle.synth_fill(SEED, c, 0, n) for c, n in zip(cids1, sz1)])
    offs = np.concatenate([[0], np.cumsum(sz1)[:-1]])
    rf, rc = oracle.scan_batch(data, offs, sz1, oracle.CdcParams(SEED, 13, 2048, 65536), True, 4, 4)
    assert rc["sha256"].tobytes() == d1


def test_native_rccl_exchange_single_rank(oracle):
    """mi_comm_* + mi_dedup_allgather: the digest exchange done by the library itself over RCCL
    (what a Go host would call).  One rank here; the multi-rank logic (ragged counts, rank-major
    global indices) is the same code and is exercised by the gloo / two-rank tests above."""
    import makisu_amd
    with makisu_amd.Engine(flags=makisu_amd.FLAG_NO_DEDUP) as eng:
        eng.comm_init_rank(1, 0, makisu_amd.Engine.comm_unique_id())
        with eng.batch() as b:
            b.add_synthetic([65536] * 400, [i % 300 for i in range(400)], seed=SEED)
            b.run()
            assert (b.chunks()["dup_of"] == -1).all()            # local marking was off
            n_total, n_unique, first = b.dedup_allgather()
            chunks = b.chunks().copy()
            # the hash-partitioned form on the REAL library (one rank: the share is a device copy, the counts and the
            # summed unique count still go through ncclAllGather): same column, same scalars
            assert b.dedup_alltoall() == (n_total, n_unique, first)
            assert np.array_equal(b.chunks()["dup_of"], chunks["dup_of"])
        want, uniq = oracle.dedup(chunks["sha256"])
        assert n_total == len(chunks) and first == 0 and n_unique == uniq
        assert np.array_equal(chunks["dup_of"], want)
        with pytest.raises(makisu_amd.MiError):
            eng.comm_init_rank(1, 0, makisu_amd.Engine.comm_unique_id())     # already initialised
        eng.comm_destroy()


def test_native_rccl_init_all_single_device(oracle):
    """Single-process form (mi_comm_init_all + mi_dedup_allgather_all) with the one device here."""
    import ctypes as C
    import makisu_amd
    lib = makisu_amd.load_library()
    with makisu_amd.Engine() as eng, eng.batch() as b:
        ctxs = (C.c_void_p * 1)(eng._h)
        assert lib.mi_comm_init_all(ctxs, 1) == 0, lib.mi_last_error(eng._h)
        b.add_synthetic([30000] * 50, [i % 20 for i in range(50)], seed=SEED)
        b.run()
        local = b.chunks()["dup_of"].copy()
        batches = (C.c_void_p * 1)(b._h)
        nt, nu = C.c_uint64(), C.c_uint64()
        assert lib.mi_dedup_allgather_all(batches, 1, C.byref(nt), C.byref(nu)) == 0, lib.mi_last_error(eng._h)
        b._lib.mi_batch_counts  # keep binding alive
        assert nt.value == len(local) and nu.value == (local < 0).sum()
        # results cache must be refreshed after the rewrite
        assert np.array_equal(b.chunks()["dup_of"], local)
        assert lib.mi_dedup_alltoall_all(batches, 1, C.byref(nt), C.byref(nu)) == 0, lib.mi_last_error(eng._h)
        assert nt.value == len(local) and nu.value == (local < 0).sum() and np.array_equal(b.chunks()["dup_of"], local)


def test_chunk_root_tree_three_levels(oracle):
    """chunk_root is a fan-out-64 tree above 64 chunks: 1.3 M tiny chunks in one file need three
    reduction passes plus the final one; a 3000-chunk (one pass) and a 5-chunk file ride along."""
    import makisu_amd
    with makisu_amd.Engine(mask_bits=6, min_size=64, max_size=256) as e:
        _compare_synth(oracle, e, [160 * 1024 * 1024, 400000, 700, 0], [40, 41, 42, 43])


@pytest.mark.skipif(False, reason="")
def test_chunk_root_definition(oracle):
    """<= 64 chunks: root = SHA-256(concat digests); above: the same over 64-wide node digests, repeated."""
    import hashlib
    rng = np.random.default_rng(3)
    for n in (0, 1, 64, 65, 4096, 4097, 300000):
        d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        nodes = [d[i].tobytes() for i in range(n)]
        while len(nodes) > 64:
            nodes = [hashlib.sha256(b"".join(nodes[i:i + 64])).digest() for i in range(0, len(nodes), 64)]
        assert oracle.chunk_root(d) == hashlib.sha256(b"".join(nodes)).digest()


def test_plain_c_consumer(oracle, tmp_path):
    """The C ABI used from plain C (tests/cabi/driver.c, gcc + -lmakisu_mi) the way the cgo shim
    would: files by path, one batch, result tables printed as text; compared with hashlib / zlib
    and the oracle's cut points."""
    import hashlib
    import os
    import subprocess
    import zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "driver")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cabi", "driver.c"), "-o", exe,
                           "-L", os.path.join(root, "makisu_amd"), "-lmakisu_mi",
                           "-Wl,-rpath," + os.path.join(root, "makisu_amd")])
    blobs = [oracle.synth_fill(SEED, 700 + i, 0, n).tobytes() for i, n in enumerate([200000, 0, 65536, 1])]
    blobs.append(blobs[0])                                  # a duplicate file
    paths = []
    for i, blob in enumerate(blobs):
        p = tmp_path / ("in%d.bin" % i)
        p.write_bytes(blob)
        paths.append(str(p))
    out = subprocess.run([exe] + paths, check=True, capture_output=True, text=True)
    frows = [l.split() for l in out.stdout.splitlines() if l.startswith("F ")]
    crows = [l.split() for l in out.stdout.splitlines() if l.startswith("C ")]
    assert len(frows) == len(blobs)
    for blob, r in zip(blobs, frows):
        assert int(r[2]) == len(blob)
        assert r[5] == "sha256:" + hashlib.sha256(blob).hexdigest()          # image.Digest format
        assert r[6] == "%x" % zlib.crc32(blob)
    data = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    sizes = [len(b) for b in blobs]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    rf, rc = oracle.scan_batch(data, offs, sizes, oracle.CdcParams(SEED, 13, 2048, 65536))
    assert [(int(r[1]), int(r[2]), int(r[3]), int(r[4]), r[5]) for r in crows] == \
        [(int(c["file_index"]), int(c["offset"]), int(c["length"]), int(c["dup_of"]),
          "sha256:" + c["sha256"].tobytes().hex()) for c in rc]
    assert [r[4] for r in frows] == ["sha256:" + f["chunk_root"].tobytes().hex() for f in rf]
    assert "unique" in out.stderr
    irows = [l.split() for l in out.stdout.splitlines() if l.startswith("I ")]
    n_distinct = len({c["sha256"].tobytes() for c in rc})
    assert irows == [["I", "0", str(n_distinct), "0", str(n_distinct)],
                     ["I", "1", "0", str(len(rc)), str(n_distinct)]]
    # files 0 and 1 (200000 bytes vs empty) have equal made-up headers but different roots
    assert [l for l in out.stdout.splitlines() if l.startswith("S ")] == ["S 1 0"]
    # error behaviour: a missing path is reported through mi_last_error, exit code 1
    bad = subprocess.run([exe, str(tmp_path / "nope")], capture_output=True, text=True)
    assert bad.returncode == 1


def test_file_beyond_4gib_offsets(oracle, eng):
    """One 5 GiB file (every 32-bit offset assumption would break): size-independent properties,
    the oracle's cuts on the first 48 MiB prefix, and re-hashed spot checks deep inside the file."""
    import hashlib
    size = 5 * (1 << 30) + 12345
    with eng.batch() as b:
        b.add_synthetic([size, 70000], [77, 78], seed=SEED + 9)
        b.run()
        files, chunks = b.files().copy(), b.chunks().copy()
    big = chunks[chunks["file_index"] == 0]
    assert int(files["n_chunks"][0]) == len(big) and int(files["size"][0]) == size
    ends = big["offset"] + big["length"]
    assert big["offset"][0] == 0 and ends[-1] == size
    assert np.array_equal(big["offset"][1:], ends[:-1])                       # exact tiling
    assert big["length"].max() <= 65536 and big["length"][:-1].min() >= 2048
    assert (big["offset"] > 2**32).sum() > 1000                               # really beyond 4 GiB
    # cut points are prefix-stable: the oracle on a 48 MiB prefix must reproduce every cut whose
    # chunk ends well before the prefix end
    n_pre = 48 << 20
    pre = oracle.synth_fill(SEED + 9, 77, 0, n_pre)
    want = oracle.cdc_two_phase(pre, oracle.CdcParams(SEED, 13, 2048, 65536))
    keep = want[want < n_pre - 65536]
    assert np.array_equal(ends[: len(keep)], keep)
    rng = np.random.default_rng(1)
    for i in list(rng.integers(0, len(big), 24)) + [len(big) - 1, int(np.searchsorted(big["offset"], 2**32))]:
        row = big[i]
        blob = oracle.synth_fill(SEED + 9, 77, int(row["offset"]), int(row["length"]))
        assert row["sha256"].tobytes() == hashlib.sha256(blob.tobytes()).digest(), int(row["offset"])
    # the root of a 600k-chunk file goes through the tree; recompute it from the chunk digests
    assert files["chunk_root"][0].tobytes() == oracle.chunk_root(big["sha256"])


def test_chunk_index_across_batches(oracle, eng):
    """mi_index_*: 'known' = the digest was held before this batch; checked against Python sets of
    the oracle-verified digests; growth from the minimum table; export -> import round trip."""
    rng = np.random.default_rng(5)
    sizes1 = rng.integers(1, 300000, 40).astype(np.uint64)
    cids1 = np.arange(40, dtype=np.uint64)
    sizes2 = np.concatenate([sizes1[:15], rng.integers(1, 300000, 25).astype(np.uint64)])
    cids2 = np.concatenate([cids1[:15], 100 + np.arange(25, dtype=np.uint64)])   # 15 files repeat
    cids2[20] = cids2[21]                                                         # an in-batch repeat
    sizes2[20] = sizes2[21]
    with eng.index() as idx:
        assert len(idx) == 0
        _, ch1 = _compare_synth(oracle, eng, sizes1, cids1)
        with eng.batch() as b:
            b.add_synthetic(sizes1, cids1, seed=SEED)
            b.run()
            known1, new1, nk1 = idx.add_batch(b)
        set1 = {r.tobytes() for r in ch1["sha256"]}
        assert new1 == len(set1) > 512                 # grew past the 1024-slot start
        assert nk1 == 0 and not known1.any() and len(idx) == len(set1)
        _, ch2 = _compare_synth(oracle, eng, sizes2, cids2)
        with eng.batch() as b:
            b.add_synthetic(sizes2, cids2, seed=SEED)
            b.run()
            known2, new2, nk2 = idx.add_batch(b)
            known_again, new_again, _ = idx.add_batch(b)
        want = np.array([r.tobytes() in set1 for r in ch2["sha256"]], dtype=np.uint8)
        assert np.array_equal(known2, want)
        assert nk2 == int(want.sum()) > 0
        set2 = {r.tobytes() for r in ch2["sha256"]}
        assert new2 == len(set2 - set1) > 0
        assert new_again == 0 and known_again.all()
        blob = idx.export()
        assert len(blob) == 32 * len(set1 | set2)
        assert {blob[i:i + 32] for i in range(0, len(blob), 32)} == set1 | set2
        with eng.index(len(set1)) as idx2:
            assert idx2.load(blob + blob[:64]) == len(set1 | set2)     # repeats inside the blob are fine
            assert idx2.load(blob[:320]) == 0
            with eng.batch() as b:
                b.add_synthetic(sizes2, cids2, seed=SEED)
                b.run()
                k, n_new, n_known = idx2.add_batch(b)
            assert k.all() and n_new == 0 and n_known == len(ch2)


def test_chunk_index_empty_batch(eng):
    with eng.index() as idx, eng.batch() as b:
        b.run()
        k, n_new, n_known = idx.add_batch(b)
        assert len(k) == 0 and n_new == 0 and n_known == 0 and idx.export() == b""


def test_dedup_mark_range_equals_full_marking(oracle, eng):
    """mi_dedup_mark_range answers only a rank's own rows but must give exactly the values the
    full marking (oracle over the whole job-wide set) has for them, for every split of the set
    into earlier / own / later rows -- including empty ranges and own rows that repeat both
    inside the range and in earlier ranks."""
    import torch
    rng = np.random.default_rng(11)
    base = rng.integers(0, 256, (4000, 32), dtype=np.uint8)
    rows = base[rng.integers(0, 4000, 20000)]                 # heavy repetition, random order
    rows[7] = rows[19999]                                      # an early row equal to the last one
    want, n_unique = oracle.dedup(rows)
    dev = torch.device("cuda", 0)
    glob = torch.from_numpy(rows).to(dev)
    firsts = 0
    bounds = [0, 1, 2500, 2500, 9000, 19999, 20000]           # 6 "ranks": sizes 1, 2499, 0, 6500, 10999, 1
    for a, b in zip(bounds[:-1], bounds[1:]):
        dup = torch.full((max(b - a, 1),), -7, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        nf = eng.dedup_mark_range(glob.data_ptr(), len(rows), a, b - a, dup.data_ptr())
        got = dup.cpu().numpy()[: b - a]
        assert np.array_equal(got, want[a:b]), (a, b)
        assert nf == int((want[a:b] < 0).sum())
        firsts += nf
    assert firsts == n_unique
    # the whole set as one range = mi_dedup_mark
    dup = torch.empty(len(rows), dtype=torch.int64, device=dev)
    assert eng.dedup_mark_range(glob.data_ptr(), len(rows), 0, len(rows), dup.data_ptr()) == n_unique
    assert np.array_equal(dup.cpu().numpy(), want)


def test_chunk_rows_view_and_prefetch(oracle):
    """mi_batch_chunks_view (rows packed on the device, one copy into the batch's pinned buffer) gives
    the rows mi_batch_chunks copies out; with MI_FLAG_PREFETCH_ROWS they arrive with mi_batch_wait;
    a global marking afterwards is visible in the next view."""
    import makisu_amd
    sizes = [0, 5, 70000, 3 * 262144 + 9, 2048, 65536] * 5
    with makisu_amd.Engine(flags=makisu_amd.FLAG_PREFETCH_ROWS) as e:
        with e.batch() as b:
            b.add_synthetic(sizes, list(range(600, 600 + len(sizes) // 2)) * 2, seed=SEED)   # second half repeats
            b.run()
            rows = b.chunks()
            view = b.chunks_view()
            assert view.dtype == rows.dtype and len(view) == len(rows) == b.counts()[1]
            assert view.tobytes() == rows.tobytes()
            blobs = [oracle.synth_fill(SEED, 600 + i % (len(sizes) // 2), 0, n) for i, n in enumerate(sizes)]
            data = np.concatenate(blobs)
            offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
            _, rc = oracle.scan_batch(data, offs, sizes, oracle.CdcParams(SEED, 13, 2048, 65536))
            for k in ("file_index", "offset", "length", "dup_of", "sha256"):
                assert np.array_equal(view[k], rc[k]), k
            assert (view["dup_of"] >= 0).sum() > 0
            b.rerun()
            assert b.chunks_view().tobytes() == rows.tobytes()


def test_sha_valu_roof_and_per_ctx_tuning(oracle):
    """mi_sha_valu_roof measures the compression-only rate of this device (what bench.py quotes the
    hashing pass against, same run); the SHA tuning fields of mi_config belong to the ctx: two ctxs of
    ONE process hash the same batch with different load schemes and workgroup counts, same digests."""
    import makisu_amd
    sizes = [65536] * 300 + [300000, 1, 0, 5 << 20]
    rows = []
    with makisu_amd.Engine() as e0:
        roof = e0.sha_valu_roof()
        assert 0.8e12 < roof < 3.0e12, roof                    # 1.57-1.78 TB/s on the pool's boxes
        assert e0.sha_valu_roof(2, 64) < 1.2 * roof             # fewer waves per SIMD: not faster
        assert e0.comm_ranks() == 0
        for kw in ({}, {"sha_load_scheme": makisu_amd.SHA_LOADS_COOP, "sha_blocks_per_cu": 1},
                   {"sha_load_scheme": makisu_amd.SHA_LOADS_LANE, "sha_blocks_per_cu": 3, "sha_coop_blocks_per_cu": 2}):
            with makisu_amd.Engine(**kw) as e, e.batch() as b:  # e0 stays alive: ctxs side by side
                b.add_synthetic(sizes, None, seed=SEED)
                b.run()
                rows.append((b.chunks()["sha256"].copy(), b.files()["chunk_root"].copy()))
    for sha, roots in rows[1:]:
        assert np.array_equal(sha, rows[0][0]) and np.array_equal(roots, rows[0][1])
    with pytest.raises(makisu_amd.MiError):
        makisu_amd.Engine(sha_load_scheme=7)
