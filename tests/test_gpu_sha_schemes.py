"""The SHA chunk pass has two load schemes (sha256.hip kCoop): lane-owned byte-aligned loads for
arenas below 9 GiB, quad-cooperative dword-aligned loads + LDS transposition above.  The parity
suite's inputs are all below the switch, so this file runs a set of cases in a child process with
the cooperative scheme forced (MI_SHA_COOP_MIN_GIB=0) and, once, forbidden: digests against
hashlib (FIPS 180-4 via OpenSSL), cut points against the oracle.  One full-size case crosses the
switch for real: a 12 GiB synthetic batch whose chunk digests are checked on a sample."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent(r'''
    import hashlib, os, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    import torch  # noqa: F401  (before the engine: one HIP runtime per process)
    import makisu_amd
    from oracle import mi_oracle as O
    SEED = 0x4D414B49
    rng = np.random.default_rng(5)
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_SHA256) as e:
        # strings at every byte alignment and every length class, incl. the padding corner cases
        blobs = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in
                 list(range(0, 200)) + [255, 256, 257, 4095, 4096, 4097, 65535, 65536, 65537, 300001]]
        got = e.sha256_many(blobs)
        assert [bytes(g).hex() for g in got] == [hashlib.sha256(b).hexdigest() for b in blobs]
        # a batch mixing tiny, small and multi-group files, odd sizes -> chunk starts at every alignment
        sizes = [1, 63, 64, 65, 1000, 70001, 262145, 3 * 262144 + 17, 5, 2048, 2049, 1 << 20, 7 * 65536 + 3] * 3
        files = [O.synth_fill(SEED, 900 + i, 0, n).tobytes() for i, n in enumerate(sizes)]
        with e.batch() as b:
            for i, f in enumerate(files):
                b.add_bytes(f, tag=i)
            b.run()
            fl, ch = b.files().copy(), b.chunks().copy()
        p = O.CdcParams(e.cfg.gear_seed, e.cfg.mask_bits, e.cfg.min_size, e.cfg.max_size)
        data = np.frombuffer(b"".join(files), dtype=np.uint8)
        sz = np.array(sizes, dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(sz)[:-1]]).astype(np.uint64)
        rf, rc = O.scan_batch(data, offs, sz, p, True, 8, 0)
        assert np.array_equal(ch["offset"], rc["offset"]) and np.array_equal(ch["length"], rc["length"])
        assert np.array_equal(ch["sha256"], rc["sha256"]) and np.array_equal(fl["chunk_root"], rf["chunk_root"])
        for i, f in enumerate(files):
            assert bytes(fl["file_sha256"][i]).hex() == hashlib.sha256(f).hexdigest()
        for r in ch[:: max(1, len(ch) // 200)]:
            f = files[int(r["file_index"])]
            assert bytes(r["sha256"]).hex() == hashlib.sha256(f[int(r["offset"]): int(r["offset"]) + int(r["length"])]).hexdigest()
    print("OK", len(ch))
''')


def _run_child(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OK" in r.stdout


def test_cooperative_scheme_forced():
    _run_child({"MI_SHA_COOP_MIN_GIB": "0"})


def test_cooperative_scheme_forced_three_workgroups_per_cu():
    _run_child({"MI_SHA_COOP_MIN_GIB": "0", "MI_SHA_COOP_BLOCKS_PER_CU": "3"})


def test_lane_owned_scheme_forced():
    _run_child({"MI_SHA_COOP_MIN_GIB": "100000"})


def test_batch_above_the_switch(oracle):
    """12 GiB arena: the cooperative scheme by the launcher's own rule; sampled chunks re-hashed on
    the host from the oracle's generator, and the batch's unique count is its chunk count."""
    import hashlib

    import numpy as np
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import makisu_amd
    n_files, size = 12 * 1024, 1 << 20
    with makisu_amd.Engine() as e:
        with e.batch() as b:
            b.add_synthetic([size] * n_files, list(range(50000, 50000 + n_files)))
            b.run()
            ch = b.chunks().copy()
            st = e.stats()
    assert st["n_unique"] == len(ch) and int(ch["length"].sum()) == n_files * size
    rng = np.random.default_rng(9)
    for i in rng.integers(0, len(ch), 300):
        r = ch[int(i)]
        blob = oracle.synth_fill(0x4D414B49, 50000 + int(r["file_index"]), 0, size)
        seg = blob[int(r["offset"]): int(r["offset"]) + int(r["length"])].tobytes()
        assert bytes(r["sha256"]).hex() == hashlib.sha256(seg).hexdigest()


def test_string_sharing_schemes_agree(oracle):
    """mi_config.sha_sched: the default (the first wave on a SIMD takes the longest quarter of the strings), every
    long-range size from 'all of them' to 'none', and the flat scheme of rounds 1-2 -- on a shape whose string
    lengths span three orders of magnitude, at 1, 2 and 3 workgroups per CU, with both load schemes: identical
    chunk digests, roots and whole-file digests, equal to the oracle's."""
    import numpy as np
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import makisu_amd
    rng = np.random.default_rng(11)
    sizes = [int(x) for x in rng.integers(1, 5000, 3000)] + [1 << 20] * 6 + [40 << 20, 70001, 0, 64, 65]
    cids = list(range(7000, 7000 + len(sizes)))
    ref = None
    variants = [{}, {"sha_sched": makisu_amd.SHA_SCHED_FLAT}]
    variants += [{"sha_sched": makisu_amd.sha_sched_long_shift(k)} for k in (0, 1, 5, 15)]
    variants += [{"sha_sched": makisu_amd.sha_sched_long_shift(3), "sha_blocks_per_cu": 1},
                 {"sha_blocks_per_cu": 3, "sha_load_scheme": makisu_amd.SHA_LOADS_LANE},
                 {"sha_load_scheme": makisu_amd.SHA_LOADS_COOP, "sha_coop_blocks_per_cu": 3},
                 {"sha_load_scheme": makisu_amd.SHA_LOADS_COOP, "sha_sched": makisu_amd.SHA_SCHED_FLAT}]
    for kw in variants:
        with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_SHA256, mask_bits=8, min_size=64, max_size=1 << 16, **kw) as e:
            with e.batch() as b:
                b.add_synthetic(sizes, cids)
                b.run()
                got = (b.chunks()["sha256"].copy(), b.chunks()["length"].copy(), b.files()["chunk_root"].copy(),
                       b.files()["file_sha256"].copy())
        if ref is None:
            ref = got
            # the oracle on the same bytes
            p = oracle.CdcParams(0x4D414B49, 8, 64, 1 << 16)
            blobs = [oracle.synth_fill(0x4D414B49, c, 0, n) for c, n in zip(cids, sizes)]
            data = np.concatenate(blobs) if blobs else np.zeros(0, np.uint8)
            sz = np.array(sizes, dtype=np.uint64)
            offs = np.concatenate([[0], np.cumsum(sz)[:-1]]).astype(np.uint64)
            rf, rc = oracle.scan_batch(data, offs, sz, p, True, 8, 0)
            assert np.array_equal(got[1], rc["length"]) and np.array_equal(got[0], rc["sha256"])
            assert np.array_equal(got[2], rf["chunk_root"])
        else:
            for a, r in zip(got, ref):
                assert np.array_equal(a, r), kw
    with pytest.raises(makisu_amd.MiError):
        makisu_amd.Engine(sha_sched=2)
    with pytest.raises(makisu_amd.MiError):
        makisu_amd.Engine(sha_sched=18 << 8)


WAVES_CHILD = textwrap.dedent(r'''
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tools"))
    import makisu_amd
    import sha_wave_stats as WS
    sizes = [65536] * 20000
    with makisu_amd.Engine() as e, e.batch() as b:
        b.add_synthetic(sizes, None)
        b.run()
        ch = b.chunks().copy()
    recs = WS.read_records(os.environ["MI_SHA_WAVE_STATS"])
    assert len(recs) == 1 and recs[0]["n"] >= len(ch)
    a = WS.analyse(recs[0])
    w = recs[0]["w"]
    # every block of every chunk was hashed by exactly one lane: data blocks + the padding block(s)
    blocks = int(((ch["length"].astype(np.int64) + 8) // 64 + 1).sum())
    assert int(w[:, 6].astype(np.int64).sum()) == blocks, (int(w[:, 6].sum()), blocks)
    assert a["waves"] == recs[0]["grid"] * 4 and set(a["waves_by_role"]) <= {0, 1, 2}
    assert a["waves_by_role"][0] >= a["waves"] // 3          # at least one first-comer per occupied SIMD
    assert 0.5 < a["lane_blocks_share_by_role"][0] <= 1.0    # ... and it does most of the work
    print("OK", json.dumps(a))
''')


def test_wave_stats_record(tmp_path):
    """MI_SHA_WAVE_STATS: one record per chunk-pass launch; the lane-blocks its waves report add up to the
    blocks of the batch's chunks (64-byte blocks incl. padding), roles are 0..2 with the first-comers doing most
    of the hashing."""
    env = dict(os.environ, MI_SHA_WAVE_STATS=str(tmp_path / "waves.bin"))
    r = subprocess.run([sys.executable, "-c", WAVES_CHILD % {"root": ROOT}], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OK" in r.stdout


def test_wave_stats_are_a_ctx_setting(tmp_path):
    """mi_debug_sha_wave_stats (round 4: the record left the launch path's process-wide getenv): one ctx records, another
    of the same process does not; turned off, the ctx stops; the digests are the same either way."""
    import numpy as np
    import makisu_amd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sha_wave_stats as WS
    path = str(tmp_path / "waves.bin")
    sizes = [65536] * 2000
    with makisu_amd.Engine() as quiet, makisu_amd.Engine() as e:
        e.debug_sha_wave_stats(path)
        with e.batch() as b, quiet.batch() as q:
            b.add_synthetic(sizes, None)
            q.add_synthetic(sizes, None)
            q.run()
            assert not os.path.exists(path)                       # the other ctx's launches leave no record
            b.run()
            assert np.array_equal(b.chunks()["sha256"], q.chunks()["sha256"])
            assert len(WS.read_records(path)) == 1
            b.rerun()
            assert len(WS.read_records(path)) == 2
            e.debug_sha_wave_stats(None)
            b.rerun()
            assert len(WS.read_records(path)) == 2


def test_an_arena_of_small_pieces_takes_the_cooperative_loads(tmp_path, monkeypatch):
    """AUTO on an arena mapped piece by piece (a batch that learns its size as it goes: every commit's; csrc/mi_arena.hip):
    the lane-owned loads run at the speed of the page-table fragments behind 32 MiB pieces, so from
    ShaTune::coop_min_bytes_pieces on (moved down to 32 MiB here) the chunk pass loads cooperatively -- the wave record's
    header says which kernel ran; a batch that was told its size (one allocation) of the same content stays lane-owned;
    the rows are the same either way; a ctx told LANE stays lane-owned on pieces too."""
    import numpy as np
    import makisu_amd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sha_wave_stats as WS
    if os.environ.get("MI_ARENA") or os.environ.get("MI_GUARD_ALLOC"):
        pytest.skip("the arena kind is forced by the environment")
    monkeypatch.setenv("MI_SHA_COOP_MIN_GIB_PIECES", "0.03125")
    rng = np.random.default_rng(28)
    blobs = [rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes() for _ in range(48)]
    rows = {}
    for name, scheme, told in (("pieces", makisu_amd.SHA_LOADS_AUTO, False), ("plain", makisu_amd.SHA_LOADS_AUTO, True),
                               ("pieces_lane", makisu_amd.SHA_LOADS_LANE, False)):
        path = str(tmp_path / (name + ".bin"))
        with makisu_amd.Engine(sha_load_scheme=scheme) as e:
            e.debug_sha_wave_stats(path)
            with (e.batch(len(blobs), len(blobs) << 20) if told else e.batch()) as b:
                for x in blobs:
                    b.add_bytes(x)
                b.run()
                rows[name] = b.chunks()["sha256"].copy()
        recs = WS.read_records(path)
        assert len(recs) == 1
        assert recs[0]["coop"] == (1 if name == "pieces" else 0), (name, recs[0]["coop"])
    assert np.array_equal(rows["pieces"], rows["plain"]) and np.array_equal(rows["pieces"], rows["pieces_lane"])
