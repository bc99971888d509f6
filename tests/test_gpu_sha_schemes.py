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
