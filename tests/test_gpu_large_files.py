"""GPU parity for the large-file CDC path (gear_cdc.hip "large files"): 256 KiB groups are cut
speculatively in parallel, validated against the previous group's exit and repaired per file.
Every case below is bit-exact against the oracle's sequential chunker; the inputs are chosen so
that each of the three passes has to do real work:
  - random data: groups re-synchronise after one or two cuts (validation pass only);
  - forced-cut-only data with max_size not dividing the group size: NO group ever re-synchronises,
    the per-file pass re-selects every group from its true entry;
  - low mask_bits: tiles with more than 64 candidates (dense), whose bitmaps are rebuilt;
  - mixtures, so that a repaired stretch is followed by groups whose speculation holds again.
Cut points: parity UNPINNED w.r.t. the reference (it has no CDC); the oracle is this repo's spec.
"""
import numpy as np
import pytest

try:
    import torch  # noqa: F401  (must come before the engine: see test_gpu_parity.py)
except ImportError:
    torch = None

pytestmark = pytest.mark.gpu

SEED = 0x4D414B49
G = 256 * 1024          # group size


def _run(oracle, eng, blobs):
    with eng.batch() as b:
        for i, blob in enumerate(blobs):
            b.add_bytes(blob, tag=i)
        b.run()
        files, chunks = b.files().copy(), b.chunks().copy()
    data = np.frombuffer(b"".join(bytes(x) for x in blobs), dtype=np.uint8)
    sizes = np.array([len(x) for x in blobs], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    c = eng.cfg
    p = oracle.CdcParams(c.gear_seed, c.mask_bits, c.min_size, c.max_size)
    rf, rc = oracle.scan_batch(data, offs, sizes, p, True, 8, 0)
    assert len(chunks) == len(rc), (len(chunks), len(rc))
    assert np.array_equal(chunks["file_index"], rc["file_index"])
    assert np.array_equal(chunks["offset"], rc["offset"]), "cut points differ"
    assert np.array_equal(chunks["length"], rc["length"]), "cut points differ"
    assert np.array_equal(chunks["sha256"], rc["sha256"]), "chunk digests differ"
    assert np.array_equal(files["n_chunks"], rf["n_chunks"])
    assert np.array_equal(files["first_chunk"], rf["first_chunk"])
    assert np.array_equal(files["chunk_root"], rf["chunk_root"])
    assert np.array_equal(chunks["dup_of"], rc["dup_of"])
    # classic streaming chunker as well (hash reset at every cut): same cuts
    for f in range(len(blobs)):
        ends = oracle.cdc_classic(np.frombuffer(bytes(blobs[f]), dtype=np.uint8), p) if len(blobs[f]) else []
        mine = chunks[chunks["file_index"] == f]
        assert np.array_equal(mine["offset"] + mine["length"], np.asarray(ends, dtype=np.uint64))
    return files, chunks


def _rand(oracle, cid, n):
    return oracle.synth_fill(SEED, cid, 0, n).tobytes()


def test_group_boundaries_random(oracle):
    import makisu_amd
    with makisu_amd.Engine() as e:
        sizes = [G - 1, G, G + 1, 2 * G, 2 * G + 1, 3 * G - 64, 7 * G + 12345, 65537, 40 * G + 5]
        _run(oracle, e, [_rand(oracle, 2000 + i, n) for i, n in enumerate(sizes)])


def test_never_resynchronising_forced_cuts(oracle):
    """All-zero files: no candidates, cuts every max_size from the file start.  With max_size =
    100 000 the cut phase differs from group to group, so no speculation is ever right."""
    import makisu_amd
    with makisu_amd.Engine(max_size=100000) as e:
        files, chunks = _run(oracle, e, [bytes(13 * G + 777), bytes(2 * G), bytes(G + 1)])
        assert (chunks["length"][chunks["file_index"] == 0][:-1] == 100000).all()


def test_forced_cuts_that_divide_the_group(oracle):
    import makisu_amd
    with makisu_amd.Engine() as e:                         # max_size 64 KiB divides 256 KiB
        _run(oracle, e, [bytes(9 * G + 5), b"\x07" * (4 * G)])


def test_repair_then_resynchronise(oracle):
    """zeros (forced cuts out of phase) followed by random data and back: the per-file pass repairs
    the zero stretch group by group, the random stretch re-synchronises by itself."""
    import makisu_amd
    with makisu_amd.Engine(max_size=90000) as e:
        blob = _rand(oracle, 3000, 3 * G + 100) + bytes(5 * G + 3000) + _rand(oracle, 3001, 6 * G) + \
            bytes(2 * G) + _rand(oracle, 3002, 70000)
        _run(oracle, e, [blob, _rand(oracle, 3003, 5 * G)])


def test_periodic_content_many_groups(oracle):
    import makisu_amd
    with makisu_amd.Engine() as e:
        period = _rand(oracle, 3100, 4096)
        _run(oracle, e, [period * 700, (b"ab" * 3) * 200000, period[:1000] * 1500])


@pytest.mark.parametrize("mask_bits,min_size,max_size", [
    (0, 64, 64), (0, 64, 4096), (2, 64, 256), (4, 128, 3000), (6, 256, 1024), (9, 512, 10000),
    (10, 1024, 8192), (16, 4096, 262144), (18, 2048, 300000), (32, 2048, 65536), (13, 2048, 1 << 20)])
def test_param_sweep_multi_group(oracle, mask_bits, min_size, max_size):
    """Dense tiles (low mask_bits), chunks longer than a group (max_size > 256 KiB), no candidates at
    all (mask 32) -- on files of several groups."""
    import makisu_amd
    with makisu_amd.Engine(mask_bits=mask_bits, min_size=min_size, max_size=max_size) as e:
        blobs = [_rand(oracle, 3200, 5 * G + 4321), _rand(oracle, 3201, G + 70000), bytes(3 * G + 17),
                 _rand(oracle, 3202, 300), _rand(oracle, 3203, 2 * G)]
        _run(oracle, e, blobs)


def test_many_large_files_in_one_batch(oracle):
    import makisu_amd
    with makisu_amd.Engine() as e:
        rng = np.random.default_rng(11)
        sizes = [int(x) for x in rng.integers(70000, 6 * G, 60)]
        cids = list(range(3300, 3300 + len(sizes)))
        with e.batch() as b:
            b.add_synthetic(sizes, cids, seed=SEED)
            b.run()
            files, chunks = b.files().copy(), b.chunks().copy()
        data = np.concatenate([oracle.synth_fill(SEED, c, 0, n) for c, n in zip(cids, sizes)])
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
        p = oracle.CdcParams(e.cfg.gear_seed, e.cfg.mask_bits, e.cfg.min_size, e.cfg.max_size)
        rf, rc = oracle.scan_batch(data, offs, sizes, p, True, 8, 0)
        assert len(chunks) == len(rc)
        assert np.array_equal(chunks["offset"], rc["offset"]) and np.array_equal(chunks["length"], rc["length"])
        assert np.array_equal(chunks["sha256"], rc["sha256"])
        assert np.array_equal(files["chunk_root"], rf["chunk_root"])
