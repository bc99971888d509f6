"""GPU tests of the host-fed path: reader-thread staging (mi_stage.hip), the per-batch inline
window, and the ctx's child accounting.  Bit-exact against the oracle like every parity test."""
import os

import numpy as np
import pytest

try:
    import torch  # noqa: F401  (must come before the engine: see test_gpu_parity.py)
except ImportError:
    torch = None

pytestmark = pytest.mark.gpu

SEED = 0x4D414B49


def _oracle_rows(oracle, blobs, cfg, threads=8):
    data = np.frombuffer(b"".join(bytes(x) for x in blobs), dtype=np.uint8)
    sizes = np.array([len(x) for x in blobs], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    p = oracle.CdcParams(cfg.gear_seed, cfg.mask_bits, cfg.min_size, cfg.max_size)
    return oracle.scan_batch(data if data.size else np.zeros(1, np.uint8), offs, sizes, p, True, threads, 0)


def _same(files, chunks, rf, rc):
    assert len(chunks) == len(rc)
    assert np.array_equal(chunks["offset"], rc["offset"]) and np.array_equal(chunks["length"], rc["length"])
    assert np.array_equal(chunks["sha256"], rc["sha256"])
    assert np.array_equal(files["chunk_root"], rf["chunk_root"])
    assert np.array_equal(chunks["dup_of"], rc["dup_of"])


def test_two_batches_filled_at_the_same_time(oracle):
    """interleaved mi_batch_add_bytes on two batches of one ctx must not mix bytes."""
    import makisu_amd
    a_blobs = [oracle.synth_fill(SEED, 5000 + i, 0, n).tobytes() for i, n in enumerate([70000, 10, 300000, 65536, 2 << 20])]
    b_blobs = [oracle.synth_fill(SEED, 5100 + i, 0, n).tobytes() for i, n in enumerate([5, 131072, 99999, 3 << 20, 64])]
    with makisu_amd.Engine() as e:
        a, b = e.batch(), e.batch()
        for x, y in zip(a_blobs, b_blobs):
            a.add_bytes(x)
            b.add_bytes(y)
        b.run()
        a.run()
        for batch, blobs in ((a, a_blobs), (b, b_blobs)):
            assert batch.read_back().tobytes() == b"".join(blobs)
            _same(batch.files(), batch.chunks(), *_oracle_rows(oracle, blobs, e.cfg))
        a.free()
        b.free()


def test_host_fed_mix_large_and_small_files(oracle, tmp_path):
    """4 x 1 GiB + many small files through mi_batch_add_path (reader threads,
    several files at once, small files sharing a slab), mixed with inline and large add_bytes."""
    import makisu_amd
    big = 1 << 30
    sizes = [big, 700, big, 65536, 0, big, 1, 4097, big] + [int(x) for x in
             np.random.default_rng(4).integers(1, 200000, 300)]
    cids = list(range(6000, 6000 + len(sizes)))
    data, offs = oracle.synth_fill_many(SEED + 1, cids, sizes, 8)
    paths = []
    for i, (o, n) in enumerate(zip(offs, sizes)):
        pth = str(tmp_path / ("f%04d" % i))
        data[int(o):int(o) + n].tofile(pth)
        paths.append(pth)
    extra = [oracle.synth_fill(SEED, 6500, 0, 5 << 20).tobytes(), b"tiny", oracle.synth_fill(SEED, 6501, 0, 90000).tobytes()]
    with makisu_amd.Engine() as e, e.batch() as b:
        for i, (pth, n) in enumerate(zip(paths, sizes)):
            b.add_path(pth, n, i)
            if i == 5:
                for x in extra:                     # caller memory between the files
                    b.add_bytes(x)
        b.run()
        files, chunks = b.files().copy(), b.chunks().copy()
        st = e.stats()
    blobs = []
    for i, (o, n) in enumerate(zip(offs, sizes)):
        blobs.append(data[int(o):int(o) + n])
        if i == 5:
            blobs += [np.frombuffer(x, dtype=np.uint8) for x in extra]
    all_sizes = np.array([len(x) for x in blobs], dtype=np.uint64)
    whole = np.concatenate(blobs)
    all_offs = np.concatenate([[0], np.cumsum(all_sizes)[:-1]]).astype(np.uint64)
    p = oracle.CdcParams(e.cfg.gear_seed, e.cfg.mask_bits, e.cfg.min_size, e.cfg.max_size)
    rf, rc = oracle.scan_batch(whole, all_offs, all_sizes, p, True, 16, 0)
    _same(files, chunks, rf, rc)
    assert st["ms_h2d"] > 0
    for pth in paths:
        os.unlink(pth)


def test_file_that_shrinks_after_add_fails_the_run(tmp_path):
    import makisu_amd
    pth = str(tmp_path / "shrinks")
    with open(pth, "wb") as f:
        f.write(os.urandom(3 << 20))
    with makisu_amd.Engine(n_streams=1) as e:
        with e.batch() as b:
            with pytest.raises(makisu_amd.MiError) as ei:      # too short at add time: this call fails
                b.add_path(pth, (3 << 20) + 1)
            assert ei.value.code == -5
        with e.batch() as b:
            blocker = os.urandom(64 << 20)
            b.add_bytes(blocker)                               # keeps the one reader thread busy
            b.add_path(pth, 3 << 20)
            os.truncate(pth, 100)                              # usually before the reader gets there
            try:
                b.run()
                ok = True
            except makisu_amd.MiError as err:
                ok = False
                assert err.code == -5 and "shorter" in str(err)
            if ok:                                             # the reader won the race: bytes are the old ones
                assert b.counts()[0] == 2


def test_ctx_destroy_refuses_with_live_children():
    import ctypes as C
    import makisu_amd
    L = makisu_amd.load_library()
    e = makisu_amd.Engine()
    b = e.batch()
    idx = e.index()
    assert L.mi_ctx_destroy(e._h) == -6                        # MI_ERR_STATE, nothing freed
    assert b"still alive" in L.mi_last_error(e._h)
    b.add_bytes(b"x" * 1000)
    b.run()                                                    # the ctx is still fully usable
    assert b.counts() == (1, 1, 1000)
    b.free()
    assert L.mi_ctx_destroy(e._h) == -6                        # the index is still there
    idx.free()
    e.close()
    assert e._h is None
    assert L.mi_ctx_destroy(None) == 0


def test_batch_reset_reuses_the_buffers(oracle, tmp_path):
    """mi_batch_reset: layer after layer through ONE batch (no reallocation), results of every pass
    bit-exact and independent of what the batch held before."""
    import makisu_amd
    with makisu_amd.Engine() as e, e.batch() as b:
        for rnd, sizes in enumerate(([300000, 10, 70000], [5], [2 << 20, 0, 4097, 65536, 99], [64])):
            blobs = [oracle.synth_fill(SEED, 7000 + 10 * rnd + i, 0, n).tobytes() for i, n in enumerate(sizes)]
            b.reset()
            assert b.counts() == (0, 0, 0)
            for i, x in enumerate(blobs):
                if i % 2 == 0:
                    b.add_bytes(x, tag=i)
                else:
                    pth = tmp_path / ("r%d_%d" % (rnd, i))
                    pth.write_bytes(x)
                    b.add_path(str(pth), len(x), i)
            b.run()
            assert b.read_back().tobytes() == b"".join(blobs)
            _same(b.files(), b.chunks(), *_oracle_rows(oracle, blobs, e.cfg))
        b.reset()
        b.add_synthetic([65536, 200000], [1, 2], seed=SEED)      # synthetic after host-fed, same buffers
        b.run()
        data = [oracle.synth_fill(SEED, c, 0, n).tobytes() for c, n in ((1, 65536), (2, 200000))]
        _same(b.files(), b.chunks(), *_oracle_rows(oracle, data, e.cfg))
        b.submit()
        with pytest.raises(makisu_amd.MiError):
            b.reset()                                            # not while in flight
        b.wait()


def test_bulk_add_paths_and_tree_walk_with_deferred_opens(oracle, tmp_path):
    """mi_batch_add_paths / mi_batch_add_tree: thousands of small files opened by the reader threads
    (nothing is opened in the adding call), mixed with other kinds of adds; bit-exact against the
    oracle; the files' order is the adding order; a file that is missing or shorter than announced
    fails the RUN with MI_ERR_IO naming it."""
    import makisu_amd
    rng = np.random.default_rng(12)
    sizes = [int(x) for x in rng.integers(0, 20000, 3000)] + [70000, 3 << 20, 0, 9 << 20, 1]
    blobs = [oracle.synth_fill(SEED, 7000 + i, 0, n).tobytes() for i, n in enumerate(sizes)]
    root = tmp_path / "tree"
    paths = []
    for i, blob in enumerate(blobs):
        d = root / ("d%02d" % (i % 37))
        d.mkdir(parents=True, exist_ok=True)
        p = d / ("f%05d" % i)
        p.write_bytes(blob)
        paths.append(str(p))
    with makisu_amd.Engine() as e:
        with e.batch() as b:
            b.add_bytes(b"inline first", tag=1)
            b.add_paths(paths[:2000], [len(x) for x in blobs[:2000]], tags=list(range(2000)))
            b.add_path(paths[2000])                                   # the eager form in between
            b.add_paths(paths[2001:])                                 # sizes from stat
            b.run()
            fl = b.files().copy()
            assert fl["user_tag"][1:2001].tolist() == list(range(2000))
            _same(fl, b.chunks(), *_oracle_rows(oracle, [b"inline first"] + blobs, e.cfg))
        # the library's own walk uses the same deferred path: filepath.Walk order
        with e.batch() as b:
            ents = b.tree_entries(b.add_tree(str(root)))          # (relpath, link, file_index, size, kind, mode)
            order = [os.path.join(str(root), en[0]) for en in ents if en[4] == makisu_amd.KIND_FILE]
            assert [en[2] for en in ents if en[4] == makisu_amd.KIND_FILE] == list(range(len(order)))
            assert sorted(order) == sorted(paths) and len(order) == len(paths)
            b.run()
            by_path = dict(zip(paths, blobs))
            _same(b.files(), b.chunks(), *_oracle_rows(oracle, [by_path[p] for p in order], e.cfg))
        # errors surface at run time
        with e.batch() as b:
            b.add_paths([paths[0], str(root / "missing"), paths[1]], [len(blobs[0]), 10, len(blobs[1])])
            with pytest.raises(makisu_amd.MiError, match="missing"):
                b.run()
        with e.batch() as b:
            b.add_paths([paths[5]], [len(blobs[5]) + 100])
            with pytest.raises(makisu_amd.MiError, match="shorter"):
                b.run()


def test_tree_walk_blocks_verified_and_reused(oracle, tmp_path):
    """Round 4: the walk's directory readers read files up to 16 KiB where they list them -- one block of host memory per
    directory, one piece of the arena each -- and larger ones still go to the reader threads as paths.  A nested tree that
    mixes both (and empty files, and directories with a single file) through ONE batch that is reset and walked three
    times, every staged span verified (MI_FLAG_VERIFY_STAGING sums the blocks' copies like any other): bit-exact against
    the oracle in filepath.Walk order each time, no mismatching span."""
    import makisu_amd
    rng = np.random.default_rng(21)
    root = tmp_path / "tree"
    blobs = {}
    k = 0
    for d in ["", "a", "a/b", "a/b/c", "z z", "m", "m/only"]:
        (root / d).mkdir(parents=True, exist_ok=True)
        n_files = 1 if d == "m/only" else 400
        for _ in range(n_files):
            n = int(rng.choice([0, 1, 255, 256, 4096, 12000, 16384, 16385, 70000, 400000], p=[.05, .1, .1, .1, .3, .2, .05, .04, .04, .02]))
            p = root / d / ("f%05d" % k)
            data = oracle.synth_fill(SEED, 9000 + k, 0, n).tobytes()
            p.write_bytes(data)
            blobs[str(p)] = data
            k += 1
    with makisu_amd.Engine(flags=makisu_amd.FLAG_VERIFY_STAGING) as e:
        b = e.batch()
        for rep in range(3):
            b.reset()
            ents = b.tree_entries(b.add_tree(str(root)))
            order = [os.path.normpath(os.path.join(str(root), en[0])) for en in ents if en[4] == makisu_amd.KIND_FILE]
            assert len(order) == len(blobs) and [en[2] for en in ents if en[4] == makisu_amd.KIND_FILE] == list(range(len(order)))
            b.run()
            _same(b.files(), b.chunks(), *_oracle_rows(oracle, [blobs[p] for p in order], e.cfg))
            st = b.stage_stats()
            assert st["spans"] > 0 and st["verified_spans"] == st["spans"] and st["mismatches"] == 0 and st["final_mismatches"] == 0, st
        b.free()


def test_reader_run_never_writes_over_an_inline_region(oracle, tmp_path):
    """A reader thread copies a run of queued files as ONE span; a small mi_batch_add_bytes that lies
    BETWEEN two such files travels through the batch's inline window instead.  With one (busy) reader
    thread the span is copied after the inline window was flushed: it must not carry zeros over it."""
    import makisu_amd
    big = oracle.synth_fill(SEED, 7100, 0, 200 << 20).tobytes()
    f1 = oracle.synth_fill(SEED, 7101, 0, 30000).tobytes()
    mid = oracle.synth_fill(SEED, 7102, 0, 50000).tobytes()          # inline (below 1 MiB)
    f2 = oracle.synth_fill(SEED, 7103, 0, 40000).tobytes()
    for name, blob in (("big", big), ("f1", f1), ("f2", f2)):
        (tmp_path / name).write_bytes(blob)
    for rep in range(3):
        with makisu_amd.Engine(n_streams=1) as e, e.batch() as b:
            b.add_path(str(tmp_path / "big"))                        # keeps the only reader busy
            b.add_path(str(tmp_path / "f1"))
            b.add_bytes(mid)
            b.add_path(str(tmp_path / "f2"))
            b.run()
            assert b.read_back().tobytes() == big + f1 + mid + f2
            _same(b.files(), b.chunks(), *_oracle_rows(oracle, [big, f1, mid, f2], e.cfg))


# ---- MI_FLAG_VERIFY_STAGING: staged bytes check themselves ---------------------
def _mix_inputs(oracle, tmp_path, n_small=300):
    """The host-fed mix of test_host_fed_mix_large_and_small_files as files on disk + blobs in add order."""
    big = 1 << 30
    sizes = [big, 700, big, 65536, 0, big, 1, 4097, big] + [int(x) for x in
             np.random.default_rng(4).integers(1, 200000, n_small)]
    cids = list(range(6000, 6000 + len(sizes)))
    data, offs = oracle.synth_fill_many(SEED + 1, cids, sizes, 8)
    paths = []
    for i, (o, n) in enumerate(zip(offs, sizes)):
        pth = str(tmp_path / ("f%04d" % i))
        data[int(o):int(o) + n].tofile(pth)
        paths.append(pth)
    extra = [oracle.synth_fill(SEED, 6500, 0, 5 << 20).tobytes(), b"tiny", oracle.synth_fill(SEED, 6501, 0, 90000).tobytes()]
    blobs = []
    for i, (o, n) in enumerate(zip(offs, sizes)):
        blobs.append(data[int(o):int(o) + n])
        if i == 5:
            blobs += [np.frombuffer(x, dtype=np.uint8) for x in extra]
    return paths, sizes, extra, blobs


def _fill_mix(b, paths, sizes, extra):
    for i, (pth, n) in enumerate(zip(paths, sizes)):
        b.add_path(pth, n, i)
        if i == 5:
            for x in extra:                     # caller memory between the files
                b.add_bytes(x)


def _first_difference(got, blobs):
    """Where the staged bytes differ from what was handed in: (file index, first, last, n, 'zeros'|'0xA5'|'other')."""
    pos = 0
    for f, want in enumerate(blobs):
        n = len(want)
        g = got[pos:pos + n]
        pos += n
        neq = np.nonzero(g != want)[0]
        if len(neq):
            vals = g[neq]
            kind = "zeros" if not vals.any() else "0xA5" if (vals == 0xA5).all() else "other"
            return f, int(neq[0]), int(neq[-1]), len(neq), kind
    return None


def test_verify_staging_clean_run_counts_every_span(oracle, tmp_path):
    """With the flag every copy is summed on both sides (readers: right after the copy; everything:
    once more when staging ends, after the arena has grown three times under this batch)."""
    import makisu_amd
    paths, sizes, extra, blobs = _mix_inputs(oracle, tmp_path, n_small=120)
    with makisu_amd.Engine(flags=makisu_amd.FLAG_VERIFY_STAGING) as e, e.batch() as b:
        _fill_mix(b, paths, sizes, extra)
        b.run()
        ss = b.stage_stats()
        files, chunks = b.files().copy(), b.chunks().copy()
        assert ss["mismatches"] == 0 and ss["final_mismatches"] == 0 and ss["note"] == "", ss
        assert ss["spans"] >= 4 * 128 and ss["bytes"] >= sum(len(x) for x in blobs)
        assert ss["spans"] - 2 <= ss["verified_spans"] < ss["spans"]   # all but the inline window's flush
        assert ss["final_spans"] == ss["spans"]
        all_sizes = np.array([len(x) for x in blobs], dtype=np.uint64)
        all_offs = np.concatenate([[0], np.cumsum(all_sizes)[:-1]]).astype(np.uint64)
        p = oracle.CdcParams(e.cfg.gear_seed, e.cfg.mask_bits, e.cfg.min_size, e.cfg.max_size)
        _same(files, chunks, *oracle.scan_batch(np.concatenate(blobs), all_offs, all_sizes, p, True, 16, 0))
        b.reset()
        assert b.stage_stats()["spans"] == 0


FAULT_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch  # noqa: F401
import makisu_amd
rng = np.random.default_rng(3)
blobs = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (20 << 20, 70000, 3 << 20, 9 << 20)]
with makisu_amd.Engine(flags=makisu_amd.FLAG_VERIFY_STAGING if %(flag)d else 0, n_streams=2) as e, e.batch() as b:
    for x in blobs:
        b.add_bytes(x)
    try:
        b.run()
        print("RAN", b.read_back().tobytes() == b"".join(blobs))
    except makisu_amd.MiError as err:
        print("ERR", err.code, err)
        try:
            b.run()
        except makisu_amd.MiError as err2:
            print("AGAIN", err2.code)
    ss = b.stage_stats()
    print("STATS", ss["mismatches"], ss["repaired"], ss["final_mismatches"])
    print("NOTE", ss["note"])
"""


def _fault_child(fault, flag=1):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI_STAGE_FAULT=fault, MI_VERIFY_STAGING=str(flag))
    r = subprocess.run([sys.executable, "-c", FAULT_CHILD % {"root": root, "flag": flag}], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_verify_staging_repairs_a_span_lost_right_after_its_copy():
    """Fault injection (MI_STAGE_FAULT=copy:N zeroes 4 KiB of the N-th span behind its copy): the reader's
    own check sees it, says what the GPU held, copies the slab again -- the run is right."""
    out = _fault_child("copy:2")
    assert "RAN True" in out and "STATS 1 1 0" in out, out
    assert "reader thread" in out and "byte(s) differ in [+" in out and "the GPU holds zeros there" in out, out
    assert "a second copy from the slab matched" in out, out
    # without the flag nobody notices: the bytes are wrong and the run "succeeds" -- what the flag is for
    out = _fault_child("copy:2", flag=0)
    assert "RAN False" in out and "STATS 0 0 0" in out, out


def test_verify_staging_fails_a_span_lost_later_and_the_failure_is_sticky():
    """A span that verified after its copy but differs when staging ends cannot be repaired (the slab
    is gone): MI_ERR_IO naming the range, and the batch keeps failing until it is reset."""
    out = _fault_child("final:1")
    assert "ERR -5" in out and "AGAIN -5" in out and "STATS 0 0 1" in out, out
    assert "end of staging" in out and "no longer hold what was copied" in out, out


def test_staging_errors_are_sticky_until_reset(oracle, tmp_path):
    """a missing file fails the run -- and the NEXT run of the same batch too (it used to
    scan the uninitialised arena region and return MI_OK); mi_batch_reset clears it."""
    import makisu_amd
    good = oracle.synth_fill(SEED, 7200, 0, 300000).tobytes()
    (tmp_path / "good").write_bytes(good)
    with makisu_amd.Engine() as e, e.batch() as b:
        b.add_paths([str(tmp_path / "good"), str(tmp_path / "missing")], [len(good), 10])
        for _ in range(3):
            with pytest.raises(makisu_amd.MiError, match="missing") as ei:
                b.run()
            assert ei.value.code == -5
        with pytest.raises(makisu_amd.MiError, match="missing"):
            b.submit()
        with pytest.raises(makisu_amd.MiError):
            b.wait()                                               # nothing was submitted
        with pytest.raises(makisu_amd.MiError, match="missing"):
            b.scan_cuts()
        b.reset()
        b.add_path(str(tmp_path / "good"))
        b.run()
        _same(b.files(), b.chunks(), *_oracle_rows(oracle, [good], e.cfg))


def soak(oracle, tmp_path, rounds, verify, big_free_every=10, log=print):
    """The host-fed mix, `rounds` times: fresh batches (the arena grows three times under the readers),
    a reset-and-reuse batch, and -- every big_free_every rounds -- right after a batch with a >= 100 GB
    arena was freed (the window in which the driver is clearing freed VRAM).  Returns
    dict(rounds, bad_rounds, mismatches, repaired, notes)."""
    import makisu_amd
    paths, sizes, extra, blobs = _mix_inputs(oracle, tmp_path)
    flags = makisu_amd.FLAG_VERIFY_STAGING if verify else 0
    res = {"rounds": 0, "bad_rounds": 0, "mismatches": 0, "repaired": 0, "final_mismatches": 0, "notes": []}
    ref = None
    with makisu_amd.Engine(flags=flags) as e:
        reused = e.batch()
        for rnd in range(rounds):
            if big_free_every and rnd % big_free_every == big_free_every - 1:
                free_b = int(torch.cuda.mem_get_info()[0]) if torch is not None else 0
                want = min(110 << 30, free_b - (24 << 30))
                if want > (8 << 30):
                    big = e.batch(bytes_hint=want)                 # arena_reserve: one allocation of that size
                    big.add_bytes(b"x" * 4096)
                    big.run()
                    big.free()                                     # ... and straight into the next round
            b = reused.reset() if rnd % 3 == 2 else e.batch()
            try:
                _fill_mix(b, paths, sizes, extra)
                b.run()
                roots = b.files()["chunk_root"].copy()
                ss = b.stage_stats()
                res["mismatches"] += ss["mismatches"]
                res["repaired"] += ss["repaired"]
                res["final_mismatches"] += ss["final_mismatches"]
                if ss["note"]:
                    res["notes"].append("round %d: %s" % (rnd, ss["note"]))
                if ref is None:
                    ref = roots
                    all_sizes = np.array([len(x) for x in blobs], dtype=np.uint64)
                    all_offs = np.concatenate([[0], np.cumsum(all_sizes)[:-1]]).astype(np.uint64)
                    p = oracle.CdcParams(e.cfg.gear_seed, e.cfg.mask_bits, e.cfg.min_size, e.cfg.max_size)
                    rf, rc = oracle.scan_batch(np.concatenate(blobs), all_offs, all_sizes, p, True, 16, 0)
                    _same(b.files(), b.chunks(), rf, rc)
                elif not np.array_equal(roots, ref):
                    res["bad_rounds"] += 1
                    d = _first_difference(b.read_back(), blobs)
                    res["notes"].append("round %d (%s batch): WRONG RESULT, %d file root(s) differ; staged bytes: %s"
                                        % (rnd, "reused" if b is reused else "fresh",
                                           int((roots != ref).any(axis=1).sum()),
                                           "identical to the input (scan-side?)" if d is None else
                                           "file %d differs in [%d, %d], %d bytes, GPU holds %s" % d))
            except makisu_amd.MiError as err:
                res["bad_rounds"] += 1
                res["notes"].append("round %d: %s" % (rnd, err))
                ss = b.stage_stats()
                res["mismatches"] += ss["mismatches"]
                res["final_mismatches"] += ss["final_mismatches"]
                if b is reused:
                    b.reset()
            finally:
                if b is not reused:
                    b.free()
            res["rounds"] += 1
        reused.free()
    for n in res["notes"]:
        log(n)
    return res


@pytest.mark.parametrize("verify", [False, True])
def test_host_fed_soak(oracle, tmp_path, verify):
    """>= 50 rounds of the 4 x 1 GiB + 300-small mix (25 per mode here; the long
    runs are tools/stage_soak.py, results in profiles/), including the first copies into fresh VRAM
    right after a >= 100 GB arena was freed.  Every round must give the first round's (oracle-checked)
    roots; with the flag no span may have needed its second copy."""
    rounds = int(os.environ.get("MI_SOAK_ROUNDS", "25"))
    res = soak(oracle, tmp_path, rounds, verify)
    assert res["bad_rounds"] == 0, res
    assert res["final_mismatches"] == 0, res
    if res["mismatches"]:                                          # repaired: right results, but on record
        import warnings
        warnings.warn("staging verification repaired %d span(s): %s" % (res["repaired"], res["notes"]))
