"""GPU tests of THE SEAM as one flow: mi_memfs_commit_layer with a ctx -- walk -> stage -> Gear CDC + SHA-256 on the GPU ->
createLayerByScan / addToLayer with the chunk roots -> layer tar written from HBM -> DigestPair (+ roots, + chunk index).

Reference behaviour it carries: MemFS.AddLayerByScan / AddLayerByCopyOps (lib/snapshot/mem_fs.go:260-341), isUpdated
(:487-503 -> tario.IsSimilarHeader, lib/tario/compare.go:24-120), step.commitLayer (lib/builder/step/common.go:67-111),
tario.WriteEntry (lib/tario/write.go:28-52); replayed cases: TestCreateLayerByScan (mem_fs_test.go:572-686),
TestAddLayerByScanWhiteout (:1038-1116), C1 = testdata/build-context as a COPY layer.  The oracle (oracle/) is the
checker for every chunk root; hashlib for every digest; python's tarfile for every member."""
import base64
import hashlib
import json
import os
import shutil
import subprocess
import sys
import textwrap
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):                   # (also run as a script, one scenario per process)
    if _p not in sys.path:
        sys.path.insert(0, _p)
import makisu_amd as M  # noqa: E402
from commit_cases import commit_to_bytes, make_tree, oracle_chunks, oracle_root, proc_io, tar_members, write_file  # noqa: E402

pytestmark = pytest.mark.gpu
MTIME = 1_600_000_000


@pytest.fixture(scope="module")
def eng():
    with M.Engine(device=0) as e:
        yield e


def _check_commit_zero(O, eng, root, files, tmp):
    """case (a): every file of a fresh tree in the layer with its bytes; every stored root = the oracle's"""
    with M.MemFS(root) as fs, M.MemFS(root) as plain:
        res, raw = commit_to_bytes(fs, tmp, "l0.tar", must_scan=True, engine=eng)
        members = tar_members(raw)
        got = {name: data for name, m, data in members if m.isfile()}
        assert got == files                                                  # exactly the files, exactly their bytes -- from HBM
        names = [n for n, _, _ in members]
        assert names == sorted(names)                                        # rangeFiles' order
        st = res["stats"]
        nonempty = [d for d in files.values() if d]
        assert st["n_scanned_files"] == len(files) and st["scanned_bytes"] == sum(map(len, nonempty))
        assert st["n_layer_files"] == len(files) and st["layer_file_bytes"] == st["scanned_bytes"]
        if os.environ.get("MI_COMMIT_FORCE_WINDOWS") == "1":                  # as if the tree did not fit the device: the roots window
            assert st["n_windows"] >= 2 and st["file_bytes_read"] == 2 * st["scanned_bytes"]   # by window, the tar from disk
        else:
            assert st["files_opened"] == len(nonempty) and st["file_bytes_read"] == st["scanned_bytes"]   # ONE open, ONE read each
        assert st["n_content_changed"] == 0 and st["n_roots_learned"] == 0
        assert st["n_chunks"] == sum(len(oracle_chunks(O, d)) for d in nonempty)           # (summed over the windows, if any)
        by = {e["relpath"]: e for e in res["layer"]}
        for rel, data in files.items():
            want = oracle_root(O, data)
            assert by[rel]["root"] == want, rel                              # the layer's record of the file
            assert fs.root_of("/" + rel) == want, rel                        # ... and the tree's, for the next isUpdated
        assert fs.root_of("/rel_link") is None and "root" not in by["empty_dir"]
        # ctx == NULL on the same tree: the reference's commit -- the same tar, byte for byte
        res0, raw0 = commit_to_bytes(plain, tmp, "l0_plain.tar", must_scan=True)
        assert raw0 == raw and res0["tar_digest"] == res["tar_digest"]
        assert all("root" not in e for e in res0["layer"]) and res0["stats"]["n_scanned_files"] == 0
        # nothing changed since: the empty layer, both ways
        for f, kw in ((fs, {"engine": eng}), (plain, {})):
            r, b = commit_to_bytes(f, tmp, "l1.tar", must_scan=True, **kw)
            assert r["n_entries"] == 0 and b == bytes(1024)
    return res


def test_commit_zero_every_root_is_the_oracles_and_the_tar_is_the_files(oracle, eng, tmp_path):
    root = str(tmp_path / "root")
    files = make_tree(root, seed=11, mtime=MTIME)
    assert any(len(d) > 16384 for d in files.values()) and any(0 < len(d) <= 16384 for d in files.values())
    assert any(len(d) == 0 for d in files.values())
    _check_commit_zero(oracle, eng, root, files, tmp_path)


@pytest.mark.parametrize("env", [{"MI_WALK_INLINE": "0"}, {"MI_WALK_INLINE_MAX_KIB": "2"}, {"MI_WALK_THREADS": "1"},
                                 {"MI_WALK_INLINE_MB": "1"}, {"MI_WALK_CLOSE_RANGE": "0"},
                                 {"MI_COMMIT_FORCE_WINDOWS": "1", "MI_COMMIT_WINDOW_MB": "1"}])
def test_commit_zero_whichever_way_the_bytes_reach_the_arena(env, tmp_path):
    """case (f): where a file lies in the arena is independent of its row -- small files travel in their directory's block,
    larger ones as paths through the reader threads, in whatever order those finish.  The walk's knobs move that boundary
    (they are read once per process: a process of its own per setting); every root must still land on its node."""
    p = subprocess.run([sys.executable, os.path.abspath(__file__), str(tmp_path)], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK commit_zero" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


def test_copy_sources_that_do_not_fit_the_device_window_by_window(tmp_path):
    """COPY ops as if their sources did not fit the device (MI_COMMIT_FORCE_WINDOWS=1, 2 MiB windows): the plan is made without a
    batch, the files numbered across the ops' walks, the roots computed in windows and = the oracle's; the tar = the header-only
    commit's; a same-second, same-size rewrite of a source is told by its root (tests/hip_stub/commit_scenarios.py, on the GPU)"""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip_stub", "commit_scenarios.py")
    env = dict(os.environ, MI_COMMIT_FORCE_WINDOWS="1", MI_COMMIT_WINDOW_MB="2", MI_TEST_ON_GPU="1")
    p = subprocess.run([sys.executable, script, str(tmp_path), "4", "oversize_copy"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK oversize_copy" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


def test_trusting_ctimes_on_wide_directories_that_change(tmp_path):
    """MI_MEMFS_TRUST_CTIME, the predicate's place among a directory's children (tests/hip_stub/commit_scenarios.py: runs of deletions,
    new names before / between / after the old ones, rewrites that keep size and mtime, two handles in turn), on the GPU: every
    commit reads exactly what is new or changed, and the rewrites are in the layer"""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip_stub", "commit_scenarios.py")
    p = subprocess.run([sys.executable, script, str(tmp_path), "4", "trust_wide"], env=dict(os.environ, MI_TEST_ON_GPU="1"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK trust_wide" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.parametrize("case,fault", [("clean", None), ("readback1", "readback:2"), ("readback3", "readback:3:3"), ("copy", "copy:1"),
                                        ("copy", "copy:5")])
def test_bytes_that_change_on_the_way_fail_the_commit_on_the_gpu(tmp_path, case, fault):
    """The end-to-end byte sums (csrc/mi_filesum.h) with real DMA on both hops (tests/hip_stub/commit_scenarios.py `verify`): nothing
    wrong -- every layer file verified, no chunk fetched twice, the reference's tar; one read-back copy with a flipped byte --
    repaired by the second fetch; three in a row -- MI_ERR_IO naming the hop HBM -> read-back window; 4 KiB lost in HBM after a
    host-to-device copy -- MI_ERR_IO naming the hop pinned slab -> HBM.  Never a layer (lib/tario/write.go:43-45)."""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip_stub", "commit_scenarios.py")
    env = dict(os.environ, MI_TEST_ON_GPU="1", MI_VERIFY_CASE=case)
    if fault:
        env["MI_STAGE_FAULT"] = fault
    p = subprocess.run([sys.executable, script, str(tmp_path), "4", "verify"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and ("OK verify " + case) in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.parametrize("env", [{"MI_VERIFY_STAGING": "1"}, {"MI_COMMIT_VERIFY": "0"}, {"MI_ARENA": "malloc"}, {"MI_ARENA_PIECE_MB": "2"},
                                 {"MI_ARENA_RANGE_MB": "2", "MI_ARENA_PIECE_MB": "2"}])
def test_commit_zero_with_the_heavier_check_without_any_and_on_either_arena(env, tmp_path):
    """commit zero (every root the oracle's, the tar the files') with MI_FLAG_VERIFY_STAGING's per-span GPU sums beside the
    default end-to-end sums; with the end-to-end sums off; with the arena as one allocation that moves (rounds 1-5); with
    2 MiB pieces; and with an address range of 2 MiB that the tree outgrows (the pieces are mapped again in a larger one)"""
    p = subprocess.run([sys.executable, os.path.abspath(__file__), str(tmp_path)], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK commit_zero" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


NO_ADDRESSES = r"""
import ctypes, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import makisu_amd as M
from commit_cases import commit_to_bytes, make_tree, oracle_root
from oracle import mi_oracle as O
O.build()
tmp = sys.argv[1]
root = os.path.join(tmp, "root")
files = make_tree(root, seed=77)
with M.Engine(device=0) as eng:
    # the HIP runtime this process runs on, by the path it was mapped from; then every 32 GiB address range it will give
    path = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][0]
    hip = ctypes.CDLL(path)
    hip.hipMemAddressReserve.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_ulonglong]
    n = 0
    while True:
        va = ctypes.c_void_p()
        if hip.hipMemAddressReserve(ctypes.byref(va), 32 << 30, 1 << 30, None, 0) != 0:
            break
        n += 1
        assert n < 100000
    assert n > 1000, n
    with M.MemFS(root) as fs, M.MemFS(root) as plain:
        res, raw = commit_to_bytes(fs, tmp, "a.tar", must_scan=True, engine=eng)
        res0, raw0 = commit_to_bytes(plain, tmp, "p.tar", must_scan=True)
        st = res["stats"]
        assert raw == raw0 and st["arena_pieces"] == 1 and st["n_verified_files"] == len(files) and st["n_refetched"] == 0, st
        by = {e["relpath"]: e for e in res["layer"]}
        for rel, data in files.items():
            assert by[rel]["root"] == oracle_root(O, data), rel
print("OK no_addresses", n)
"""


def test_a_process_without_address_ranges_left_still_commits(tmp_path):
    """csrc/mi_arena.hip never gives an address range back, and a process has 4 094 of an arena's size (tools/vmm_va_probe.hip,
    profiles/r06_vmm_va_probe.txt).  With all of them taken -- here by the test itself, through the process's own HIP runtime -- a
    walk-fed batch's arena is ONE allocation that moves when it grows, as every batch's was until round 5: the commit is the
    reference's, every root the oracle's."""
    p = subprocess.run([sys.executable, "-c", NO_ADDRESSES % {"root": ROOT}, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK no_addresses" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


def test_warming_a_ctx_changes_nothing_but_when_the_first_use_is_paid(oracle, tmp_path):
    """mi_ctx_warm: reader threads, pinned slabs and the kernels' code objects before the first commit instead of inside it.  The
    commit after it is the commit without it -- same tar, same roots, same statistics of the ctx's own batches --, twice is harmless,
    and it works on a ctx that has already run batches"""
    root = str(tmp_path / "root")
    files = make_tree(root, seed=5)
    with M.Engine(device=0) as cold, M.Engine(device=0) as warm:
        warm.warm()
        warm.warm()
        with M.MemFS(root) as a, M.MemFS(root) as b:
            ra, raw_a = commit_to_bytes(a, str(tmp_path), "cold.tar", must_scan=True, engine=cold)
            rb, raw_b = commit_to_bytes(b, str(tmp_path), "warm.tar", must_scan=True, engine=warm)
        assert raw_a == raw_b and [e.get("root") for e in ra["layer"]] == [e.get("root") for e in rb["layer"]]
        assert {e["relpath"]: e for e in rb["layer"]}[sorted(files)[0]]["root"] == oracle_root(oracle, files[sorted(files)[0]])
        sa, sb = cold.stats(), warm.stats()
        assert (sa["n_files"], sa["n_chunks"], sa["bytes_in"]) == (sb["n_files"], sb["n_chunks"], sb["bytes_in"]), (sa, sb)
        cold.warm()                                                            # after use: nothing to do, and nothing broken
        assert cold.stats() == sa                                              # ... and the ctx's statistics are still about ITS batches
        with M.MemFS(root) as c:
            rc_, raw_c = commit_to_bytes(c, str(tmp_path), "again.tar", must_scan=True, engine=cold)
        assert raw_c == raw_a


def test_the_arena_never_moves_on_the_gpu(tmp_path):
    """tests/hip_stub/commit_scenarios.py `known_tree` with real device memory: 262 MB learned 1 024 files at a time -- one address
    range, pieces mapped behind the walk, every file's bytes in the tar"""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip_stub", "commit_scenarios.py")
    p = subprocess.run([sys.executable, script, str(tmp_path), "4", "known_tree"], env=dict(os.environ, MI_TEST_ON_GPU="1"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK known_tree" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.parametrize("n,extra", [(2, {}), (8, {}), (3, {"MI_COMMIT_PIPELINE": "0"}), (2, {"MI_STAGE_FAULT": "readback:1"}),
                                     (2, {"MI_COMMIT_SPLIT_MIB": "2"}), (4, {"MI_COMMIT_SPLIT_MIB": "2", "MI_COMMIT_PIPELINE": "0"}),
                                     (8, {"MI_COMMIT_SPLIT_MIB": "4", "MI_STAGE_FAULT": "readback:3"})])
def test_the_commit_over_several_ctxs_on_the_gpu(tmp_path, n, extra):
    """mi_memfs_commit_layer_n with n ctxs on the box's one GPU (tests/hip_stub/commit_scenarios.py `many_gpus`): every root = the
    oracle's = the one-ctx commit's, the tar byte-identical to n = 1 and to the header-only commit, one read per file, the chunk
    index (on the LAST ctx: every other ctx's digests reach it through the host) holds what the one-ctx commit's holds, a
    same-second rewrite is caught, a corrupted read-back repaired.  With MI_COMMIT_SPLIT_MIB lowered to 2 / 4 MiB the larger files are
    SPLIT over the ctxs as parts (what happens to files of 256 MiB and more): their roots -- mi_chunk_root over the parts' digests, the
    parts having agreed on their boundary cuts -- are the oracle's roots of the whole files, their bytes in the tar are the files',
    every chunk of them verified.  No claim about more than one physical GPU."""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip_stub", "commit_scenarios.py")
    env = dict(os.environ, MI_TEST_ON_GPU="1", MI_TEST_N_CTXS=str(n), **extra)
    p = subprocess.run([sys.executable, script, str(tmp_path), "4", "many_gpus"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and ("OK many_gpus %d" % n) in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


def test_a_file_of_600_mib_is_split_over_three_ctxs_at_the_real_threshold(oracle, eng, tmp_path):
    """files of 256 MiB and more are split across the GPUs.  One 600 MiB file (+ a 300 MiB one, + small ones) committed over
    three ctxs with the default MI_COMMIT_SPLIT_MIB: two split files, their roots the ORACLE's roots of the whole files, the TarDigest
    the header-only commit's, every file's bytes verified against the sums taken where they were read; a rewrite inside the big file
    that keeps size and second is caught by the next commit"""
    root = str(tmp_path / "root")
    rng = np.random.default_rng(600)
    big = rng.integers(0, 256, 600 << 20, dtype=np.uint8).tobytes()
    mid = rng.integers(0, 256, (300 << 20) + 777, dtype=np.uint8).tobytes()
    files = {"blobs/big.bin": big, "blobs/mid.bin": mid, "etc/a.conf": b"a = 1\n" * 100, "etc/b.bin": rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes()}
    for rel, data in files.items():
        write_file(os.path.join(root, rel), data, 0o644, MTIME)
    for dp, dns, fns in os.walk(root):
        os.utime(dp, (MTIME, MTIME))
    more = [M.Engine(device=0, n_streams=4) for _ in range(2)]
    try:
        with M.MemFS(root) as fs, M.MemFS(root) as plain:
            res = fs.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF, engine=[eng] + more)
            res0 = plain.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF)
            st = res["stats"]
            assert res["tar_digest"] == res0["tar_digest"] and res["tar_bytes"] == res0["tar_bytes"]
            assert st["n_split_files"] == 2 and st["n_ctxs"] == 3 and st["n_verified_files"] == 4 and st["n_refetched"] == 0, st
            assert st["ctx_bytes_max"] <= 2 * st["ctx_bytes_min"], st                                # (200 + 150 MiB against 200: parts of two files)
            by = {e["relpath"]: e for e in res["layer"]}
            for rel, data in files.items():
                assert by[rel]["root"] == oracle_root(oracle, data), rel
            changed = bytearray(big)
            changed[(400 << 20) + 5] ^= 0x10                                   # inside the big file's third part, same size, same second
            write_file(os.path.join(root, "blobs/big.bin"), bytes(changed), 0o644, MTIME)
            os.utime(os.path.join(root, "blobs"), (MTIME, MTIME))
            res = fs.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF, engine=[eng] + more)
            assert [e["relpath"] for e in res["layer"] if e["kind"] == M.KIND_FILE] == ["blobs/big.bin"] and res["stats"]["n_content_changed"] == 1
            assert {e["relpath"]: e for e in res["layer"]}["blobs/big.bin"]["root"] == oracle_root(oracle, bytes(changed))
    finally:
        for e in more:
            e.close()


def _rewrite_same_size_same_second(path, rng):
    st = os.stat(path)
    old = open(path, "rb").read()
    data = old
    while data == old:                                                       # (a one-byte file: one draw in 256 is the byte it holds)
        data = rng.integers(0, 256, st.st_size, dtype=np.uint8).tobytes()
    with open(path, "r+b") as f:
        f.write(data)
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns))
    st2 = os.stat(path)
    assert (st2.st_size, st2.st_mtime_ns, st2.st_mode, st2.st_uid) == (st.st_size, st.st_mtime_ns, st.st_mode, st.st_uid)
    return data


def test_a_same_size_same_second_edit_is_in_the_layer_only_with_the_gpu(oracle, eng, tmp_path):
    """case (b): the edit tario.IsSimilarHeader cannot see (compare.go:101-103 "ignores path and content").  With a ctx the
    next commit holds exactly that file and its ancestors; with ctx == NULL it holds nothing -- the reference's answer."""
    root = str(tmp_path / "root")
    files = make_tree(root, seed=12, mtime=MTIME)
    rng = np.random.default_rng(5)
    with M.MemFS(root) as fs, M.MemFS(root) as plain:
        commit_to_bytes(fs, tmp_path, "a.tar", must_scan=True, engine=eng)
        commit_to_bytes(plain, tmp_path, "b.tar", must_scan=True)
        for rel in ("d01/nested/deeper/f003.bin", "d02/f001.bin"):           # one above the inline limit, one below
            new = _rewrite_same_size_same_second(os.path.join(root, rel), rng)
            assert new != files[rel] and len(new) == len(files[rel])
            files[rel] = new
            res, raw = commit_to_bytes(fs, tmp_path, "a1.tar", must_scan=True, engine=eng)
            parts = rel.split("/")
            want = ["/".join(parts[:k]) for k in range(1, len(parts) + 1)]
            assert [e["relpath"] for e in res["layer"]] == want               # the file + its ancestors, nothing else
            assert res["stats"]["n_content_changed"] == 1 and res["stats"]["n_layer_files"] == 1
            assert [(n, d) for n, m, d in tar_members(raw) if m.isfile()] == [(rel, new)]
            assert res["layer"][-1]["root"] == oracle_root(oracle, new) == fs.root_of("/" + rel)
            res0, raw0 = commit_to_bytes(plain, tmp_path, "b1.tar", must_scan=True)
            assert res0["n_entries"] == 0 and raw0 == bytes(1024)             # invisible without the content scan
            r, b = commit_to_bytes(fs, tmp_path, "a2.tar", must_scan=True, engine=eng)
            assert r["n_entries"] == 0 and b == bytes(1024)                   # and seen once


def test_roots_are_learned_by_the_first_content_scan_of_an_unchanged_tree(oracle, eng, tmp_path):
    """A tree committed WITHOUT a ctx (or merged from a base layer) holds no roots.  The first commit with a ctx finds every
    header unchanged -- an empty layer -- and keeps the roots it computed; from then on content is watched."""
    root = str(tmp_path / "root")
    files = make_tree(root, seed=13, mtime=MTIME)
    rng = np.random.default_rng(6)
    with M.MemFS(root) as fs:
        commit_to_bytes(fs, tmp_path, "p.tar", must_scan=True)                # the reference's commit: no roots
        assert fs.root_of("/d00/f001.bin") is None
        res, raw = commit_to_bytes(fs, tmp_path, "q.tar", must_scan=True, engine=eng)
        assert res["n_entries"] == 0 and res["stats"]["n_roots_learned"] == len(files)
        assert fs.root_of("/d00/f001.bin") == oracle_root(oracle, files["d00/f001.bin"])
        new = _rewrite_same_size_same_second(os.path.join(root, "d00/f001.bin"), rng)
        res, raw = commit_to_bytes(fs, tmp_path, "r.tar", must_scan=True, engine=eng)
        assert [e["relpath"] for e in res["layer"]] == ["d00", "d00/f001.bin"] and res["stats"]["n_roots_learned"] == 0
        assert fs.root_of("/d00/f001.bin") == oracle_root(oracle, new)
    # the same through a base layer's headers: UpdateFromTarReader gives the tree its nodes, the first scan their roots
    with M.MemFS(root) as fs:
        fs.update_from_entries(M.tree_walk(root, root, (), M.TREE_SCAN, full=True)[1:])
        res, _ = commit_to_bytes(fs, tmp_path, "s.tar", must_scan=True, engine=eng)
        assert res["n_entries"] == 0 and res["stats"]["n_roots_learned"] == len(files)


def test_scan_cases_of_the_reference_through_the_one_call(oracle, eng, tmp_path):
    """case (c): TestAddLayerByScanWhiteout + TestCreateLayerByScan's Simple / Symlink / Whiteout, a hard link, a blacklisted
    directory -- the layer's paths are the reference's, whatever computes the diff"""
    root = str(tmp_path / "root")
    for p in ("test1/test2/test3.txt", "test1/test4/test5/test6.txt"):
        write_file(os.path.join(root, p), b"hello", 0o755)
    names = lambda res: ["/" + e["relpath"] for e in res["layer"]]            # noqa: E731
    with M.MemFS(root) as fs:
        res, raw = commit_to_bytes(fs, tmp_path, "w0.tar", must_scan=True, engine=eng)
        assert names(res) == ["/test1", "/test1/test2", "/test1/test2/test3.txt", "/test1/test4", "/test1/test4/test5",
                              "/test1/test4/test5/test6.txt"]
        a, b = res["layer"][2], res["layer"][5]
        assert a["root"] == b["root"] == oracle_root(oracle, b"hello")         # equal content, equal roots
        shutil.rmtree(os.path.join(root, "test1"))
        res, raw = commit_to_bytes(fs, tmp_path, "w1.tar", must_scan=True, engine=eng)
        assert names(res) == ["/.wh.test1"] and res["stats"]["n_scanned_files"] == 0
        assert [(n, m.size) for n, m, _ in tar_members(raw)] == [(".wh.test1", 0)]
        assert fs.entries() == []
    root = str(tmp_path / "root2")
    write_file(os.path.join(root, "test1/test2/test3.txt"), b"hello", 0o755)
    os.symlink("test2/test3.txt", os.path.join(root, "test1/link"))
    os.symlink(os.path.join(root, "test1/test2"), os.path.join(root, "test1/abs"))
    os.link(os.path.join(root, "test1/test2/test3.txt"), os.path.join(root, "test1/hard"))
    write_file(os.path.join(root, "skipme/secret.bin"), b"x" * 5000)
    write_file(os.path.join(root, "test1/.wh..wh.aufs"), b"meta")              # AUFS metadata: shouldSkip
    with M.MemFS(root, blacklist=[os.path.join(root, "skipme")]) as fs:
        res, raw = commit_to_bytes(fs, tmp_path, "x0.tar", must_scan=True, engine=eng)
        by = {"/" + e["relpath"]: e for e in res["layer"]}
        assert sorted(by) == ["/test1", "/test1/abs", "/test1/hard", "/test1/link", "/test1/test2", "/test1/test2/test3.txt"]
        assert by["/test1/link"]["link_target"] == "test2/test3.txt" and by["/test1/abs"]["link_target"] == "/test1/test2"
        assert by["/test1/hard"]["root"] == by["/test1/test2/test3.txt"]["root"] == oracle_root(oracle, b"hello")
        assert res["stats"]["n_scanned_files"] == 2                            # two names of one inode; nothing of skipme
        os.unlink(os.path.join(root, "test1/test2/test3.txt"))
        res, raw = commit_to_bytes(fs, tmp_path, "x1.tar", must_scan=True, engine=eng)
        assert names(res) == ["/test1", "/test1/test2", "/test1/test2/.wh.test3.txt"]
        assert [e["relpath"] for e in fs.entries()] == ["test1", "test1/abs", "test1/hard", "test1/link", "test1/test2"]


def test_c1_build_context_as_a_copy_layer(oracle, eng, tmp_path):
    """case (d): BASELINE.json configs[0]'s tree (testdata/build-context, carried by tests/golden/build_context_c1.json) as
    the layer of `COPY ctx /app/`: members, bytes (SHA-256 per file as the fixture states), roots; the same COPY again adds
    only the destination chain; a same-size same-second edit of one source is in the third layer -- and is not without a
    ctx."""
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "build_context_c1.json")))["entries"]
    src_root, root = str(tmp_path / "context"), str(tmp_path / "root")
    os.makedirs(root)
    for e in gold:
        write_file(os.path.join(src_root, "ctx", e["path"]), base64.b64decode(e["b64"]), 0o644, MTIME)
    for dp, _, _ in os.walk(src_root):
        os.utime(dp, (MTIME, MTIME))
    op = {"src_root": src_root, "srcs": ["ctx"], "dst": "/app/", "uid": 0, "gid": 0}
    with M.MemFS(root, now_sec=MTIME) as fs, M.MemFS(root, now_sec=MTIME) as plain:
        res, raw = commit_to_bytes(fs, tmp_path, "c0.tar", ops=[op], engine=eng)
        res0, raw0 = commit_to_bytes(plain, tmp_path, "c0p.tar", ops=[op])
        assert raw == raw0                                                     # the reference's layer, byte for byte
        members = tar_members(raw)
        assert {n: hashlib.sha256(d).hexdigest() for n, m, d in members if m.isfile()} == {"app/" + e["path"]: e["sha256"] for e in gold}
        st = res["stats"]
        assert st["n_scanned_files"] == 28 and st["scanned_bytes"] == 10355 and st["files_opened"] == sum(1 for e in gold if e["size"])
        for e in res["layer"]:
            if e["kind"] == M.KIND_FILE:
                data = base64.b64decode(next(g["b64"] for g in gold if "app/" + g["path"] == e["relpath"]))
                assert e["root"] == oracle_root(oracle, data) == fs.root_of("/" + e["relpath"]), e["relpath"]
                assert e["src"] == os.path.join(src_root, "ctx", e["relpath"][len("app/"):])
        res, raw = commit_to_bytes(fs, tmp_path, "c1.tar", ops=[op], engine=eng)
        assert [e["relpath"] for e in res["layer"]] == ["app"]                 # addAncestors(inclusive) only
        victim = next(g for g in gold if g["size"] > 100)
        new = _rewrite_same_size_same_second(os.path.join(src_root, "ctx", victim["path"]), np.random.default_rng(9))
        res, raw = commit_to_bytes(fs, tmp_path, "c2.tar", ops=[op], engine=eng)
        got = [(n, d) for n, m, d in tar_members(raw) if m.isfile()]
        assert got == [("app/" + victim["path"], new)] and res["stats"]["n_content_changed"] == 1
        commit_to_bytes(plain, tmp_path, "c1p.tar", ops=[op])
        res0, raw0 = commit_to_bytes(plain, tmp_path, "c2p.tar", ops=[op])
        assert [e["relpath"] for e in res0["layer"]] == ["app"]                # the reference does not see it


def test_copy_ops_errors_keep_their_place(eng, tmp_path):
    """an op whose source does not exist fails the commit with the reference's chain, after the ops before it were applied"""
    src_root, root = str(tmp_path / "context"), str(tmp_path / "root")
    os.makedirs(root)
    write_file(os.path.join(src_root, "a/x.bin"), b"x" * 3000)
    ops = [{"src_root": src_root, "srcs": ["a"], "dst": "/one/"}, {"src_root": src_root, "srcs": ["missing"], "dst": "/two/"}]
    with M.MemFS(root) as fs:
        with pytest.raises(M.MiError) as ei:
            fs.commit_layer(ops=ops, engine=eng)
        assert "failed to generate diff layer: write diffs: create layer by copy ops: stat src" in str(ei.value)
        assert "one/x.bin" in [e["relpath"] for e in fs.entries()]            # op 1 was applied, as in the interleaved loop
        assert fs.commit_layer(ops=ops[:1], engine=eng)["n_entries"] == 1      # the handle stays usable: only /one again


@pytest.mark.parametrize("env,where", [({}, ("create layer by copy ops: copy src",)),
                                       ({"MI_WALK_INLINE": "0"}, ("gpu scan: read ", "to tar writer: staged file")),   # whoever notices first
                                       ({"MI_WALK_INLINE": "0", "MI_COMMIT_PIPELINE": "0"}, ("create layer by copy ops: gpu scan: read ",))])
def test_a_file_shorter_than_its_size_fails_the_commit_wherever_it_is_noticed(env, where, tmp_path):
    """io.CopyN delivers exactly h.Size bytes or an error (lib/tario/write.go:43-45).  A sysfs attribute says 4096 bytes and
    holds a dozen: a COPY of such a directory must fail -- in the walk when its directory readers read the file where they
    list it, in the scan when the reader threads do (MI_WALK_INLINE=0; with the pipelined commit that verdict arrives while
    the tar is being written) -- and the handle stays usable."""
    code = textwrap.dedent(r'''
        import os, sys
        sys.path.insert(0, %r)
        import makisu_amd as M
        src = "/sys/kernel/mm/transparent_hugepage"
        st = os.stat(src + "/enabled")
        assert st.st_size > len(open(src + "/enabled", "rb").read()) > 0
        root = sys.argv[1]
        os.makedirs(root + "/ok")
        open(root + "/ok/f", "wb").write(b"x" * 70000)
        with M.Engine(device=0) as eng, M.MemFS(root) as fs:
            try:
                fs.commit_layer(ops=[{"src_root": "/", "srcs": [src.lstrip("/")], "dst": "/thp/"}], engine=eng)
                print("NO ERROR")
            except M.MiError as e:
                print("ERR", e)
            r = fs.commit_layer(must_scan=True, engine=eng)
            print("THEN", [e["relpath"] for e in r["layer"] if e["relpath"].startswith("ok")], r["stats"]["pipelined"])
    ''') % ROOT
    p = subprocess.run([sys.executable, "-c", code, str(tmp_path / "root")], env=dict(os.environ, **env), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr[-2000:]
    err = [ln for ln in p.stdout.splitlines() if ln.startswith("ERR")]
    assert err and any(w in err[0] for w in where) and "shorter than the size given" in err[0], p.stdout
    then = [ln for ln in p.stdout.splitlines() if ln.startswith("THEN")][0]
    assert "['ok', 'ok/f']" in then and then.endswith("0" if env.get("MI_COMMIT_PIPELINE") == "0" else "1"), then


def test_the_tar_holds_the_bytes_the_root_describes_while_a_file_is_being_rewritten(oracle, eng, tmp_path):
    """One read per file: a writer keeps rewriting a file (same size) while commits run.  Whatever instant the stager
    caught, the tar holds THOSE bytes and the stored root is the oracle's root of exactly them (write.go:43-45 reads a file
    once; a design that reads it a second time for the tar commits bytes its root does not describe)."""
    root = str(tmp_path / "root")
    files = make_tree(root, seed=14, n_dirs=2, mtime=MTIME)
    hot = os.path.join(root, "d00/hot.bin")
    size = 3 << 20
    write_file(hot, bytes(size))
    stop = threading.Event()

    def writer():
        rng = np.random.default_rng(1)
        fd = os.open(hot, os.O_WRONLY)
        while not stop.is_set():
            os.pwrite(fd, rng.integers(0, 256, size, dtype=np.uint8).tobytes(), 0)
        os.close(fd)
    th = threading.Thread(target=writer)
    th.start()
    try:
        seen = set()
        with M.MemFS(root) as fs:
            for k in range(8):
                res, raw = commit_to_bytes(fs, tmp_path, "h%d.tar" % k, must_scan=True, engine=eng)
                mem = {n: d for n, m, d in tar_members(raw) if m.isfile()}
                if "d00/hot.bin" not in mem:
                    continue                                                   # caught between two rewrites: unchanged
                e = next(x for x in res["layer"] if x["relpath"] == "d00/hot.bin")
                assert e["root"] == oracle_root(oracle, mem["d00/hot.bin"]) == fs.root_of("/d00/hot.bin")
                seen.add(e["root"])
        assert len(seen) >= 2                                                  # the file did move under the commits
    finally:
        stop.set()
        th.join()


def test_every_file_is_read_once_as_the_kernel_counts_it(eng, tmp_path):
    """/proc/self/io around a commit of 64 MiB in files of every size: the process read the files' bytes ONCE (rchar), with
    a ctx; the reference's commit reads them once too (the tar writer); the GPU commit must not read them twice."""
    root = str(tmp_path / "root")
    rng = np.random.default_rng(3)
    total = 0
    for i in range(160):
        size = int(rng.integers(100_000, 700_000))
        write_file(os.path.join(root, "d%d/f%03d" % (i % 7, i)), rng.integers(0, 256, size, dtype=np.uint8).tobytes())
        total += size
    with M.MemFS(root) as fs:
        r0, _ = proc_io()
        res = fs.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)
        r1, _ = proc_io()
        assert res["stats"]["file_bytes_read"] == total and res["stats"]["files_opened"] == 160
        assert total <= r1 - r0 <= total + total // 50 + (1 << 20), (r1 - r0, total)


def test_the_commit_feeds_the_chunk_index(oracle, eng, tmp_path):
    """mi_memfs_set_index: the batch of every content-aware commit joins the index (keyvalue.Store seam) -- a second tree
    with the same content is all known chunks"""
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    make_tree(a, seed=15, mtime=MTIME)
    make_tree(b, seed=15, mtime=MTIME)
    with eng.index() as idx, M.MemFS(a) as fa, M.MemFS(b) as fb:
        fa.set_index(idx)
        fb.set_index(idx)
        ra = fa.commit_layer(must_scan=True, engine=eng)
        assert ra["stats"]["n_index_known"] == 0 and ra["stats"]["n_index_new"] == len(idx) > 0
        assert 0 <= ra["stats"]["scanned_bytes"] - ra["stats"]["index_new_bytes"] <= 4   # random files: every byte is news (but two
                                                                                           # of the four 1-byte files may be the same byte)
        rb = fb.commit_layer(must_scan=True, engine=eng)
        assert rb["stats"]["n_index_new"] == 0 and rb["stats"]["n_chunks"] >= rb["stats"]["n_index_known"] > 0
        assert rb["tar_digest"] == ra["tar_digest"] and rb["stats"]["index_new_bytes"] == 0
        # a file that repeats one the index knows (whole) and a new one: only the new one's bytes are news
        write_file(os.path.join(b, "d00/copy_of_f003.bin"), open(os.path.join(b, "d00/f003.bin"), "rb").read(), 0o644, MTIME)
        write_file(os.path.join(b, "d00/fresh.bin"), os.urandom(50_000), 0o644, MTIME)
        rc = fb.commit_layer(must_scan=True, engine=eng)
        assert rc["stats"]["n_layer_files"] == 2 and rc["stats"]["index_new_bytes"] == 50_000


def test_trusting_the_inode_reads_only_what_changed_and_still_catches_the_same_second_rewrite(oracle, eng, tmp_path):
    """MI_MEMFS_TRUST_CTIME: files whose inode (device, number, size, mtime, ctime to the nanosecond) is what it was when
    they were hashed are not read again -- a commit that changed nothing reads nothing -- and the rewrite that keeps size and
    mtime SECOND (and even the mtime to the nanosecond) is still caught: the kernel moved its ctime.  Layers and roots are
    those of a handle that reads everything."""
    import time
    root = str(tmp_path / "root")
    files = make_tree(root, seed=18, mtime=MTIME)
    rng = np.random.default_rng(7)
    time.sleep(0.06)
    with M.MemFS(root) as fast, M.MemFS(root) as full:
        fast.set_options(trust_ctime=True)
        a, raw_a = commit_to_bytes(fast, tmp_path, "fa.tar", must_scan=True, engine=eng)
        b, raw_b = commit_to_bytes(full, tmp_path, "fb.tar", must_scan=True, engine=eng)
        assert raw_a == raw_b and a["stats"]["n_content_trusted"] == 0
        a, raw_a = commit_to_bytes(fast, tmp_path, "fa.tar", must_scan=True, engine=eng)
        assert a["n_entries"] == 0 and a["stats"]["n_content_trusted"] == len(files) and a["stats"]["files_opened"] == 0
        for step, rel in enumerate(("d02/f003.bin", "d01/nested/deeper/f002.bin")):
            new = _rewrite_same_size_same_second(os.path.join(root, rel), rng)          # (restores atime and mtime to the ns)
            files[rel] = new
            time.sleep(0.06)
            a, raw_a = commit_to_bytes(fast, tmp_path, "fa.tar", must_scan=True, engine=eng)
            b, raw_b = commit_to_bytes(full, tmp_path, "fb.tar", must_scan=True, engine=eng)
            assert raw_a == raw_b                                                       # the same layer, byte for byte
            assert [(n, d) for n, m, d in tar_members(raw_a) if m.isfile()] == [(rel, new)]
            st = a["stats"]
            assert st["n_content_changed"] == 1 and st["n_scanned_files"] == 1 and st["files_opened"] == 1, st
            assert st["n_content_trusted"] == len(files) - 1
            assert fast.root_of("/" + rel) == full.root_of("/" + rel) == oracle_root(oracle, new)
        # racily clean: a file hashed within the clock tick of its own last change is not trusted -- it is read once more
        rel = "d00/f002.bin"
        files[rel] = _rewrite_same_size_same_second(os.path.join(root, rel), rng)
        a = fast.commit_layer(must_scan=True, engine=eng)                              # (no pause: hashed right after the write)
        assert a["stats"]["n_scanned_files"] == 1 and [e["relpath"] for e in a["layer"]][-1] == rel
        time.sleep(0.06)
        a = fast.commit_layer(must_scan=True, engine=eng)
        assert a["n_entries"] == 0 and a["stats"]["n_scanned_files"] == 1 and a["stats"]["n_content_trusted"] == len(files) - 1
        a = fast.commit_layer(must_scan=True, engine=eng)
        assert a["n_entries"] == 0 and a["stats"]["n_scanned_files"] == 0 and a["stats"]["n_content_trusted"] == len(files)
        for rel, data in files.items():
            assert fast.root_of("/" + rel) == oracle_root(oracle, data), rel


def test_the_handles_batch_can_be_made_ahead_of_the_first_commit(oracle, eng, tmp_path):
    """mi_memfs_reserve_device: arena and reader threads before the first commit (a ctx's first use is what costs); too small a
    guess only means the arena grows, a handle that already committed is re-sized in place"""
    root = str(tmp_path / "root")
    files = make_tree(root, seed=17, mtime=MTIME)
    total = sum(map(len, files.values()))
    for guess in (total, total // 10):
        with M.MemFS(root) as fs:
            fs.reserve_device(eng, len(files), guess)
            res, raw = commit_to_bytes(fs, tmp_path, "r.tar", must_scan=True, engine=eng)
            assert {n: d for n, m, d in tar_members(raw) if m.isfile()} == files
            assert fs.root_of("/d00/f001.bin") == oracle_root(oracle, files["d00/f001.bin"])
            fs.reserve_device(eng, len(files), 2 * total)
            assert commit_to_bytes(fs, tmp_path, "r2.tar", must_scan=True, engine=eng)[0]["n_entries"] == 0


def test_read_file_serves_any_range_of_any_staged_file(eng):
    """mi_batch_read_file against the bytes that were added: whole files, ranges across window boundaries, backwards"""
    rng = np.random.default_rng(8)
    sizes = [0, 1, 511, 4096, 300_000, 9_000_000, 70_000, 13 << 20]
    blobs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in sizes]
    with eng.batch() as b:
        for d in blobs:
            b.add_bytes(d)
        with pytest.raises(M.MiError):
            b.read_file(0, 0, 0)                                               # not staged yet
        b.run()
        for i in (7, 0, 3, 5, 4, 1, 6, 2):
            assert b.read_file(i, 0, sizes[i]) == blobs[i]
        for _ in range(200):
            i = int(rng.integers(1, len(sizes)))
            off = int(rng.integers(0, sizes[i]))
            n = int(rng.integers(0, min(sizes[i] - off, 3 << 20) + 1))
            assert b.read_file(i, off, n) == blobs[i][off:off + n]
        assert (b.roots() == b.files()["chunk_root"]).all()
        for bad in ((len(sizes), 0, 1), (1, 2, 0), (2, 500, 12)):
            with pytest.raises(M.MiError):
                b.read_file(*bad)


def test_the_layer_writer_takes_content_from_a_batch(eng, tmp_path):
    """mi_layer_add_batch_file: the same tar as mi_layer_add from the path; a size that is not the staged one is refused"""
    rng = np.random.default_rng(4)
    paths = []
    for i, n in enumerate([5, 70_000, 0, 2_500_000]):
        p = str(tmp_path / ("f%d" % i))
        write_file(p, rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        paths.append((p, n))
    E = lambda i: {"relpath": "x/f%d" % i, "kind": M.KIND_FILE, "mode": 0o100644, "mtime_sec": 7, "size": paths[i][1]}   # noqa: E731
    with eng.batch() as b:
        b.add_paths([p for p, _ in paths])
        b.run()
        with M.Layer(gzip_level=M.GZIP_OFF) as la, M.Layer(gzip_level=M.GZIP_OFF) as lb:
            for i, (p, n) in enumerate(paths):
                la.add(E(i), p)
                lb.add_batch_file(E(i), b, i)
            assert la.io_counts() == (4, sum(n for _, n in paths)) and lb.io_counts() == (0, 0)
            with pytest.raises(M.MiError) as ei:
                lb.add_batch_file(dict(E(1), size=69_999), b, 1)
            assert "staged file has 70000" in str(ei.value)
            ra = la.finish()
        with M.Layer(gzip_level=M.GZIP_OFF) as lc:
            for i in range(4):
                lc.add_batch_file(E(i), b, i)
            assert lc.finish()["tar_digest"] == ra["tar_digest"]


def test_the_commit_from_plain_c(eng, tmp_path):
    """case (e): tests/cabi/commit_driver.c -- ctx, MemFS handle, ONE call per commit, the DigestPair and the stats printed;
    the digests are the ones the Python host gets for the same tree"""
    exe = str(tmp_path / "commit_driver")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi", "commit_driver.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "makisu_amd"), "-lmakisu_mi", "-Wl,-rpath," + os.path.join(ROOT, "makisu_amd")])
    root = str(tmp_path / "root")
    make_tree(root, seed=16, mtime=MTIME)
    victim = os.path.join(root, "d02/f001.bin")
    out = subprocess.run([exe, root, victim], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = dict(ln.split(" ", 1) for ln in out.stdout.splitlines())
    assert lines["C0"].split()[0] == lines["P0"].split()[0] != lines["C2"].split()[0]   # with and without a ctx: one tar
    assert lines["C1"].split()[0] == hashlib.sha256(bytes(1024)).hexdigest()   # nothing changed: the empty layer
    c2 = lines["C2"].split()
    assert c2[1] == "entries=2" and c2[2] == "content_changed=1", lines["C2"]  # d02 + the rewritten file
    c3 = lines["P2"].split()
    assert c3[1] == "entries=0", lines["P2"]                                   # the same edit without a ctx: nothing
    n0 = lines["N0"].split()                                                    # three ctxs, one call: the header-only commit's tar
    assert n0[0] == lines["N0p"].split()[0] and n0[2] == "ctxs=3" and n0[1] == lines["N0p"].split()[1], (lines["N0"], lines["N0p"])
    assert int(n0[3].split("=")[1]) > 0


def _content_state(root):
    out = {}
    for dp, dns, fns in os.walk(root):
        for fn in fns:
            p = os.path.join(dp, fn)
            rel = os.path.relpath(p, root)
            out[rel] = ("link", os.readlink(p)) if os.path.islink(p) else ("file", open(p, "rb").read(), os.stat(p).st_mode & 0o7777)
        for dn in dns:
            p = os.path.join(dp, dn)
            out[os.path.relpath(p, root)] = ("link", os.readlink(p)) if os.path.islink(p) else ("dir",)
    return out


def _apply_layer(root, raw):
    """a layer tar onto a root the way a container runtime stacks it: whiteouts delete, everything else replaces"""
    for name, m, data in tar_members(raw):
        path = os.path.join(root, name)
        base = os.path.basename(name.rstrip("/"))
        if base.startswith(".wh."):
            victim = os.path.join(os.path.dirname(path.rstrip("/")), base[4:])
            if os.path.isdir(victim) and not os.path.islink(victim):
                shutil.rmtree(victim)
            elif os.path.lexists(victim):
                os.unlink(victim)
        elif m.isdir():
            os.makedirs(path, exist_ok=True)
        elif m.issym():
            if os.path.lexists(path):
                os.unlink(path)
            os.symlink(m.linkname, path)
        elif m.isfile():
            if os.path.lexists(path):
                os.unlink(path)
            with open(path, "wb") as f:
                f.write(data)
            os.chmod(path, m.mode & 0o7777)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_the_layers_of_a_build_replay_to_the_tree_bytes_included(oracle, eng, tmp_path, seed):
    """The point of a layer: stacked one on the other the layers of a build give the build's tree.  (Stacked the way a
    container runtime does it -- every member replaces what is there.  The reference's own untar, untarOneItem, keeps a file
    that is "already there and similar" by tario.IsSimilarHeader, mem_fs.go:607-612: it has the same blind spot as isUpdated
    and would keep the stale bytes whatever the layer holds.)  A seeded sequence of steps -- new files, deletions,
    renames, chmod, a symlink retargeted, and REWRITES WITHIN THE SAME SECOND at the same size -- is committed step by step with
    a ctx and, beside it, without one.  Replayed, the content-aware layers reproduce every byte after every step, and every
    root the tree holds is the oracle's root of what is on disk; the reference's layers fall behind exactly at the files that
    were rewritten within their second."""
    rng = np.random.default_rng(seed)
    root, r_gpu, r_cpu = str(tmp_path / "root"), str(tmp_path / "replay_gpu"), str(tmp_path / "replay_cpu")
    os.makedirs(r_gpu)
    os.makedirs(r_cpu)
    files = make_tree(root, seed=100 + seed, n_dirs=4, files_per_dir=7, mtime=MTIME)
    hidden = set()                                                           # rewritten within their second since the start
    with M.MemFS(root) as gpu, M.MemFS(root) as cpu:
        if os.environ.get("MI_SOAK_TRUST") == "1":                             # (tools/commit_soak.py: the same property with
            gpu.set_options(trust_ctime=True)                                  #  MI_MEMFS_TRUST_CTIME -- a rewrite moves the ctime)
        for step in range(6):
            if step:
                live = sorted(files)
                for _ in range(int(rng.integers(1, 4))):                       # rewrites the header cannot show
                    rel = live[int(rng.integers(len(live)))]
                    if files[rel]:
                        files[rel] = _rewrite_same_size_same_second(os.path.join(root, rel), rng)
                        hidden.add(rel)
                rel = live[int(rng.integers(len(live)))]                       # an ordinary edit: new size, new mtime
                files[rel] = rng.integers(0, 256, int(rng.integers(1, 90_000)), dtype=np.uint8).tobytes()
                write_file(os.path.join(root, rel), files[rel], 0o640, MTIME + step)
                hidden.discard(rel)
                gone = live[int(rng.integers(len(live)))]                      # a deletion
                if gone != rel:
                    os.unlink(os.path.join(root, gone))
                    files.pop(gone)
                    hidden.discard(gone)
                new = "d00/new_%d_%d.bin" % (seed, step)                       # a new file, sometimes above the inline limit
                files[new] = rng.integers(0, 256, int(rng.integers(0, 60_000)), dtype=np.uint8).tobytes()
                write_file(os.path.join(root, new), files[new], 0o600, MTIME + step)
                if step == 3:
                    os.unlink(os.path.join(root, "rel_link"))
                    os.symlink("d02/f002.bin", os.path.join(root, "rel_link"))
                if step == 4:
                    shutil.rmtree(os.path.join(root, "d02"))
                    for rel in [r for r in files if r.startswith("d02/")]:
                        files.pop(rel)
                        hidden.discard(rel)
                    os.unlink(os.path.join(root, "rel_link"))
            want = _content_state(root)
            res, raw_g = commit_to_bytes(gpu, tmp_path, "g%d.tar" % step, must_scan=True, engine=eng)
            _, raw_c = commit_to_bytes(cpu, tmp_path, "c%d.tar" % step, must_scan=True)
            for rel, data in files.items():
                assert gpu.root_of("/" + rel) == oracle_root(oracle, data), (step, rel)
            _apply_layer(r_gpu, raw_g)
            _apply_layer(r_cpu, raw_c)
            got = _content_state(r_gpu)
            got.pop("abs_link", None)                                          # (createHeader trims the root off an absolute
            exp = dict(want)                                                   #  target: compared apart, in the scan cases)
            exp.pop("abs_link", None)
            assert got == exp, (step, sorted(set(got) ^ set(exp))[:5], [k for k in got if k in exp and got[k] != exp[k]][:5])
            got_cpu = _content_state(r_cpu)
            differs = {k for k in exp if got_cpu.get(k) != exp[k]}
            assert differs == hidden, (step, sorted(differs ^ hidden))          # the reference's layers: stale exactly there
    assert hidden or seed == 0


def test_the_commit_table_of_the_bench_line(eng):
    """tools/commit_layer_bench.commit_e2e (bench.py's `commit_e2e`): three commits with and without the GPU scan -- the
    rewrites within the same second are in the GPU's layer only, everything else agrees"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from commit_layer_bench import commit_e2e
    for n, size in ((3000, 4096), (12, 8 << 20)):
        t = commit_e2e(eng, n, size)
        new, same, some = t["commits"]
        assert new["gpu"]["layer_files"] == new["cpu_header_only"]["layer_files"] == n
        assert new["gpu"]["tar_bytes"] == new["cpu_header_only"]["tar_bytes"]
        assert new["gpu"]["files_read"] == new["cpu_header_only"]["files_read"] == n        # once each, either way
        assert same["gpu"]["layer_entries"] == same["cpu_header_only"]["layer_entries"] == 0
        assert same["gpu"]["files_read"] == n and same["cpu_header_only"]["files_read"] == 0   # the price of watching content
        k = max(1, n // 1000)
        hidden = some["gpu"]["content_only_changes"]
        assert hidden >= 1 and some["gpu"]["layer_files"] == k and some["cpu_header_only"]["layer_files"] == k - hidden
        assert new["gpu_trust_ctime"]["tar_bytes"] == new["gpu"]["tar_bytes"] and new["gpu_trust_ctime"]["files_trusted"] == 0
        assert same["gpu_trust_ctime"]["layer_entries"] == 0 and same["gpu_trust_ctime"]["files_trusted"] >= n - n // 20 - 1   # (the last
        assert same["gpu_trust_ctime"]["files_read"] == n - same["gpu_trust_ctime"]["files_trusted"]       # files written may be racy)
        assert some["gpu_trust_ctime"]["layer_files"] == k and some["gpu_trust_ctime"]["content_only_changes"] == hidden
        assert some["gpu_trust_ctime"]["tar_bytes"] == some["gpu"]["tar_bytes"]
        for row in t["commits"]:
            for side in ("gpu", "gpu_trust_ctime", "cpu_header_only"):
                r = row[side]
                serial = r["s_walk_stage"] + r["s_diff"] + r["s_write"] + (0 if r["scan_overlapped"] else r["s_scan"])
                assert r["s_total"] >= serial - 1e-3 and r["scan_overlapped"] == (side != "cpu_header_only" and r["files_read"] > 0)


@pytest.mark.timeout(900)
def test_the_bench_line_carries_the_products_numbers():
    """bench.py at N = 1 (a small C2): `with_rows_on_host` -- the steps with the result rows delivered -- and `commit_e2e`
    beside roofline and cpu_baseline"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--files", "20000", "--steps", "4", "--warmup", "1",
                          "--no-commit-e2e"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    w = j["with_rows_on_host"]
    assert w["rows_per_step"] == j["config"]["chunks_last_batch"] and w["files_per_step"] == 20000
    assert w["bytes_to_host_per_step"] == 64 * w["rows_per_step"] + 96 * 20000 and 0.3 < w["vs_rows_left_on_device"] < 1.2
    assert j["roofline"]["frac"] > 0 and j["cpu_baseline"]["value"] > 0 and "commit_e2e" not in j
    assert j["config"]["with_rows_ratio"] == w["vs_rows_left_on_device"]                 # ... where the driver's record keeps it


def test_the_commit_table_reaches_the_fields_the_driver_keeps(eng):
    """the driver's BENCH record keeps `config`, `roofline`, `cpu_baseline` in full and every other key as a
    name -- bench.py's summary of `commit_e2e` inside `cpu_baseline.commit_s` is built from the table and stays small"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from commit_layer_bench import commit_e2e
    small, large = commit_e2e(eng, 2000, 4096, all_new_rounds=3), commit_e2e(eng, 6, 8 << 20)
    src = open(os.path.join(ROOT, "bench.py")).read()
    a = src.index("        try:\n            ce = out.get(\"commit_e2e\") or {}")
    b = src.index("        except Exception as e:                                      # noqa: BLE001\n            print(\"bench.py: the summary")
    import textwrap
    out = {"commit_e2e": {"small_files": small, "large_files": large}, "cpu_baseline": {"value": 1.0}, "config": {},
           "with_rows_on_host": {"vs_rows_left_on_device": 0.99}}
    exec(textwrap.dedent(src[a:b]).replace("try:\n", "if True:\n", 1), {"out": out, "round": round})
    c = out["cpu_baseline"]["commit_s"]
    assert len(json.dumps(c)) <= 900 and set(c) >= {"all_new", "nothing_changed", "changed_0p1pct", "all_new_gpu_over_header_only"}
    assert c["all_new"]["large"] == [large["commits"][0][k]["s_total"] for k in ("gpu", "gpu_trust_ctime", "cpu_header_only")]
    # the small tree's "all new": three rounds on fresh handles, each side's best; the last round is the table's own row
    rs = small["commits"][0]["all_new_rounds_s"]
    assert all(len(rs[k]) == 3 and rs[k][-1] == small["commits"][0][k]["s_total"] for k in ("gpu", "gpu_trust_ctime", "cpu_header_only"))
    assert c["all_new"]["small"] == [min(rs[k]) for k in ("gpu", "gpu_trust_ctime", "cpu_header_only")] and "best of 3" in c["order"]
    assert c["all_new_gpu_over_header_only"]["small"] == round(min(rs["gpu"]) / min(rs["cpu_header_only"]), 4)
    assert c["large_all_new_verified"][0] == 6 and c["large_all_new_verified"][1] == 0 and c["large_all_new_verified"][2] == 0
    assert out["config"]["with_rows_ratio"] == 0.99


if __name__ == "__main__":                                                     # one scenario in a process of its own
    from oracle import mi_oracle as O
    O.build()
    tmp = sys.argv[1]
    root = os.path.join(tmp, "root")
    files = make_tree(root, seed=21, mtime=MTIME)
    with M.Engine(device=0) as e:
        _check_commit_zero(O, e, root, files, tmp)
    print("OK commit_zero")
