"""Scenarios for tests/test_host_commit_double.py: the HOST side of the content-aware commit (mi_memfs_commit_layer with a
ctx) on the HIP test double.  Kernels do not run there -- every chunk root is the double's fill pattern, all alike -- so
what is checked is what the host answers for: the walk hands every file to ONE batch, each file is opened and read once,
the diff runs on the batch's rows, and the layer writer takes every file's bytes out of the arena (device memory = host
memory here) from the place the file table says, whatever order the bytes arrived in.  The tar must equal the reference's
commit (ctx == NULL), byte for byte.  Run with LD_PRELOAD=<the double>; prints one "OK <name>" line per scenario."""
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import makisu_amd as M  # noqa: E402
from commit_cases import commit_to_bytes, make_tree, proc_io, tar_members, write_file  # noqa: E402

MTIME = 1_600_000_000


def scenario_scan(tmp, eng):
    root = os.path.join(tmp, "scan_root")
    files = make_tree(root, seed=31, mtime=MTIME)
    nonempty = [d for d in files.values() if d]
    with M.MemFS(root) as fs, M.MemFS(root) as plain:
        with M.MemFS(root) as probe:                                       # (digests only: nothing of the tar is read back here)
            r0, _ = proc_io()
            probe.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)
            r1, _ = proc_io()
        res, raw = commit_to_bytes(fs, tmp, "s0.tar", must_scan=True, engine=eng)
        assert {n: d for n, m, d in tar_members(raw) if m.isfile()} == files
        st = res["stats"]
        total = sum(map(len, nonempty))
        assert (st["n_scanned_files"], st["scanned_bytes"]) == (len(files), total), st
        assert (st["files_opened"], st["file_bytes_read"]) == (len(nonempty), total), st      # one open, one read per file
        assert total <= r1 - r0 <= total + (256 << 10), (r1 - r0, total)                        # ... as the kernel counts it
        assert st["n_layer_files"] == len(files) and st["n_layer_entries"] == res["n_entries"] and st["layer_file_bytes"] == total
        res0, raw0 = commit_to_bytes(plain, tmp, "s0p.tar", must_scan=True)
        assert raw0 == raw                                                                      # the reference's tar
        assert res0["stats"]["files_opened"] == len(files) and res0["stats"]["file_bytes_read"] == total   # (its writer reads once
                                                                                                             #  too, and opens empty files)
        import ctypes
        big_mallocs = ctypes.CDLL(None).mi_hip_stub_big_mallocs                      # (the double counts allocations of a MiB and more)
        big_mallocs.restype = ctypes.c_long
        n0 = big_mallocs()
        res, raw = commit_to_bytes(fs, tmp, "s1.tar", must_scan=True, engine=eng)
        assert res["n_entries"] == 0 and raw == bytes(1024)
        assert big_mallocs() == n0, "the handle's batch is kept between commits: no new arena, no new tables"

        # a changed file (new mtime, new size): that file + its ancestors, its bytes from the arena of the REUSED batch
        rel = "d03/nested/deeper/f003.bin"
        new = os.urandom(len(files[rel]) + 77)
        write_file(os.path.join(root, rel), new, 0o755, MTIME + 5)
        os.utime(os.path.join(root, "d03/nested/deeper"), (MTIME, MTIME))
        res, raw = commit_to_bytes(fs, tmp, "s2.tar", must_scan=True, engine=eng)
        assert [e["relpath"] for e in res["layer"]] == ["d03", "d03/nested", "d03/nested/deeper", rel]
        assert [(n, d) for n, m, d in tar_members(raw) if m.isfile()] == [(rel, new)]
        res0, raw0 = commit_to_bytes(plain, tmp, "s2p.tar", must_scan=True)
        assert raw0 == raw
        # a deleted directory: one whiteout, nothing scanned for it
        import shutil
        shutil.rmtree(os.path.join(root, "d04"))
        os.utime(root, (MTIME, MTIME))
        res, raw = commit_to_bytes(fs, tmp, "s3.tar", must_scan=True, engine=eng)
        assert [e["relpath"] for e in res["layer"]] == [".wh.d04"]
        res0, raw0 = commit_to_bytes(plain, tmp, "s3p.tar", must_scan=True)
        assert raw0 == raw
        # a handle that never saw a ctx holds no roots; its first commit with one learns them (the layer stays empty)
        assert plain.root_of("/d00/f001.bin") is None
        r = plain.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)
        assert r["n_entries"] == 0 and r["stats"]["n_roots_learned"] == len(files) - 9 and plain.root_of("/d00/f001.bin") is not None
        # the handle gives the device back and takes a fresh batch for the next commit
        fs.release_device()
        res, raw = commit_to_bytes(fs, tmp, "s4.tar", must_scan=True, engine=eng)
        assert res["n_entries"] == 0 and res["stats"]["n_scanned_files"] == len(files) - 9
    print("OK scan")


def scenario_copy(tmp, eng):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "build_context_c1.json")))["entries"]
    src_root, root = os.path.join(tmp, "context"), os.path.join(tmp, "copy_root")
    os.makedirs(root)
    for e in gold:
        write_file(os.path.join(src_root, "ctx", e["path"]), base64.b64decode(e["b64"]), 0o644, MTIME)
    big = os.urandom(3_000_000)                                          # several pieces for the reader threads
    write_file(os.path.join(src_root, "blob/big.bin"), big, 0o600, MTIME)
    ops = [{"src_root": src_root, "srcs": ["ctx"], "dst": "/app/"},
           {"src_root": src_root, "srcs": ["blob/big.bin"], "dst": "/opt/data/big.bin", "uid": 7, "gid": 8},
           {"src_root": src_root, "srcs": ["ctx/simple", "blob"], "dst": "/both/"}]
    with M.MemFS(root, now_sec=MTIME) as fs, M.MemFS(root, now_sec=MTIME) as plain:
        res, raw = commit_to_bytes(fs, tmp, "c0.tar", ops=ops, engine=eng)
        res0, raw0 = commit_to_bytes(plain, tmp, "c0p.tar", ops=ops)
        assert raw == raw0
        mem = {n: d for n, m, d in tar_members(raw) if m.isfile()}
        assert mem["opt/data/big.bin"] == big == mem["both/big.bin"]
        for e in gold:
            assert mem["app/" + e["path"]] == base64.b64decode(e["b64"])
        st = res["stats"]
        n_simple = sum(1 for e in gold if e["path"].startswith("simple/"))
        assert st["n_scanned_files"] == 28 + 1 + n_simple + 1, st          # every op's sources, as often as they are copied
        assert st["n_layer_files"] == st["n_scanned_files"]
        assert st["n_windows"] == 0                                        # (they fit: one batch, staged while the sources are walked)
        # an op that fails where the reference's loop fails: after the ops before it
        bad = ops[:1] + [{"src_root": src_root, "srcs": ["nope"], "dst": "/x/"}]
        try:
            fs.commit_layer(ops=bad, engine=eng)
            raise SystemExit("a missing source did not fail the commit")
        except M.MiError as ex:
            assert "create layer by copy ops: stat src" in str(ex), str(ex)
        res, raw = commit_to_bytes(fs, tmp, "c1.tar", ops=ops[:1], engine=eng)
        assert [e["relpath"] for e in res["layer"]] == ["app"]
    print("OK copy")


def scenario_many(tmp, eng):
    """a few thousand small files in few directories + some large ones, committed twice through one reused batch"""
    root = os.path.join(tmp, "many_root")
    rng = np.random.default_rng(2)
    files = {}
    for d in range(4):
        for k in range(700):
            rel = "m%d/f%04d" % (d, k)
            size = int(rng.integers(0, 9000)) if k % 50 else int(rng.integers(20_000, 900_000))
            data = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
            write_file(os.path.join(root, rel), data, 0o644, MTIME)
            files[rel] = data
    with M.MemFS(root) as fs:
        fs.reserve_device(eng, len(files), sum(map(len, files.values())))        # the arena and the reader threads, ahead of time
        res, raw = commit_to_bytes(fs, tmp, "m0.tar", must_scan=True, engine=eng)
        assert {n: d for n, m, d in tar_members(raw) if m.isfile()} == files
        nonempty = sum(1 for d in files.values() if d)
        assert res["stats"]["files_opened"] == nonempty
        victims = ["m1/f0007", "m2/f0050", "m3/f0699"]
        for rel in victims:
            files[rel] = os.urandom(len(files[rel]) + 3)
            write_file(os.path.join(root, rel), files[rel], 0o644, MTIME + 1)
        res, raw = commit_to_bytes(fs, tmp, "m1.tar", must_scan=True, engine=eng)
        assert {n: d for n, m, d in tar_members(raw) if m.isfile()} == {r: files[r] for r in victims}
    print("OK many")


def scenario_trust(tmp, eng):
    """MI_MEMFS_TRUST_CTIME: a commit does not read again what the kernel says has not changed; it reads what has -- also a
    rewrite that keeps size and mtime (the ctime moves); a file hashed in the same clock tick as its last change is read again"""
    import time
    root = os.path.join(tmp, "trust_root")
    files = make_tree(root, seed=41, mtime=MTIME)
    nonempty = sum(1 for d in files.values() if d)
    time.sleep(0.06)                                                         # (the tree is older than the slack)
    with M.MemFS(root) as fs:
        fs.set_options(trust_ctime=True)
        r = fs.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)
        assert r["stats"]["n_content_trusted"] == 0 and r["stats"]["files_opened"] == nonempty
        r = fs.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)
        st = r["stats"]
        assert r["n_entries"] == 0 and st["n_content_trusted"] == len(files) and st["files_opened"] == 0 and st["n_scanned_files"] == 0, st
        # a rewrite that keeps size and mtime: the inode's ctime moves -- that file is read again, nothing else
        victim = os.path.join(root, "d02/f003.bin")
        sb = os.stat(victim)
        with open(victim, "r+b") as f:
            f.write(os.urandom(sb.st_size))
        os.utime(victim, ns=(sb.st_atime_ns, sb.st_mtime_ns))
        assert os.stat(victim).st_mtime_ns == sb.st_mtime_ns and os.stat(victim).st_ctime_ns != sb.st_ctime_ns
        time.sleep(0.06)
        r = fs.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)
        st = r["stats"]
        assert st["n_content_trusted"] == len(files) - 1 and st["files_opened"] == 1 and st["n_scanned_files"] == 1, st
        # an ordinary edit (new size): read, in the layer, its bytes from the arena of a batch that holds nothing else
        new = os.urandom(70001)
        write_file(os.path.join(root, "d00/f001.bin"), new, 0o644, MTIME + 9)
        time.sleep(0.06)
        res, raw = commit_to_bytes(fs, tmp, "t3.tar", must_scan=True, engine=eng)
        assert [(n, d) for n, m, d in tar_members(raw) if m.isfile()] == [("d00/f001.bin", new)]
        assert res["stats"]["n_scanned_files"] == 1 and res["stats"]["n_content_trusted"] == len(files) - 1
        # without the option everything is read again
        fs.set_options(trust_ctime=False)
        r = fs.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)
        assert r["stats"]["n_content_trusted"] == 0 and r["stats"]["files_opened"] == sum(1 for d in files.values() if d) + (0 if files["d00/f001.bin"] else 1)
    print("OK trust")


def scenario_trust_wide(tmp, eng):
    """MI_MEMFS_TRUST_CTIME on wide directories that change between commits: the predicate keeps, per directory reader, a place
    among the directory's children in the tree and expects the next file to be the next child.  So: files deleted in runs (more
    than the few steps it walks), new names before, between and after the old ones, a directory that was a file, two handles
    committing in turn (each has its own tree) -- every commit reads exactly the files that are new or changed"""
    import time
    rng = np.random.default_rng(77)
    root = os.path.join(tmp, "wide_root")
    files = {}
    for d in ("w0", "w1", "w2/inner"):
        for k in range(0, 400, 2):                                            # even numbers: room between the names
            rel = "%s/f%04d" % (d, k)
            files[rel] = rng.integers(0, 256, int(rng.integers(1, 600)), dtype=np.uint8).tobytes()
            write_file(os.path.join(root, rel), files[rel], 0o644, MTIME)
    time.sleep(0.06)
    with M.MemFS(root) as a, M.MemFS(root) as b:
        for fs in (a, b):
            fs.set_options(trust_ctime=True)
            r = fs.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)
            assert r["stats"]["files_opened"] == len(files) and r["stats"]["n_content_trusted"] == 0
        for step in range(4):
            touched = set()
            live = sorted(files)
            lo = int(rng.integers(0, len(live) - 40))
            deleted = live[lo:lo + 30]
            for rel in deleted:                                               # a run of deletions
                os.unlink(os.path.join(root, rel))
                files.pop(rel)
            for rel in rng.choice(sorted(files), size=3, replace=False):      # a header-only change: the inode's ctime moves with it
                rel = str(rel)
                os.chmod(os.path.join(root, rel), 0o600 + step)
                touched.add(rel)
            for d in ("w0", "w1", "w2/inner"):
                for name in ("a_first_%d" % step, "f%04d" % (2 * int(rng.integers(0, 200)) + 1), "f%04d_x%d" % (2 * int(rng.integers(0, 200)), step),
                             "zz_last_%d" % step):
                    rel = d + "/" + name
                    files[rel] = rng.integers(0, 256, int(rng.integers(1, 600)), dtype=np.uint8).tobytes()
                    write_file(os.path.join(root, rel), files[rel], 0o644, MTIME + step)
                    touched.add(rel)
            rewritten = set()
            for rel in rng.choice(sorted(set(files) - touched), size=7, replace=False):   # rewrites that keep size and mtime
                rel = str(rel)
                rewritten.add(rel)
                pth = os.path.join(root, rel)
                sb = os.stat(pth)
                files[rel] = bytes(x ^ 0x5A for x in files[rel])
                with open(pth, "r+b") as f:
                    f.write(files[rel])
                os.utime(pth, ns=(sb.st_atime_ns, sb.st_mtime_ns))
                touched.add(rel)
            time.sleep(0.06)
            for fs in ((a, b) if step % 2 else (b, a)):
                res, raw = commit_to_bytes(fs, tmp, "w%d.tar" % step, must_scan=True, engine=eng)
                st = res["stats"]
                assert st["files_opened"] == len(touched) and st["n_content_trusted"] == len(files) - len(touched), (step, st, len(touched))
                members = tar_members(raw)
                gone = sorted(os.path.join(os.path.dirname(n), os.path.basename(n)[4:]) for n, m, d in members if os.path.basename(n).startswith(".wh."))
                assert gone == sorted(deleted), (step, gone[:3], deleted[:3])     # every deletion has its whiteout, beside files nobody looked at
                got = {n: d for n, m, d in members if m.isfile() and not os.path.basename(n).startswith(".wh.")}
                in_layer = touched if os.environ.get("MI_TEST_ON_GPU") == "1" else touched - rewritten   # (the double's roots are all
                assert got == {rel: files[rel] for rel in in_layer}, (step, sorted(set(got) ^ in_layer)[:5])  #  alike: read, not told apart)
    print("OK trust_wide")


def scenario_known_tree(tmp, eng):
    """THE ARENA NEVER MOVES (mi_arena.hip).  A handle that knows nothing about its tree, a sequential walk (no enumeration runs
    ahead: the arena learns the tree's size 1 024 files at a time) and 94 MB of files: until round 5 that was three arenas in
    steps -- each one a drain of the reader threads and a copy of what the last one held; now it is ONE address range whose front
    is mapped piece by piece, and nothing of a MiB or more is hipMalloc'ed for it.  The same with the range reserved too small
    (MI_ARENA_RANGE_MB: the pieces are mapped again in a larger range -- they still hold their bytes) and with a mapper that takes
    2 ms per MiB (MI_HIP_STUB_MAP_US: a box that charges fresh device memory by the byte; readers and tar writer wait where they
    touch memory the mapper has not reached).  Every variant: the layer tar holds every file's bytes."""
    import ctypes
    on_gpu = os.environ.get("MI_TEST_ON_GPU") == "1"                      # (no double to ask: mi_commit_stats' arena_* fields alone)
    if on_gpu:
        big_mallocs = vm_ranges = vm_pieces = lambda: 0
    else:
        lib = ctypes.CDLL(None)
        big_mallocs, vm_ranges, vm_pieces = lib.mi_hip_stub_big_mallocs, lib.mi_hip_stub_vm_ranges, lib.mi_hip_stub_vm_pieces
        for f in (big_mallocs, vm_ranges, vm_pieces):
            f.restype = ctypes.c_long
    root = os.path.join(tmp, "known_root")
    rng = np.random.default_rng(5)
    entries, files = [], {}
    n_dirs, per_dir = 24, 60                                              # 94 MB: twelve pieces of 8 MiB (the tests' MI_ARENA_PIECE_MB), three of 32
    for d in range(n_dirs):
        entries.append({"relpath": "k%02d" % d, "kind": M.KIND_DIR, "mode": 0o40755, "mtime_sec": MTIME, "size": 0})
        for k in range(per_dir):
            rel = "k%02d/f%02d" % (d, k)
            data = rng.integers(0, 256, 65_536, dtype=np.uint8).tobytes()
            write_file(os.path.join(root, rel), data, 0o644, MTIME)
            files[rel] = data
            entries.append({"relpath": rel, "kind": M.KIND_FILE, "mode": 0o100644, "mtime_sec": MTIME, "size": len(data)})
    for d in range(n_dirs):
        os.utime(os.path.join(root, "k%02d" % d), (MTIME, MTIME))
    total = sum(map(len, files.values()))
    before = os.environ.get("MI_WALK_THREADS")
    if os.environ.get("MI_TEST_PARALLEL_WALK") == "1":                    # the enumeration runs AHEAD of what is staged: the promise is ahead of
        os.environ["MI_WALK_THREADS"] = "8"                               # the bytes, the mapper works on pieces nobody waits for yet -- and a
    else:                                                                 # range that is outgrown then finds it in the middle of a piece
        os.environ["MI_WALK_THREADS"] = "1"                               # (the sequential walk: no enumeration runs ahead of what is
    try:                                                                  #  staged, the arena learns the tree's size file by file)
        for name in ("fresh", "merged"):
            with M.MemFS(root) as fs:
                if name == "merged":
                    assert fs.update_from_entries(entries) == len(entries)
                n0, r0, p0 = big_mallocs(), vm_ranges(), vm_pieces()
                res, raw = commit_to_bytes(fs, tmp, "k_%s.tar" % name, must_scan=True, engine=eng)
                st = res["stats"]
                assert st["n_scanned_files"] == len(files) and st["files_opened"] == len(files)
                ranges, pieces = vm_ranges() - r0, vm_pieces() - p0
                if os.environ.get("MI_ARENA_RANGE_MB"):
                    assert on_gpu or ranges >= 2, "a range of %s MiB holds 94 MB?" % os.environ["MI_ARENA_RANGE_MB"]      # (outgrown: a larger one)
                    assert st["arena_moves"] >= 1, st
                else:
                    assert on_gpu or ranges == 1, "the arena moved: %d address ranges" % ranges
                    assert st["arena_moves"] == 0, st
                assert on_gpu or pieces == st["arena_pieces"], (pieces, st)
                assert st["arena_pieces"] >= 2 and st["arena_bytes"] >= total, st                                          # mapped in pieces
                if name == "fresh":
                    assert {n: d for n, m, d in tar_members(raw) if m.isfile()} == files
                else:
                    assert res["n_entries"] == 0 and st["n_roots_learned"] == len(files)      # the headers are the merged ones: nothing new
            assert on_gpu or vm_pieces() == p0, "the handle is closed: its arena's pieces are given back"
    finally:
        if before is None:
            del os.environ["MI_WALK_THREADS"]
        else:
            os.environ["MI_WALK_THREADS"] = before
    print("OK known_tree")


def scenario_verify(tmp, eng):
    """END-TO-END BYTE SUMS (mi_filesum.h).  The commit frames the tar from bytes that crossed PCIe twice; every
    file the writer takes out of HBM is held, 1 MiB chunk by chunk, against sums taken where the bytes were READ.  MI_VERIFY_CASE:
        clean      nothing is wrong: every layer file verified, nothing fetched twice; the reference's tar
        readback1  MI_STAGE_FAULT=readback:N -- ONE copy into a read-back window arrives with a flipped byte: the chunk is fetched
                   again and is right; the commit succeeds, n_refetched = 1, the tar is the reference's
        readback3  readback:N:3 -- the second fetch is wrong too: MI_ERR_IO naming the file, the arena range and the hop
                   "HBM -> pinned read-back window" (a third, plain copy finds the arena right)
        copy       MI_STAGE_FAULT=copy:N -- a staged span loses 4 KiB in HBM right after its copy: no second fetch can help;
                   MI_ERR_IO naming the hop "pinned slab -> HBM"
    A tree of files whose sizes are multiples of 256 (no alignment gaps: a flipped byte always lies in some file) that take both
    ways into the arena -- directory blocks and paths, several chunks, sizes that are not multiples of 8 do not exist here but do
    in every other scenario, which all run with the check on."""
    case = os.environ.get("MI_VERIFY_CASE", "clean")
    root = os.path.join(tmp, "verify_root")
    rng = np.random.default_rng(77)
    files = {}
    for d in range(3):
        for k, size in enumerate((256, 4096, 12_288, 3 << 20, 1_048_576 + 512, 700_160)):
            rel = "v%d/f%d.bin" % (d, k)
            data = rng.integers(1, 256, size, dtype=np.uint8).tobytes()       # (no zero bytes: a range that reads as zeros differs)
            write_file(os.path.join(root, rel), data, 0o644, MTIME)
            files[rel] = data
    for dp, dns, fns in os.walk(root):
        os.utime(dp, (MTIME, MTIME))
    with M.MemFS(root) as plain:
        _, want = commit_to_bytes(plain, tmp, "vp.tar", must_scan=True)
    with M.MemFS(root) as fs:
        if case in ("clean", "readback1"):
            res, raw = commit_to_bytes(fs, tmp, "v.tar", must_scan=True, engine=eng)
            st = res["stats"]
            assert raw == want, "the tar is not the reference's"
            assert st["n_verified_files"] == len(files) == st["n_layer_files"], st
            assert st["verified_bytes"] == sum(map(len, files.values())), st
            assert st["n_refetched"] == (1 if case == "readback1" else 0), st
        else:
            try:
                commit_to_bytes(fs, tmp, "v.tar", must_scan=True, engine=eng)
            except M.MiError as e:
                msg = str(e)
                assert "MI_ERR_IO" in msg and "copy file v" in msg and "arena [" in msg and "sums where the bytes were read" in msg, msg
                hop = "HBM -> pinned read-back window" if case == "readback3" else "pinned slab -> HBM"
                assert hop in msg, msg
            else:
                raise AssertionError("a commit whose bytes changed on the way produced a layer")
    print("OK verify " + case)


def scenario_many_gpus(tmp, eng):
    """THE COMMIT OVER SEVERAL CTXS (mi_memfs_commit_layer_n) -- MI_TEST_N_CTXS of them (2 and 8 in the tests), all
    on the one device there is (the double's, or the box's).  Against the one-ctx commit and the header-only commit of the same
    tree: the SAME tar, byte for byte; every file opened and read once (/proc/self/io agrees); every root = the one-ctx commit's (on
    the GPU: = the oracle's); the bytes spread over the ctxs within a factor of two; then the life of a build -- nothing changed
    (empty layer), a same-size rewrite within the second (on the GPU: caught by its root), a deleted directory (whiteout), a COPY
    op; a chunk index on ctx 0 fed by all ctxs (as many digests as the one-ctx commit's index holds); a corrupted read-back on
    whichever ctx serves the second window copy is repaired (MI_STAGE_FAULT=readback:1)."""
    n = int(os.environ.get("MI_TEST_N_CTXS", "2"))
    on_gpu = os.environ.get("MI_TEST_ON_GPU") == "1"
    root = os.path.join(tmp, "many_root")
    files = make_tree(root, seed=41, n_dirs=10, files_per_dir=12, big_every=3, mtime=MTIME)
    split_mib = int(os.environ.get("MI_COMMIT_SPLIT_MIB", "256"))
    n_split = 0
    if split_mib <= 8:                                                      # files that are SPLIT over the ctxs as parts (GPU only: the
        rng = np.random.default_rng(43)                                     # parts agree on their boundary cuts through the kernels' records)
        for name, size in (("big/a.bin", 5 << 20), ("big/b.bin", (9 << 20) + 13), ("big/c.bin", split_mib << 20), ("big/d.bin", (split_mib << 20) - 1),
                           ("big/zeros.bin", 6 << 20), ("big/shifted_zeros.bin", (7 << 20) + 5)):
            # (shifted_zeros: 30 001 random bytes, then zeros -- the forced cuts lie on a grid that begins at the prefix's last content
            #  cut, so every part's assumed entry is wrong and the owners need a round per boundary to agree: group_resolve_parts' loop)
            data = bytes(size) if "zeros" in name else rng.integers(0, 256, size, dtype=np.uint8).tobytes()
            if "shifted" in name:
                data = rng.integers(0, 256, 30001, dtype=np.uint8).tobytes() + data[30001:]
            write_file(os.path.join(root, name), data, 0o644, MTIME)
            files[name] = data
            n_split += 1 if size >= (split_mib << 20) and size >= (2 << 20) else 0
        os.utime(os.path.join(root, "big"), (MTIME, MTIME))
        os.utime(root, (MTIME, MTIME))
    total = sum(map(len, files.values()))
    nonempty = [d for d in files.values() if d]
    engines = [eng] + [M.Engine(n_streams=2, staging_bytes=1 << 20) for _ in range(n - 1)]
    try:
        with M.MemFS(root) as many, M.MemFS(root) as one, M.MemFS(root) as plain:
            idx_many, idx_one = M.ChunkIndex(engines[-1]), M.ChunkIndex(eng)          # (the index of the group on its LAST ctx: every
                                                                                      #  other ctx's digests reach it through the host)
            many.set_index(idx_many)
            one.set_index(idx_one)
            r0, _ = proc_io()
            with M.MemFS(root) as probe:                                    # (digests only: nothing of the tar is read back here)
                probe.commit_layer(must_scan=True, engine=engines, gzip_level=M.GZIP_OFF)
            r1, _ = proc_io()
            windowed = os.environ.get("MI_COMMIT_FORCE_WINDOWS") == "1"     # (roots in windows, the tar's files from disk: two reads)
            halos = n_split * n * (256 << 10)                               # (a part re-reads 256 KiB in front of its range: its halo)
            assert windowed or total <= r1 - r0 <= total + halos + (512 << 10), (r1 - r0, total)  # one read per file, as the kernel counts it
            res, raw = commit_to_bytes(many, tmp, "m0.tar", must_scan=True, engine=engines)
            res1, raw1 = commit_to_bytes(one, tmp, "m0_one.tar", must_scan=True, engine=eng)
            res0, raw0 = commit_to_bytes(plain, tmp, "m0_plain.tar", must_scan=True)
            assert raw == raw1 == raw0, "the tar over %d ctxs is not the one-ctx commit's / the reference's" % n
            st = res["stats"]
            assert st["n_ctxs"] == n and st["n_scanned_files"] == len(files) and st["scanned_bytes"] == total, st
            if windowed:
                assert st["n_windows"] >= 2 and st["n_verified_files"] == 0, st
            else:
                assert st["n_split_files"] == n_split, st
                if n_split:
                    assert len(nonempty) < st["files_opened"] <= len(nonempty) + n_split * n and total < st["file_bytes_read"] <= total + halos, st
                else:
                    assert (st["files_opened"], st["file_bytes_read"]) == (len(nonempty), total), st
                assert st["n_verified_files"] == len(files), st
                # (MI_STAGE_FAULT=readback:k flips a byte in the k-th window copy of EVERY ctx: each member may fetch a chunk twice)
                assert st["n_refetched"] <= (n if os.environ.get("MI_STAGE_FAULT") else 0), st
                assert st["ctx_bytes_min"] > 0 and st["ctx_bytes_max"] <= 2 * st["ctx_bytes_min"] + (1 << 20), st     # spread by bytes
            assert st["n_chunks"] == res1["stats"]["n_chunks"] or not on_gpu
            by, by1 = {e["relpath"]: e for e in res["layer"]}, {e["relpath"]: e for e in res1["layer"]}
            assert list(by) == list(by1)
            if on_gpu:
                from oracle import mi_oracle as O
                O.build()
                from commit_cases import oracle_root
                for rel, data in files.items():
                    assert by[rel]["root"] == by1[rel]["root"] == oracle_root(O, data), rel
                    assert many.root_of("/" + rel) == oracle_root(O, data), rel
                assert len(idx_many) == len(idx_one) and st["n_index_new"] == res1["stats"]["n_index_new"]
                assert st["index_new_bytes"] == res1["stats"]["index_new_bytes"]
            # nothing changed: the empty layer; the handle kept its batches
            res, raw = commit_to_bytes(many, tmp, "m1.tar", must_scan=True, engine=engines)
            assert res["n_entries"] == 0 and raw == bytes(1024) and res["stats"]["n_ctxs"] == n
            # a same-size rewrite within the second + an ordinary edit + a deleted directory
            rel_a, rel_b = "d02/f001.bin", "d04/f005.bin"
            new_a = bytearray(files[rel_a]); new_a[len(new_a) // 2] ^= 0x55
            write_file(os.path.join(root, rel_a), bytes(new_a), 0o644, MTIME)
            new_b = os.urandom(len(files[rel_b]) + 301)
            write_file(os.path.join(root, rel_b), new_b, 0o644, MTIME + 9)
            import shutil
            shutil.rmtree(os.path.join(root, "d06"))
            for dp in (root, os.path.join(root, "d02"), os.path.join(root, "d04")):
                os.utime(dp, (MTIME, MTIME))
            res, raw = commit_to_bytes(many, tmp, "m2.tar", must_scan=True, engine=engines)
            got = {nm: d for nm, m_, d in tar_members(raw) if m_.isfile()}
            assert got.get(rel_b) == new_b and ".wh.d06" in [e["relpath"] for e in res["layer"]], sorted(got)
            if on_gpu:
                assert got.get(rel_a) == bytes(new_a) and res["stats"]["n_content_changed"] == 1, sorted(got)
            # a COPY op over the same ctxs
            src_root = os.path.join(tmp, "many_src")
            write_file(os.path.join(src_root, "pkg/a.bin"), os.urandom(70_000), 0o644, MTIME)
            write_file(os.path.join(src_root, "pkg/b.bin"), os.urandom(3_000), 0o644, MTIME)
            ops = [{"src_root": src_root, "srcs": ["pkg"], "dst": "/opt/pkg/", "uid": 0, "gid": 0}]
            res, raw = commit_to_bytes(many, tmp, "m3.tar", ops=ops, engine=engines)
            res0, raw0 = commit_to_bytes(plain, tmp, "m3_plain_pre.tar", must_scan=True)      # (the plain handle catches up with the edits first)
            res0, raw0 = commit_to_bytes(plain, tmp, "m3_plain.tar", ops=ops)
            assert sorted(nm for nm, m_, d in tar_members(raw) if m_.isfile()) == ["opt/pkg/a.bin", "opt/pkg/b.bin"] and raw == raw0
            # back to ONE ctx on the same handle: the group is given back, a single batch takes over
            res, raw = commit_to_bytes(many, tmp, "m4.tar", must_scan=True, engine=eng)
            assert res["n_entries"] == 0 and res["stats"]["n_ctxs"] == 1
    finally:
        for e in engines[1:]:
            e.close()
    print("OK many_gpus %d" % n)


def scenario_mapper_fails(tmp, eng):
    """the arena's mapper cannot get its k-th piece (MI_HIP_STUB_VM_FAIL: the device runs out under it, or the runtime refuses a map /
    an access change): the commit FAILS -- an error in the reference's chain that says what happened, no hang (reader threads, scan
    thread and tar writer all wait for the mapper somewhere), nothing leaked; the handle and the ctx are usable afterwards"""
    import ctypes
    lib = ctypes.CDLL(None)
    vm_pieces = lib.mi_hip_stub_vm_pieces
    vm_pieces.restype = ctypes.c_long
    root = os.path.join(tmp, "mapper_root")
    rng = np.random.default_rng(3)
    for d in range(6):
        for k in range(10):
            write_file(os.path.join(root, "m%d/f%d.bin" % (d, k)), rng.integers(0, 256, 1_500_000, dtype=np.uint8).tobytes(), 0o644, MTIME)
    for dp, dns, fns in os.walk(root):
        os.utime(dp, (MTIME, MTIME))
    p0 = vm_pieces()
    if os.environ["MI_HIP_STUB_VM_FAIL"].startswith("reserve:"):
        # no address range to be had (ranges are never given back: a process can run out of them): the batch takes ONE allocation
        # that moves when it grows, as every batch did until round 5 -- the commit succeeds, nothing was mapped piecewise
        with M.MemFS(root) as fs, M.MemFS(root) as plain:
            res, raw = commit_to_bytes(fs, tmp, "mf.tar", must_scan=True, engine=eng)
            res0, raw0 = commit_to_bytes(plain, tmp, "mf_plain.tar", must_scan=True)
            st = res["stats"]
            assert raw == raw0 and res["n_entries"] == 66 and st["n_verified_files"] == 60 and st["n_refetched"] == 0, st
            assert st["arena_pieces"] == 1 and st["arena_bytes"] >= 60 * 1_500_000 and vm_pieces() == p0, st      # (one allocation counts as one piece)
            res, raw = commit_to_bytes(fs, tmp, "mf1.tar", must_scan=True, engine=eng)        # the handle goes on with that arena
            assert res["n_entries"] == 0 and raw == bytes(1024)
        print("OK mapper_fails")
        return
    with M.MemFS(root) as fs:
        try:
            commit_to_bytes(fs, tmp, "mf.tar", must_scan=True, engine=eng)
        except M.MiError as e:
            msg = str(e)
            assert "failed to generate diff layer" in msg and "the arena's next" in msg, msg
            what = os.environ["MI_HIP_STUB_VM_FAIL"].split(":")[0]
            assert ("hipMemCreate" in msg) == (what == "create") and ("hipMemMap / hipMemSetAccess" in msg) == (what != "create"), msg
            assert ("MI_ERR_NOMEM" in msg or "MI_ERR_IO" in msg or "MI_ERR_HIP" in msg), msg
        else:
            raise AssertionError("a commit whose arena could not be mapped produced a layer")
    assert vm_pieces() == p0, "pieces leaked: %d" % (vm_pieces() - p0)
    with M.MemFS(root) as plain:                                              # the process goes on: the reference's commit of the same tree
        res, raw = commit_to_bytes(plain, tmp, "mf_plain.tar", must_scan=True)
        assert res["n_entries"] == 66
    print("OK mapper_fails")


def scenario_slash(tmp, eng):
    """the root of every real build is "/": the same commit with the handle rooted there (a node's source IS its path, nothing is
    trimmed), everything but one directory of this test's blacklisted -- with a ctx, with MI_MEMFS_TRUST_CTIME, and without"""
    import shutil
    import time
    if os.geteuid() != 0:
        print("OK slash")
        return
    import fcntl
    top = "/mi_slash_%d" % os.getpid()
    lock = open("/tmp/mi_slash.lock", "w")                                   # (one at a time: a sibling's directory appearing at "/"
    fcntl.flock(lock, fcntl.LOCK_EX)                                         #  between two commits would be part of the second)
    try:
        files = make_tree(top, seed=51, n_dirs=3, mtime=MTIME)
        blacklist = ["/" + n for n in os.listdir("/") if "/" + n != top]
        time.sleep(0.06)
        with M.MemFS("/", blacklist=blacklist) as fs, M.MemFS("/", blacklist=blacklist) as plain:
            fs.set_options(trust_ctime=True)
            res, raw = commit_to_bytes(fs, tmp, "r0.tar", must_scan=True, engine=eng)
            res0, raw0 = commit_to_bytes(plain, tmp, "r0p.tar", must_scan=True)
            assert raw == raw0
            got = {n: d for n, m, d in tar_members(raw) if m.isfile()}
            assert got == {top[1:] + "/" + rel: data for rel, data in files.items()}
            assert all(e["src"] == "/" + e["relpath"] for e in res["layer"])
            assert fs.root_of(top + "/d00/f001.bin") is not None
            r = fs.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)
            assert r["n_entries"] == 0 and r["stats"]["n_content_trusted"] == len(files) and r["stats"]["files_opened"] == 0, r["stats"]
            os.unlink(top + "/d00/f001.bin")
            res, raw = commit_to_bytes(fs, tmp, "r1.tar", must_scan=True, engine=eng)
            assert [e["relpath"] for e in res["layer"]] == [top[1:], top[1:] + "/d00", top[1:] + "/d00/.wh.f001.bin"]
    finally:
        shutil.rmtree(top, ignore_errors=True)
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()
    print("OK slash")


def scenario_oversize(tmp, eng):
    """a tree larger than the device (the double refuses large allocations: MI_HIP_STUB_MALLOC_LIMIT_MB): the commit does not
    fail -- the roots come window by window, the writer reads the layer's files from disk; the tar is the reference's"""
    root = os.path.join(tmp, "over_root")
    rng = np.random.default_rng(9)
    files = {}
    for d in range(6):
        for k in range(12):
            rel = "o%d/f%02d" % (d, k)
            size = int(rng.integers(0, 30_000)) if k % 4 else int(rng.integers(300_000, 1_200_000))
            data = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
            write_file(os.path.join(root, rel), data, 0o644, MTIME)
            files[rel] = data
    total = sum(map(len, files.values()))
    assert total > 9 << 20
    with M.MemFS(root) as fs, M.MemFS(root) as plain:
        if os.environ.get("MI_TEST_TRUST"):
            fs.set_options(trust_ctime=True)
        res, raw = commit_to_bytes(fs, tmp, "o0.tar", must_scan=True, engine=eng)
        st = res["stats"]
        assert st["n_windows"] >= 4 and st["pipelined"] == 0, st
        assert st["n_scanned_files"] == len(files) and st["scanned_bytes"] == total
        assert {n: d for n, m, d in tar_members(raw) if m.isfile()} == files
        _, raw0 = commit_to_bytes(plain, tmp, "o0p.tar", must_scan=True)
        assert raw == raw0
        assert 2 * total <= st["file_bytes_read"] <= 3 * total            # once for the roots, once for the tar (the price) -- and
                                                                          # what the first attempt had read when it ran out of room
        assert fs.root_of("/o0/f01") is not None
        res, raw = commit_to_bytes(fs, tmp, "o1.tar", must_scan=True, engine=eng)
        assert res["n_entries"] == 0
        if not os.environ.get("MI_TEST_TRUST"):
            assert res["stats"]["n_windows"] >= 4
            assert res["stats"]["file_bytes_read"] == total               # (no second attempt at one batch: straight to the windows)
        else:
            # (right after the first commit: the files written within the racy window of its start are read again -- how many that
            #  is, and whether they fit the device in one batch, depends on how long the tree took to write)
            assert res["stats"]["file_bytes_read"] <= 2 * total
            import time
            time.sleep(0.06)                                              # a window's files were hashed files: their inodes are on record --
            r3 = fs.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)   # the next commit trusts them and fits
            assert r3["n_entries"] == 0 and r3["stats"]["n_windows"] == 0 and r3["stats"]["files_opened"] == 0, r3["stats"]
        write_file(os.path.join(root, "o3/f05"), b"new" * 1000, 0o644, MTIME + 3)
        res, raw = commit_to_bytes(fs, tmp, "o2.tar", must_scan=True, engine=eng)
        assert [(n, d) for n, m, d in tar_members(raw) if m.isfile()] == [("o3/f05", b"new" * 1000)]
    print("OK oversize")


def scenario_oversize_copy(tmp, eng):
    """COPY sources larger than the device (three ops, one source copied twice): planned again without a batch, the roots window by
    window by the files' ordinals across the ops' walks, the tar's files from disk; the tar is the reference's, the tree holds
    the roots (a later COPY of the same bytes with another mtime-less header is told by them: same second, other bytes)"""
    src_root, root = os.path.join(tmp, "over_ctx"), os.path.join(tmp, "over_copy_root")
    os.makedirs(root)
    rng = np.random.default_rng(11)
    files = {}
    for d in ("a", "b"):
        for k in range(14):
            rel = "%s/f%02d" % (d, k)
            size = int(rng.integers(0, 30_000)) if k % 3 else int(rng.integers(500_000, 1_300_000))
            data = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
            write_file(os.path.join(src_root, rel), data, 0o644, MTIME)
            files[rel] = data
    total = sum(map(len, files.values()))
    assert total > 7 << 20
    ops = [{"src_root": src_root, "srcs": ["a"], "dst": "/one/"},
           {"src_root": src_root, "srcs": ["b", "a/f00"], "dst": "/two/", "uid": 3, "gid": 4},
           {"src_root": src_root, "srcs": ["a/f03"], "dst": "/three/file"}]
    with M.MemFS(root, now_sec=MTIME) as fs, M.MemFS(root, now_sec=MTIME) as plain:
        res, raw = commit_to_bytes(fs, tmp, "oc0.tar", ops=ops, engine=eng)
        st = res["stats"]
        assert st["n_windows"] >= 3 and st["pipelined"] == 0, st
        assert st["n_scanned_files"] == 14 + 14 + 1 + 1, st
        _, raw0 = commit_to_bytes(plain, tmp, "oc0p.tar", ops=ops)
        assert raw == raw0
        mem = {n: d for n, m, d in tar_members(raw) if m.isfile()}
        for k in range(14):
            assert mem["one/f%02d" % k] == files["a/f%02d" % k] and mem["two/f%02d" % k if k else "two/f00"] in (files["b/f%02d" % k], files["a/f00"])
        assert mem["three/file"] == files["a/f03"]
        on_gpu = os.environ.get("MI_TEST_ON_GPU") == "1"                   # (the double launches no kernels: its roots are not roots)
        if on_gpu:
            from oracle import mi_oracle as O
            O.build()
            from commit_cases import oracle_root
        for path, rel in (("/one/f05", "a/f05"), ("/two/f07", "b/f07"), ("/three/file", "a/f03")):
            got = fs.root_of(path)
            assert got is not None and (not on_gpu or got == oracle_root(O, files[rel])), path
        if on_gpu:
            # the same second, the same size, other bytes: only the roots can tell
            changed = bytearray(files["a/f06"]); changed[len(changed) // 2] ^= 0x40
            write_file(os.path.join(src_root, "a/f06"), bytes(changed), 0o644, MTIME)
            res, raw = commit_to_bytes(fs, tmp, "oc1.tar", ops=ops[:1], engine=eng)
            assert [(n, d) for n, m, d in tar_members(raw) if m.isfile()] == [("one/f06", bytes(changed))], [n for n, m, d in tar_members(raw)]
            assert res["stats"]["n_content_changed"] == 1 and res["stats"]["n_windows"] >= 2
    print("OK oversize_copy")


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "known_tree":
        with M.Engine(n_streams=int(sys.argv[2]), staging_bytes=1 << 20) as eng:
            scenario_known_tree(sys.argv[1], eng)
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "mapper_fails":
        with M.Engine(n_streams=int(sys.argv[2]), staging_bytes=1 << 20) as eng:
            scenario_mapper_fails(sys.argv[1], eng)
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "many_gpus":
        with M.Engine(n_streams=int(sys.argv[2]), staging_bytes=1 << 20) as eng:
            scenario_many_gpus(sys.argv[1], eng)
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "verify":
        with M.Engine(n_streams=int(sys.argv[2]), staging_bytes=1 << 20) as eng:
            scenario_verify(sys.argv[1], eng)
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "trust_wide":
        with M.Engine(n_streams=int(sys.argv[2]), staging_bytes=1 << 20) as eng:
            scenario_trust_wide(sys.argv[1], eng)
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "oversize_copy":
        with M.Engine(n_streams=int(sys.argv[2]), staging_bytes=1 << 20) as eng:
            scenario_oversize_copy(sys.argv[1], eng)
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "oversize":
        with M.Engine(n_streams=int(sys.argv[2]), staging_bytes=1 << 20) as eng:
            scenario_oversize(sys.argv[1], eng)
        sys.exit(0)
    tmp, threads = sys.argv[1], int(sys.argv[2])
    with M.Engine(n_streams=threads, staging_bytes=1 << 20) as eng:
        scenario_scan(tmp, eng)
        scenario_copy(tmp, eng)
        scenario_many(tmp, eng)
        scenario_trust(tmp, eng)
        scenario_trust_wide(tmp, eng)
        scenario_slash(tmp, eng)
