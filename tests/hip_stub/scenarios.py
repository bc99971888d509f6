"""Scenarios for tests/test_host_hip_double.py: the library's host-fed path on the HIP test double (mi_hip_stub.cpp).
Run with LD_PRELOAD=<the double>; kernels do not run there, so what is checked is what the HOST side is responsible for:
every byte of every file, buffer and range lands in the batch's arena where the file table says, whatever the order of
adds, the number of reader threads, the slab size, arena growth under way, batch reuse, two batches alternating -- and
errors stay with the batch they belong to.  Prints one "OK <name>" line per scenario."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import makisu_amd as M  # noqa: E402


def make_files(d, sizes, seed):
    rng = np.random.default_rng(seed)
    os.makedirs(d, exist_ok=True)
    out = []
    for i, sz in enumerate(sizes):
        p = os.path.join(d, "f%05d" % i)
        data = rng.integers(0, 256, sz, dtype=np.uint8).tobytes()
        with open(p, "wb") as f:
            f.write(data)
        out.append((p, data))
    return out


def check(b, want, name):
    got = bytes(b.read_back())
    if got != want:
        n = min(len(got), len(want))
        first = next((i for i in range(n) if got[i] != want[i]), n)
        raise SystemExit("%s: arena differs from the files at byte %d of %d (got %r, want %r)" %
                         (name, first, len(want), got[first:first + 8], want[first:first + 8]))


def scenario_mix(tmp, threads, slab):
    """big and small files through every door, in one batch"""
    rng = np.random.default_rng(7)
    sizes = [0, 1, 7, 100, 4095, 4096, 4097, 70000, slab - 1, slab, slab + 1, 3 * slab + 5, 1 << 20] + \
        [int(x) for x in rng.integers(0, 20000, 300)]
    files = make_files(os.path.join(tmp, "mix"), sizes, 11)
    with M.Engine(n_streams=threads, staging_bytes=slab) as eng, eng.batch() as b:
        want = b""
        for k, (p, data) in enumerate(files[:100]):
            if k % 3 == 0:
                b.add_path(p)
            elif k % 3 == 1:
                b.add_bytes(data)                       # small ones: the inline window; large: reader threads
            else:
                b.add_path_range(p, 0, len(data))
            want += data
        b.add_paths([p for p, _ in files[100:250]])     # opened by the reader threads
        want += b"".join(d for _, d in files[100:250])
        p, data = files[12]                             # a range out of the middle of a file
        b.add_path_range(p, 1000, 50000)
        want += data[1000:51000]
        big = rng.integers(0, 256, 5 * slab + 17, dtype=np.uint8).tobytes()
        b.add_bytes(big)
        want += big
        n = b.add_tree(os.path.join(tmp, "mix"))         # the whole directory once more, in walk order
        want += b"".join(d for _, d in sorted(files))
        assert n == len(files) + 1
        b.run()
        assert b.counts()[0] == 100 + 150 + 2 + len(files) and b.counts()[2] == len(want)
        check(b, want, "mix")


def scenario_growth_and_reuse(tmp, threads, slab):
    """tiny hints: the arena grows again and again while readers are busy; then the batch is reset and reused"""
    files = make_files(os.path.join(tmp, "grow"), [3 * slab + 1] * 6 + [5000] * 200 + [slab * 4] * 3, 13)
    with M.Engine(n_streams=threads, staging_bytes=slab) as eng, eng.batch(1, 1) as b:
        for rnd in range(3):
            want = b""
            order = files if rnd % 2 == 0 else files[::-1]
            for k, (p, data) in enumerate(order):
                if k % 2:
                    b.add_path(p)
                else:
                    b.add_bytes(data)
                want += data
            b.run()
            check(b, want, "growth round %d" % rnd)
            b.reset()


def scenario_two_batches(tmp, threads, slab):
    """two batches of one ctx alternate: one is submitted while the other is being fed by the same reader threads"""
    files = make_files(os.path.join(tmp, "two"), [2 * slab + 3, 100, slab, 9000, 12] * 20, 17)
    with M.Engine(n_streams=threads, staging_bytes=slab) as eng, eng.batch() as b0, eng.batch() as b1:
        bs = [b0, b1]
        wants = [b"", b""]
        for step in range(6):
            cur, other = bs[step % 2], bs[(step + 1) % 2]
            if step >= 2:
                cur.wait()
                check(cur, wants[step % 2], "two batches, step %d" % step)
                cur.reset()
            sel = files[step * 7 % 50:][:40]
            cur.add_paths([p for p, _ in sel[:20]])
            for p, data in sel[20:]:
                cur.add_bytes(data)
            wants[step % 2] = b"".join(d for _, d in sel)
            cur.submit()
            del other
        for k in (0, 1):
            bs[k].wait()
            check(bs[k], wants[k], "two batches, end %d" % k)


def scenario_interleaved(tmp, threads, slab):
    """two batches of one ctx fed file by file in turn, files of one size: the readers' queue holds A's and B's pieces
    alternately at the SAME arena offsets -- a run of pieces that travels as one copy must stay inside one batch"""
    fa = make_files(os.path.join(tmp, "ila"), [4096] * 150, 31)
    fb = make_files(os.path.join(tmp, "ilb"), [4096] * 150, 37)
    with M.Engine(n_streams=threads, staging_bytes=slab) as eng, eng.batch(300, 2 << 20) as a, eng.batch(300, 2 << 20) as b:
        for (pa, _), (pb, _) in zip(fa, fb):
            a.add_path(pa)
            b.add_path(pb)
        a.run()
        b.run()
        check(a, b"".join(d for _, d in fa), "interleaved fills, batch a")
        check(b, b"".join(d for _, d in fb), "interleaved fills, batch b")


def scenario_errors(tmp, threads, slab):
    """a file that is gone, a file that shrank: the batch that holds it fails -- and keeps failing -- the other does not"""
    files = make_files(os.path.join(tmp, "err"), [slab + 10, 3000, 40000], 23)
    with M.Engine(n_streams=threads, staging_bytes=slab) as eng, eng.batch() as bad, eng.batch() as good:
        gone = os.path.join(tmp, "err", "gone")
        bad.add_paths([files[0][0], gone, files[2][0]], sizes=[slab + 10, 10000, 40000])   # deferred open: the reader
                                                                                            # thread finds it missing
        good.add_path(files[2][0])
        good.add_bytes(files[0][1])
        for _ in range(2):                              # sticky
            try:
                bad.run()
            except M.MiError as e:
                assert e.code == -5 and "gone" in str(e), str(e)
            else:
                raise SystemExit("a batch with a missing file ran")
        good.run()
        check(good, files[2][1] + files[0][1], "the other batch")
        bad.reset()                                     # the error goes with the contents
        bad.add_path(files[1][0])
        bad.run()
        check(bad, files[1][1], "the failed batch after reset")
        short = os.path.join(tmp, "err", "short")
        open(short, "wb").write(b"y" * 100)
        bad.reset()
        bad.add_paths([short], sizes=[5000])            # shorter than the size given
        try:
            bad.run()
        except M.MiError as e:
            assert "shorter than the size given" in str(e), str(e)
        else:
            raise SystemExit("a short file was staged")


def scenario_api(tmp, threads, slab):
    """the order of calls: what comes too early, twice or too late is MI_ERR_STATE and changes nothing"""
    def code(fn):
        try:
            fn()
        except M.MiError as e:
            return e.code
        return 0
    with M.Engine(n_streams=threads, staging_bytes=slab) as eng, eng.batch() as b:
        assert [code(b.files), code(b.chunks), code(b.read_back), code(b.wait)] == [-6, -6, -6, -6]
        data = os.urandom(3 * slab + 5)
        b.add_bytes(data)
        b.add_bytes(b"")
        assert code(b.submit) == 0 and code(b.submit) == -6 and code(lambda: b.add_bytes(b"x")) == -6
        assert code(b.wait) == 0 and code(b.wait) == -6 and code(b.run) == -6 and code(b.rerun) == 0
        assert code(lambda: b.add_bytes(b"x")) == -6
        assert len(b.files()) == 2 and b.counts()[2] == len(data)
        check(b, data, "api")
        assert code(lambda: b.context_checksum(b"", [("a", None, 0)])) == -6       # needs MI_FLAG_FILE_CRC32
        b.reset()
        assert code(lambda: b.add_tree(os.path.join(tmp, "nowhere"))) == -5
        assert code(lambda: b.add_path(os.path.join(tmp, "nowhere"), size=10)) == -5
        assert code(lambda: b.add_path_range(__file__, 10, 1 << 40)) == -5           # shorter than the range given
        b.add_bytes(b"after the errors")
        b.run()
        check(b, b"after the errors", "api, after errors")


def scenario_warm(tmp, threads, slab):
    """mi_ctx_warm: all reader threads up and one synthetic batch through the pipeline, before, between and after the ctx's own
    batches -- their bytes land where they landed without it, and the ctx's statistics are about ITS batches"""
    with M.Engine(n_streams=threads, staging_bytes=slab) as eng:
        eng.warm()
        eng.warm()
        data = os.urandom(5 * slab + 3)
        with eng.batch() as b:
            b.add_bytes(data)
            b.run()
            check(b, data, "warm, first batch")
            before = eng.stats()
            eng.warm()                                          # with a batch of the caller's alive
            assert eng.stats() == before
            b.reset()
            b.add_bytes(data[:slab + 1])
            b.run()
            check(b, data[:slab + 1], "warm, after the warm call")


def scenario_tree_reserves_ahead(tmp, threads, slab):
    """mi_batch_add_tree without size hints: the enumeration runs ahead of the files handed over and the arena is sized
    for what it has seen -- a handful of (re)allocations for a tree of 6 000 files, not one per 1.5x step"""
    import ctypes
    rng = np.random.default_rng(5)
    d = os.path.join(tmp, "tree")
    want = {}
    for k in range(30):
        os.makedirs(os.path.join(d, "d%02d" % k), exist_ok=True)
        for f in range(200):
            data = rng.integers(0, 256, 6000, dtype=np.uint8).tobytes()
            p = os.path.join(d, "d%02d" % k, "f%03d" % f)
            with open(p, "wb") as fh:
                fh.write(data)
            want[p] = data
    big = ctypes.CDLL(None).mi_hip_stub_big_mallocs
    big.restype = ctypes.c_long
    with M.Engine(n_streams=threads, staging_bytes=slab) as eng, eng.batch() as b:
        before = big()
        b.add_tree(d)
        b.run()
        grown = big() - before
        check(b, b"".join(want[p] for p in sorted(want)), "tree")
        assert grown <= 4, "the 36 MB arena was allocated %d times" % grown
        print("   arena allocations for 6 000 files without hints:", grown, flush=True)
    # a mixed tree: small files are read where the walk lists them (one block per directory), larger ones go to the reader
    # threads as paths, empty files are rows without bytes; nested directories, a symlink, names that sort between them --
    # the file table keeps filepath.Walk's order whatever way the bytes took
    d2 = os.path.join(tmp, "mixed")
    want2 = {}
    sizes = [0, 1, 255, 256, 257, 4096, 16384, 16385, 32768, 70000, 300000, 1 << 20, 5, 0, 33]
    for k, rel in enumerate(["a/x", "a/y/z", "a/y", "b", ".", "a/y/z/deep/er", "c c", "a.b"]):
        os.makedirs(os.path.join(d2, rel), exist_ok=True)
        for j, n in enumerate(sizes[k:] + sizes[:k]):
            data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            p = os.path.join(d2, rel, "f%02d-%d" % (j, n)) if rel != "." else os.path.join(d2, "f%02d-%d" % (j, n))
            with open(p, "wb") as fh:
                fh.write(data)
            want2[os.path.normpath(p)] = data
    os.symlink("f00-0", os.path.join(d2, "a", "link"))

    def walk_order(root):                                   # filepath.Walk: lexical, a directory before its contents
        out = []
        for name in sorted(os.listdir(root)):
            p = os.path.join(root, name)
            if os.path.islink(p):
                continue
            if os.path.isdir(p):
                out += walk_order(p)
            else:
                out.append(p)
        return out
    with M.Engine(n_streams=threads, staging_bytes=slab) as eng, eng.batch() as b:
        n_entries = b.add_tree(d2)
        b.run()
        order = walk_order(d2)
        assert len(order) == len(want2) == len(b.files())
        check(b, b"".join(want2[os.path.normpath(p)] for p in order), "tree (mixed)")
        ents = b.tree_entries(n_entries)
        files = [e for e in ents if e[4] == M.KIND_FILE]
        assert [e[2] for e in files] == list(range(len(files))), "file indices follow the walk"
        assert [e[3] for e in files] == [len(want2[os.path.normpath(p)]) for p in order]


def scenario_two_ctxs(tmp, threads, slab):
    """"multiple ctxs may run concurrently" (include/makisu_mi.h): two engines driven from two host threads at once"""
    import threading
    errs = []

    def work(k):
        try:
            scenario_growth_and_reuse(os.path.join(tmp, "ctx%d" % k), threads, slab)
            scenario_api(os.path.join(tmp, "ctx%d" % k), threads, slab)
        except BaseException as e:                      # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def scenario_blocks_recycle(tmp, threads, slab):
    """100 directories of small files, a batch with size hints (its arena is there before the walk): what the walk reports
    about its blocks (MI_WALK_TIMING=1, stderr) is checked by the caller; here, the bytes"""
    rng = np.random.default_rng(9)
    d = os.path.join(tmp, "recycle")
    want = {}
    for k in range(100):
        os.makedirs(os.path.join(d, "d%03d" % k), exist_ok=True)
        for f in range(100):
            data = rng.integers(0, 256, 12000, dtype=np.uint8).tobytes()
            p = os.path.join(d, "d%03d" % k, "f%03d" % f)
            with open(p, "wb") as fh:
                fh.write(data)
            want[p] = data
    with M.Engine(n_streams=threads, staging_bytes=slab) as eng, eng.batch(10000, 10000 * 12288) as b:
        b.add_tree(d)
        b.run()
        check(b, b"".join(want[p] for p in sorted(want)), "recycle")


def scenario_long_strings(tmp, threads, slab):
    """Whole-string SHA-256 of strings too long for a GPU lane (route_long_strings): the reader threads hash them out of the arena
    (MI_FLAG_FILE_SHA256), host threads hash them out of the caller's memory (mi_sha256_many) -- SHA-NI streams, so these digests
    are RIGHT on the double too (its kernels do not run: the short strings' digests are its fill pattern)."""
    import hashlib
    rng = np.random.default_rng(9)
    long_sizes = [40 << 20, (9 << 20) + 13, 5 << 20]
    blobs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in long_sizes + [100, 4096, 0, 70_000]]
    with M.Engine(n_streams=threads, staging_bytes=slab, flags=M.FLAG_FILE_SHA256) as eng:
        with eng.batch() as b:
            paths = []
            for i, d in enumerate(blobs):
                if i % 2:
                    b.add_bytes(d)
                else:
                    p = os.path.join(tmp, "long%d" % i)
                    with open(p, "wb") as fh:
                        fh.write(d)
                    b.add_path(p, len(d))
            b.run()
            rows = b.files()
            for i in range(len(long_sizes)):
                assert rows["file_sha256"][i].tobytes() == hashlib.sha256(blobs[i]).digest(), "file %d" % i
            b.rerun()
            assert b.files()["file_sha256"][0].tobytes() == hashlib.sha256(blobs[0]).digest()
        got = eng.sha256_many(blobs)
        for i in range(len(long_sizes)):
            assert got[i] == hashlib.sha256(blobs[i]).digest(), "blob %d" % i
        assert len(got) == len(blobs)
        # what route_long_strings must NOT send to the host: many strings of one chunk-sized class (C2's shape) -- every lane is
        # busy, the pass is at its throughput roof; the double's lanes do not hash, so "stayed on the GPU" shows as "not hashlib's"
        same = [rng.integers(0, 256, 65536, dtype=np.uint8).tobytes() for _ in range(600)]
        got = eng.sha256_many(same)
        assert sum(1 for d, s_ in zip(got, same) if d == hashlib.sha256(s_).digest()) == 0, "chunk-sized strings left the GPU"
        # ... and a long string beside them leaves it (with them or without: whichever side finishes first by the model)
        got = eng.sha256_many(same + [blobs[0]])
        assert got[-1] == hashlib.sha256(blobs[0]).digest()


def main():
    tmp = sys.argv[1]
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    slab = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
    for name, fn in [("mix", scenario_mix), ("growth", scenario_growth_and_reuse), ("two", scenario_two_batches), ("interleaved", scenario_interleaved),
                     ("errors", scenario_errors), ("api", scenario_api), ("warm", scenario_warm), ("tree", scenario_tree_reserves_ahead),
                     ("two_ctxs", scenario_two_ctxs), ("recycle", scenario_blocks_recycle), ("long_strings", scenario_long_strings)]:
        if (name not in only) if only else name in ("recycle", "long_strings"):    # these run only when asked for
            continue
        fn(tmp, threads, slab)
        print("OK", name, flush=True)


if __name__ == "__main__":
    main()
