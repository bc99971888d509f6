"""One read per file against two: a content-aware commit of a tree on a real file system,

    one   mi_memfs_commit_layer(fs, ctx, ...): every file is staged once; the layer tar is written from HBM;
    two   round 4's shape, call by call: mi_batch_add_tree + mi_batch_run (read 1: the stager), mi_memfs_add_layer_by_scan with the
          roots, then mi_layer_add(entry, src_path) per entry (read 2: the writer opens and reads every file of the layer again),

each with the page cache WARM and COLD (every file of the tree dropped with posix_fadvise(DONTNEED) after a sync; tmpfs
cannot be dropped -- the tool says what the tree lives on).  Prints wall seconds, what /proc/self/io counted (rchar = bytes
through read calls, read_bytes = bytes fetched from the storage layer) and the TarDigest (the same four times).

usage: one_read_vs_two.py [dir = $TMPDIR] [small files = 20000] [small bytes = 65536] [large files = 64] [large MiB = 16]"""
import ctypes as C
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd as M  # noqa: E402


def proc_io():
    d = dict(ln.split(": ") for ln in open("/proc/self/io").read().splitlines())
    return int(d["rchar"]), int(d.get("read_bytes", 0))


def fs_of(path):
    best = ("", "?")
    for ln in open("/proc/mounts"):
        f = ln.split()
        if len(f) >= 3 and (path == f[1] or path.startswith(f[1].rstrip("/") + "/")) and len(f[1]) > len(best[0]):
            best = (f[1], f[2])
    return best[1]


def drop_cache(paths):
    os.sync()
    for p in paths:
        fd = os.open(p, os.O_RDONLY)
        try:
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        finally:
            os.close(fd)


def make_tree(root, n_small, small, n_large, large):
    rng = np.random.default_rng(1)
    paths = []
    blob = bytearray(rng.integers(0, 256, small, dtype=np.uint8).tobytes())
    for i in range(n_small):
        if i % 200 == 0:
            d = os.path.join(root, "s%04d" % (i // 200))
            os.mkdir(d)
        blob[:8] = i.to_bytes(8, "little")
        p = os.path.join(d, "f%03d" % (i % 200))
        with open(p, "wb") as f:
            f.write(blob)
        paths.append(p)
    os.mkdir(os.path.join(root, "large"))
    big = bytearray(rng.integers(0, 256, large, dtype=np.uint8).tobytes())
    for i in range(n_large):
        big[:8] = (1 << 40 | i).to_bytes(8, "little")
        p = os.path.join(root, "large", "b%03d" % i)
        with open(p, "wb") as f:
            f.write(big)
        paths.append(p)
    return paths


def commit_one(eng, root):
    with M.MemFS(root) as fs:
        res = fs.commit_layer(must_scan=True, engine=eng, gzip_level=M.GZIP_OFF)
        return str(res["tar_digest"]), res["stats"]


def commit_two(eng, root):
    """round 4's six calls; the per-entry mi_layer_add loop runs over prebuilt ctypes rows (0.3 us of Python per entry)"""
    L = M.load_library()
    with M.MemFS(root) as fs, eng.batch() as b:
        n = b.add_tree(root, root, (), M.TREE_SCAN)
        b.run()
        roots = b.roots()
        walked = (M.TreeEntry * max(n, 1))()
        assert L.mi_batch_tree_entries(b._h, walked, n) == 0
        h, ne = C.c_void_p(), C.c_uint64()
        rc = L.mi_memfs_add_layer_by_scan(fs._h, walked, n, roots.ctypes.data, 32, C.byref(h), C.byref(ne))
        assert rc == 0, fs._lib.mi_memfs_error(fs._h)
        ne = ne.value
        ents, srcs = (M.TreeEntry * max(ne, 1))(), (C.c_char_p * max(ne, 1))()
        assert L.mi_copy_layer_entries(h, ents, srcs, ne) == 0
        cfg = M.LayerConfig()
        L.mi_layer_config_default(C.byref(cfg))
        cfg.gzip_level = M.GZIP_OFF
        lw = C.c_void_p()
        assert L.mi_layer_begin(C.byref(cfg), C.byref(lw)) == 0
        add = L.mi_layer_add
        for i in range(ne):
            rc = add(lw, C.byref(ents[i]), srcs[i] if ents[i].kind == 1 and srcs[i] else None)
            assert rc == 0, L.mi_layer_error(lw)
        res = M.LayerResult()
        assert L.mi_layer_finish(lw, C.byref(res)) == 0
        L.mi_layer_free(lw)
        L.mi_copy_layer_free(h)
        return str(M.Digest.from_raw(res.tar_sha256)), None


def main():
    a = sys.argv[1:]
    base = a[0] if a else tempfile.gettempdir()
    n_small, small = int(a[1]) if len(a) > 1 else 20000, int(a[2]) if len(a) > 2 else 65536
    n_large, large = int(a[3]) if len(a) > 3 else 64, (int(a[4]) if len(a) > 4 else 16) << 20
    root = tempfile.mkdtemp(prefix="mi_one_read_", dir=base)
    try:
        paths = make_tree(root, n_small, small, n_large, large)
        total = n_small * small + n_large * large
        print("%d files, %.2f GB (%d x %d B + %d x %d MiB) on %s (%s)" % (len(paths), total / 1e9, n_small, small, n_large, large >> 20, root, fs_of(root)))
        digests = set()
        with M.Engine(device=0) as eng:
            commit_one(eng, root)                                      # (the process's first walk and first batch: not timed)
            for cache in ("warm", "cold", "warm", "cold"):
                for name, fn in (("one", commit_one), ("two", commit_two)):
                    if cache == "cold":
                        drop_cache(paths)
                    r0, b0 = proc_io()
                    t0 = time.perf_counter()
                    dg, st = fn(eng, root)
                    dt = time.perf_counter() - t0
                    r1, b1 = proc_io()
                    digests.add(dg)
                    print("%-4s cache, %s read%s: %7.3f s   rchar %.2f x the tree, read from storage %.2f x   %s" %
                          (cache, name, " " if name == "one" else "s", dt, (r1 - r0) / total, (b1 - b0) / total,
                           ("(stage+scan %.3f, tar %.3f)" % (st["s_walk_stage"] + st["s_scan"], st["s_write"])) if st else ""), flush=True)
        print("TarDigest %s%s" % (sorted(digests)[0][:26], " -- the same every time" if len(digests) == 1 else " -- DIFFERENT DIGESTS: %s" % digests))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
