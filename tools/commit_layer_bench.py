"""step.commitLayer end to end, with and without the GPU scan inside -- what the GPU buys (and costs) a build.

A tree of n files of `bytes` bytes in /dev/shm (page cache: both sides measure work, not a disk), two MemFS handles on
it -- one committing with a ctx (mi_memfs_commit_layer: walk + stage + GPU scan + content-aware diff + tar from HBM), one
without (the reference's commit: headers decide, the writer reads the changed files), and a third with a ctx and
MI_MEMFS_TRUST_CTIME (files whose inode is what it was when they were hashed are not read again) -- and three commits by scan each:
    all new            every file framed into the layer tar and digested;
    nothing changed    the reference: walk + lstat-level diff.  With a ctx: every file is read and hashed again -- the
                       price of watching content;
    0.1 % changed      n/1000 files rewritten (same size): nine in ten with a new mtime -- both see them -- one in ten
                       within the same second -- only the content scan sees those.
Wall seconds per commit, split as mi_commit_stats splits them (with a ctx the scan runs BESIDE the diff and the tar writer:
its seconds are not part of the sum).  bench.py puts the table into its JSON line
(`commit_e2e`); as a program: commit_layer_bench.py [files = 100000] [bytes = 4096]   (needs an MI355X)
MI_WALK_TIMING / MI_MEMFS_TIMING lines (stderr) show where the host time of each commit went."""
import os
import shutil
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd as M  # noqa: E402

MTIME = 1_700_000_000


def _make_tree(root, n, size, per_dir):
    rng = np.random.default_rng(n ^ size)
    blob = bytearray(rng.integers(0, 256, size, dtype=np.uint8).tobytes())
    paths = []
    for i in range(n):
        if i % per_dir == 0:
            dn = os.path.join(root, "d%05d" % (i // per_dir))
            os.mkdir(dn)
        blob[:8] = int(i).to_bytes(8, "little")                    # distinct files
        p = os.path.join(dn, "f%04d" % (i % per_dir))
        with open(p, "wb") as f:
            f.write(blob)
        os.utime(p, (MTIME, MTIME))
        paths.append(p)
    return paths


def _side(st, res, wall):
    # s_total: the harness's wall clock around MemFS.commit_layer -- the C call (s_call: the library's own clock around it) plus
    # ctypes' own overhead; since round 6 the harness does not turn the layer's entries into dicts inside the timed call (0.1 s per
    # 100 000 entries of python, +-20 ms of allocator noise that fell on whichever side ran first)
    return {"s_total": round(wall, 4), "s_call": round(st["s_total"], 4), "s_walk_stage": round(st["s_walk_stage"], 4), "s_scan": round(st["s_scan"], 4),
            "s_diff": round(st["s_diff"], 4), "s_write": round(st["s_write"], 4), "layer_entries": int(res["n_entries"]),
            "layer_files": int(st["n_layer_files"]), "tar_bytes": int(res["tar_bytes"]),
            "files_read": int(st["files_opened"]), "bytes_read": int(st["file_bytes_read"]),
            "content_only_changes": int(st["n_content_changed"]), "scan_overlapped": bool(st["pipelined"]),
            "files_trusted": int(st["n_content_trusted"]), "files_verified": int(st["n_verified_files"]), "chunks_refetched": int(st["n_refetched"]),
            "arena_moves": int(st["arena_moves"]), "arena_pieces": int(st["arena_pieces"])}


def commit_e2e(eng, n_files, file_bytes, base=None, gzip_level=None, all_new_rounds=1):
    """-> dict(tree, commits=[{what, gpu={...}, cpu_header_only={...}}, ...]); gzip off unless gzip_level is given.
    all_new_rounds > 1: a handle can commit a tree "all new" only once, and one sample of a 0.3 s commit moves by 15 % with the host
    (profiles/r06_bench_n1_*.json: the header-only side 0.318 ... 0.368 s on four boxes) -- so k - 1 more rounds on FRESH handles of
    the three sides first, all of them in the first row as `all_new_rounds_s` (bench.py's summary takes each side's best)."""
    base = base or ("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None)
    import tempfile
    root = tempfile.mkdtemp(prefix="mi_commit_e2e_", dir=base)
    gz = M.GZIP_OFF if gzip_level is None else gzip_level
    try:
        per_dir = 200 if file_bytes < (1 << 20) else 8
        paths = _make_tree(root, n_files, file_bytes, per_dir)
        time.sleep(0.05)
        rng = np.random.default_rng(3)
        k = max(1, n_files // 1000)
        victims = [paths[int(i)] for i in rng.choice(n_files, size=k, replace=False)]
        out = {"tree": "%d files x %d bytes in %d directories under %s (page cache)" % (n_files, file_bytes, (n_files + per_dir - 1) // per_dir, base or "TMPDIR"),
               "tree_bytes": n_files * file_bytes, "gzip": "off" if gzip_level is None else gzip_level, "commits": []}
        rounds = {"gpu": [], "gpu_trust_ctime": [], "cpu_header_only": []}
        for _ in range(max(0, all_new_rounds - 1)):
            with M.MemFS(root) as gpu, M.MemFS(root) as trust, M.MemFS(root) as plain:
                trust.set_options(trust_ctime=True)
                sides = (("gpu", gpu, {"engine": eng}), ("gpu_trust_ctime", trust, {"engine": eng}), ("cpu_header_only", plain, {}))
                if os.environ.get("MI_BENCH_ORDER") == "cpu_first":
                    sides = sides[::-1]
                for name, fs, kw in sides:
                    t0 = time.perf_counter()
                    fs.commit_layer(must_scan=True, gzip_level=gz, want_layer=False, **kw)
                    rounds[name].append(round(time.perf_counter() - t0, 4))
        with M.MemFS(root) as gpu, M.MemFS(root) as trust, M.MemFS(root) as plain:
            trust.set_options(trust_ctime=True)
            for step, what in enumerate(("all new", "nothing changed", "0.1 % changed")):
                if step == 2:
                    n_same_second = 0
                    for j, p in enumerate(victims):
                        data = rng.integers(0, 256, file_bytes, dtype=np.uint8).tobytes()
                        with open(p, "r+b") as f:
                            f.write(data)
                        if j % 10 == 9 or j == k - 1:
                            os.utime(p, (MTIME, MTIME))             # the same second: tario.IsSimilarHeader cannot tell
                            n_same_second += 1
                        else:
                            os.utime(p, (MTIME + 7, MTIME + 7))
                    what = "%d files rewritten (0.1 %%), %d of them within the same second" % (k, n_same_second)
                    time.sleep(0.05)                                 # (older than the racy-clean slack of MI_MEMFS_TRUST_CTIME)
                row = {"what": what}
                sides = (("gpu", gpu, {"engine": eng}), ("gpu_trust_ctime", trust, {"engine": eng}), ("cpu_header_only", plain, {}))
                if os.environ.get("MI_BENCH_ORDER") == "cpu_first":     # (which side pays a process's first big commit: allocator, page faults)
                    sides = sides[::-1]
                for name, fs, kw in sides:
                    t0 = time.perf_counter()
                    res = fs.commit_layer(must_scan=True, gzip_level=gz, want_layer=False, **kw)   # (the layer's entries stay in the library:
                                                                                                    #  100 000 python dicts are not part of a commit)
                    row[name] = _side(res["stats"], res, time.perf_counter() - t0)
                    if step == 0:
                        rounds[name].append(row[name]["s_total"])
                if step == 0 and all_new_rounds > 1:
                    row["all_new_rounds_s"] = rounds
                out["commits"].append(row)
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    gz = os.environ.get("MI_BENCH_GZIP")                          # e.g. -1: the reference's default compression (tario.CompressionLevel)
    with M.Engine(device=0) as eng:
        res = commit_e2e(eng, n, size, gzip_level=int(gz) if gz not in (None, "") else None)
    print(res["tree"])
    for row in res["commits"]:
        print("  " + row["what"])
        for side in ("gpu", "gpu_trust_ctime", "cpu_header_only"):
            r = row[side]
            print("    %-16s %7.3f s (the C call %.3f) = walk%s %.3f + diff %.3f + tar %.3f; scan %.3f%s | layer: %d entries, %d files, %d tar bytes | "
                  "read %d files, %d bytes (%d trusted) | content-only changes %d" %
                  (side, r["s_total"], r["s_call"], "+stage" if side.startswith("gpu") else "", r["s_walk_stage"], r["s_diff"], r["s_write"], r["s_scan"],
                   " (beside diff and tar)" if r["scan_overlapped"] else "",
                   r["layer_entries"], r["layer_files"], r["tar_bytes"], r["files_read"], r["bytes_read"], r["files_trusted"], r["content_only_changes"]))


if __name__ == "__main__":
    main()
