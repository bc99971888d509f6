"""What a process's FIRST GPU commit pays, and for what (round 6: 0.08-0.15 s once, profiles/r06_first_commit_order.txt).
One process per variant (first use happens once):  cold -- ctx, then the commit;  warm_synth -- a 4-file synthetic batch first
(the kernels' code objects);  warm_paths -- a 4-file batch by path first (reader threads, pinned slabs, the copy path);  warm_both;
warm_call -- mi_ctx_warm (what the library offers for this: all reader threads + a four-file synthetic batch).
Prints the seconds of each step and of three commits of fresh 100 x 64 KiB trees (all new).
usage: first_commit_probe.py <cold|warm_synth|warm_paths|warm_both|warm_call>"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import makisu_amd as M  # noqa: E402
from commit_cases import write_file  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "cold"
tmp = tempfile.mkdtemp(prefix="mi_first_")
rng = np.random.default_rng(1)
roots = []
for r in range(3):
    root = os.path.join(tmp, "r%d" % r)
    for i in range(100):
        write_file(os.path.join(root, "d%d/f%d.bin" % (i % 5, i)), rng.integers(0, 256, 65536, dtype=np.uint8).tobytes(), 0o644, 1_600_000_000)
    roots.append(root)
warm_files = []
for i in range(4):
    p = os.path.join(tmp, "w%d.bin" % i)
    write_file(p, rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes())
    warm_files.append(p)
out = []
t0 = time.perf_counter()
eng = M.Engine(device=0)
out.append(("ctx", time.perf_counter() - t0))
if mode in ("warm_synth", "warm_both"):
    t0 = time.perf_counter()
    with eng.batch() as b:
        b.add_synthetic([65536] * 4, None, seed=1)
        b.run()
    out.append(("synthetic batch", time.perf_counter() - t0))
if mode in ("warm_paths", "warm_both"):
    t0 = time.perf_counter()
    with eng.batch() as b:
        b.add_paths(warm_files)
        b.run()
    out.append(("batch by path", time.perf_counter() - t0))
if mode == "warm_call":
    t0 = time.perf_counter()
    eng.warm()
    out.append(("mi_ctx_warm", time.perf_counter() - t0))
for r, root in enumerate(roots):
    with M.MemFS(root) as fs:
        t0 = time.perf_counter()
        fs.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF, engine=eng, want_layer=False)
        out.append(("commit %d" % r, time.perf_counter() - t0))
    with M.MemFS(root) as fs:
        t0 = time.perf_counter()
        fs.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF, want_layer=False)
        out.append(("header-only %d" % r, time.perf_counter() - t0))
# ... and ONE handle that lives on (a build stage): 100 new files per commit, the handle's batch, windows and arena reused
live = os.path.join(tmp, "live")
write_file(os.path.join(live, "seed.bin"), b"x" * 1000, 0o644, 1_600_000_000)
with M.MemFS(live) as fs, M.MemFS(live) as plain:
    for k in range(4):
        for i in range(100):
            write_file(os.path.join(live, "k%d/f%d.bin" % (k, i)), rng.integers(0, 256, 65536, dtype=np.uint8).tobytes(), 0o644, 1_600_000_000)
        t0 = time.perf_counter()
        res = fs.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF, engine=eng, want_layer=False)
        out.append(("same handle %d" % k, time.perf_counter() - t0))
        if os.environ.get("MI_PROBE_STATS") == "1":
            st = res["stats"]
            print("   same handle %d: " % k + " ".join("%s %.4f" % (n, st[n]) for n in ("s_walk_stage", "s_scan", "s_diff", "s_write", "s_total")), file=sys.stderr)
        t0 = time.perf_counter()
        plain.commit_layer(must_scan=True, gzip_level=M.GZIP_OFF, want_layer=False)
        out.append(("header-only", time.perf_counter() - t0))
eng.close()
print("%-11s " % mode + "  ".join("%s %.4f" % kv for kv in out))
