"""Latency of a small layer (C1-sized: 28 files, ~10 KB) through the ABI: fresh batch vs reused batch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd  # noqa: E402

rng = np.random.default_rng(0)
blobs = [rng.integers(0, 256, int(n), dtype=np.uint8).tobytes() for n in rng.integers(10, 2000, 28)]
with makisu_amd.Engine() as e:
    for mode in ("fresh", "fresh", "fresh", "reset", "reset", "reset"):
        t0 = time.perf_counter()
        if mode == "fresh":
            b = e.batch()
        else:
            b.reset()
        t1 = time.perf_counter()
        for i, x in enumerate(blobs):
            b.add_bytes(x, i)
        t2 = time.perf_counter()
        b.run()
        t3 = time.perf_counter()
        f, c = b.files(), b.chunks()
        t4 = time.perf_counter()
        if mode == "fresh":
            pass
        print("%s: begin/reset %.3f ms, add %.3f ms, run %.3f ms, results %.3f ms, total %.3f ms (device pipeline %.3f ms)"
              % (mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3, e.stats()["ms_total"]))
        if mode == "fresh":
            keep = b
    # medium layer: 2000 files x 16 KiB
    blobs2 = [rng.integers(0, 256, 16384, dtype=np.uint8).tobytes() for _ in range(2000)]
    for _ in range(3):
        b.reset()
        t0 = time.perf_counter()
        for i, x in enumerate(blobs2):
            b.add_bytes(x, i)
        b.run()
        f, c = b.files(), b.chunks()
        print("2000 x 16 KiB reset: total %.3f ms (device %.3f)" % ((time.perf_counter() - t0) * 1e3, e.stats()["ms_total"]))
