"""Where does a ctx's FIRST host-fed batch spend its time (round 5: 0.28 s in the first mi_batch_add_tree of 48 x 128 MiB against 0.001 s
in the second)?  Times, on a fresh ctx: a device allocation by size (mi_batch_begin with a bytes hint), the reader threads' bring-up
(the first mi_batch_add_bytes of a megabyte), then add_tree + run of a tree of large files three times on one reused batch.
usage: first_use_probe.py [files = 48] [MiB = 128]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd as M  # noqa: E402


def t(f):
    t0 = time.perf_counter()
    r = f()
    return r, time.perf_counter() - t0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    root = tempfile.mkdtemp(prefix="mi_first_use_", dir="/dev/shm")
    try:
        blob = np.random.default_rng(1).integers(0, 256, mib << 20, dtype=np.uint8)
        for i in range(n):
            blob[:8] = np.frombuffer(np.uint64(i).tobytes(), dtype=np.uint8)
            blob.tofile(os.path.join(root, "f%03d" % i))
        ns = int(os.environ.get("MI_PROBE_STREAMS", "0"))
        eng, dt = t(lambda: M.Engine(device=0, n_streams=ns) if ns else M.Engine(device=0))
        print("mi_ctx_create                      %.3f s" % dt)
        for gb in (1, 4, 8, 16):
            b, dt = t(lambda: eng.batch(0, gb << 30))
            _, df = t(b.free)
            print("mi_batch_begin, arena for %2d GiB    %.3f s (free %.3f)" % (gb, dt, df))
        b = eng.batch()
        _, dt = t(lambda: b.add_bytes(bytes(2 << 20)))
        print("first mi_batch_add_bytes of 2 MiB  %.3f s (the reader threads come up: a pinned slab and a stream each)" % dt)
        b.free()
        if os.environ.get("MI_PROBE_WARM_GIB"):                  # does the FIRST traffic of the process pay, or every fresh arena's?
            g = int(os.environ["MI_PROBE_WARM_GIB"])
            blob1 = np.zeros(g << 30, dtype=np.uint8)
            for r in range(2):
                bw = eng.batch()
                _, dt = t(lambda: (bw.add_bytes(blob1), bw.run()))
                print("warm-up batch %d: %d GiB from host memory in %.3f s" % (r, g, dt))
                bw.free()
        b = eng.batch()
        for k in range(3):
            if k:
                b.reset()
            _, d1 = t(lambda: b.add_tree(root, root, (), M.TREE_SCAN))
            _, d2 = t(b.run)
            st = eng.stats()
            print("round %d: add_tree %.3f s, run %.3f s  (%.1f GB/s end to end); device: h2d %.1f ms, cdc %.1f, sort %.1f, sha %.1f, files %.1f, "
                  "dedup %.1f, first kernel to last %.1f" % (k, d1, d2, n * (mib << 20) / (d1 + d2) / 1e9, st["ms_h2d"], st["ms_cdc"], st["ms_sort"],
                                                        st["ms_sha_chunks"], st["ms_sha_files"], st["ms_dedup"], st["ms_total"]))
        b.free()
        eng.close()
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
