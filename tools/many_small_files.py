"""Host-fed rate with MANY SMALL files (what a node_modules-heavy layer looks like): a directory tree of
n_dirs x per_dir files of `kib` KiB in /dev/shm, scanned through mi_batch_add_tree (walk + open + read +
H2D + scan, all inside the library) and through a mi_batch_add_path loop.
usage: many_small_files.py [n_dirs] [per_dir] [kib]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import makisu_amd as M  # noqa: E402


def main():
    n_dirs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    per_dir = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    kib = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    root = tempfile.mkdtemp(prefix="mi_small_", dir="/dev/shm")
    try:
        rng = np.random.default_rng(0)
        blob = rng.integers(0, 256, kib << 10, dtype=np.uint8)
        paths = []
        for d in range(n_dirs):
            dd = os.path.join(root, "d%04d" % d)
            os.mkdir(dd)
            for f in range(per_dir):
                blob[:8] = np.frombuffer(np.uint64(d * per_dir + f).tobytes(), dtype=np.uint8)
                p = os.path.join(dd, "f%05d" % f)
                blob.tofile(p)
                paths.append(p)
        n, total = len(paths), len(paths) * (kib << 10)
        with M.Engine() as e:
            if os.environ.get("PREWARM"):                      # what of the first run's cost is the ctx's first host-fed batch?
                t0 = time.perf_counter()
                w = e.batch(1, 4 << 20)
                w.add_bytes(bytes(4 << 20), 0)
                w.run()
                w.free()
                print("prewarm: a 4 MiB host-fed batch first: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
            t0 = time.perf_counter()
            b = e.batch(n, total)
            print("batch begin with hints (%d files, %d MB): %.1f ms" % (n, total >> 20, (time.perf_counter() - t0) * 1e3), flush=True)
            for rep in range(3):
                b.reset()
                t0 = time.perf_counter()
                b.add_tree(root)
                t1 = time.perf_counter()
                b.run()
                t2 = time.perf_counter()
                print("add_tree   %d files x %d KiB: add %.1f ms (%.2f us/file), run %.1f ms, end to end %.1f GB/s"
                      % (n, kib, (t1 - t0) * 1e3, (t1 - t0) / n * 1e6, (t2 - t1) * 1e3, total / (t2 - t0) / 1e9), flush=True)
            sizes = [kib << 10] * n
            for rep in range(3):
                b.reset()
                t0 = time.perf_counter()
                b.add_paths(paths, sizes)
                t1 = time.perf_counter()
                b.run()
                t2 = time.perf_counter()
                print("add_paths  %d files x %d KiB: add %.1f ms (%.2f us/file, incl. building the argument), run %.1f ms, end to end %.1f GB/s"
                      % (n, kib, (t1 - t0) * 1e3, (t1 - t0) / n * 1e6, (t2 - t1) * 1e3, total / (t2 - t0) / 1e9), flush=True)
            for rep in range(2):
                b.reset()
                t0 = time.perf_counter()
                for i, p in enumerate(paths):
                    b.add_path(p, kib << 10, i)
                t1 = time.perf_counter()
                b.run()
                t2 = time.perf_counter()
                print("add_path   %d files x %d KiB: add %.1f ms (%.2f us/file, incl. the Python call), run %.1f ms, end to end %.1f GB/s"
                      % (n, kib, (t1 - t0) * 1e3, (t1 - t0) / n * 1e6, (t2 - t1) * 1e3, total / (t2 - t0) / 1e9), flush=True)
            b.free()
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
