"""PCIe-inclusive rate: host memory / page-cache files -> reader threads + pinned slabs -> H2D -> scan.
Usage: host_feed_bench.py [n_files] [MiB per file]   (MI_STAGE_THREADS picks the reader-thread count)"""
import os
import sys
import tempfile
import time

import numpy as np

if os.environ.get("MI_FEED_TORCH"):      # PyTorch-ROCm bundles its own (older) HIP runtime: loaded first it serves the engine too
    import torch  # noqa: F401
    if os.environ.get("MI_FEED_TORCH") == "cuda":
        torch.cuda.set_device(0)
        torch.cuda.synchronize()
        print("torch cuda initialised", torch.cuda.is_available(), flush=True)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    size = (int(sys.argv[2]) if len(sys.argv) > 2 else 128) << 20
    rng = np.random.default_rng(0)
    blob = rng.integers(0, 256, size, dtype=np.uint8)
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="mi_feed_", dir=base)
    paths = []
    for i in range(n):
        blob[:8] = np.frombuffer(np.uint64(i).tobytes(), dtype=np.uint8)
        p = os.path.join(d, "f%04d" % i)
        blob.tofile(p)
        paths.append(p)
    thr = os.environ.get("MI_STAGE_THREADS", "default")
    flags = makisu_amd.FLAG_FILE_SUMS if os.environ.get("MI_FEED_SUMS") else 0       # (the end-to-end sums' cost on the reader threads)
    with makisu_amd.Engine(flags=flags) as e:
        b = e.batch(n, n * size)
        for mode in (os.environ.get("MI_FEED_MODES") or "add_bytes,add_bytes,add_path,add_path,add_bytes,add_path").split(","):
            if os.environ.get("MI_FEED_FRESH"):        # a new arena every pass (fresh VRAM is cleared by the driver)
                b.free()
                b = e.batch(n, n * size)
            else:
                b.reset()
            t0 = time.perf_counter()
            for i in range(n):
                if mode == "add_bytes":
                    b.add_bytes(blob, tag=i)
                else:
                    b.add_path(paths[i], size, i)
            t1 = time.perf_counter()
            b.run()
            t2 = time.perf_counter()
            st = e.stats()
            print("threads %s %s: %d x %d MiB: add %.3f s, run %.3f s (ms_h2d %.1f), end to end %.1f GB/s, "
                  "device pipeline alone %.1f GB/s"
                  % (thr, mode, n, size >> 20, t1 - t0, t2 - t1, st["ms_h2d"], n * size / (t2 - t0) / 1e9,
                     st["bytes_in"] / st["ms_total"] / 1e6), flush=True)
        b.free()
    for p in paths:
        os.unlink(p)
    os.rmdir(d)


if __name__ == "__main__":
    main()
