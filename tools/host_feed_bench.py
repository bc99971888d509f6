"""PCIe-inclusive rate: host buffers -> mi_batch_add_bytes (pinned ring + hipMemcpyAsync) -> scan."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd  # noqa: E402


def main():
    n, size = 64, 64 << 20
    rng = np.random.default_rng(0)
    blob = rng.integers(0, 256, size, dtype=np.uint8)
    with makisu_amd.Engine() as e:
        for rep in range(3):
            b = e.batch(n, n * size)
            t0 = time.perf_counter()
            for i in range(n):
                b.add_bytes(blob, tag=i)
            t1 = time.perf_counter()
            b.run()
            t2 = time.perf_counter()
            st = e.stats()
            print("host feed: %d x %d MiB: add %.3f s (%.1f GB/s into the pinned ring + H2D), run %.3f s, "
                  "end to end %.1f GB/s, device pipeline alone %.1f GB/s"
                  % (n, size >> 20, t1 - t0, n * size / (t1 - t0) / 1e9, t2 - t1,
                     n * size / (t2 - t0) / 1e9, st["bytes_in"] / st["ms_total"] / 1e6))
            b.free()


if __name__ == "__main__":
    main()
