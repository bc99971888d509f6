"""Where batching pays: the scan of ONE batch of `n` files x `kib` KiB on the GPU -- bytes handed over from host memory
(mi_batch_add_bytes into a reused batch: the cgo caller's path) and already resident -- next to the same work on the
host cores (the oracle's scanner on all usable cores: what a caller without a GPU would run).  One lane hashes one
chunk, so a batch cannot finish faster than its longest chunk (64 bytes per ~2.8 us: a 64 KiB chunk takes ~3 ms); the
table says from which batch size on that floor is hidden, i.e. how many layers a host should put into one batch
(INTEGRATION.md "Batching policy").
usage: batch_crossover.py [kib=64] -> one line per batch size"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import makisu_amd as M  # noqa: E402
from oracle import mi_oracle as O  # noqa: E402   (the CPU side of the comparison; a tool, not the product)
import bench  # noqa: E402


def main():
    kib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    O.build()
    cores, _ = bench.usable_cores()
    p = O.CdcParams(0x4D414B49, 13, 2048, 65536)
    rng = np.random.default_rng(1)
    print("# batch of n files x %d KiB: GPU end to end from host memory (reused batch, add_bytes + run + counts), GPU with the "
          "bytes resident (rerun), host scanner on %d cores; best of 5" % (kib, cores))
    with M.Engine() as e:
        b = e.batch()
        for n in (16, 64, 256, 1024, 2048, 4096, 16384):
            size = kib << 10
            data = rng.integers(0, 256, n * size, dtype=np.uint8)
            offs = np.arange(n, dtype=np.uint64) * size
            sizes = np.full(n, size, dtype=np.uint64)
            blobs = [data[i * size:(i + 1) * size] for i in range(n)]
            best_fed, best_res, best_cpu = 1e9, 1e9, 1e9
            for rep in range(5):
                b.reset()
                t0 = time.perf_counter()
                for i, x in enumerate(blobs):
                    b.add_bytes(x, i)
                b.run()
                b.counts()
                best_fed = min(best_fed, time.perf_counter() - t0)
                t0 = time.perf_counter()
                b.rerun()
                best_res = min(best_res, time.perf_counter() - t0)
                t0 = time.perf_counter()
                O.scan_batch(data, offs, sizes, p, True, cores, 0)
                best_cpu = min(best_cpu, time.perf_counter() - t0)
            mb = n * size / 1e6
            print("%6d files, %8.1f MB: GPU host-fed %7.2f ms = %6.2f GB/s | resident %6.2f ms = %7.2f GB/s | %d host cores %7.2f ms = %5.2f GB/s"
                  % (n, mb, best_fed * 1e3, mb / best_fed / 1e3, best_res * 1e3, mb / best_res / 1e3, cores, best_cpu * 1e3, mb / best_cpu / 1e3),
                  flush=True)
        b.free()


if __name__ == "__main__":
    main()
