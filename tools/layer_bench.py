"""The reference's own hot path as this library restates it (mi_layer: tar framing + SHA-256 of the
tar + gzip + SHA-256 of the blob, host threads, no GPU): GB/s of file bytes by gzip level, next to
the GPU content scan of the same files fed from the page cache.  usage: layer_bench.py [files] [MiB each]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import makisu_amd as M  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    root = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rng = np.random.default_rng(3)
        # half random (incompressible), half text-like (compressible) content
        line = b"the quick brown fox jumps over the lazy dog 0123456789\n"
        text = (line * ((mib << 20) // len(line) + 1))[: mib << 20]
        paths = []
        for i in range(n):
            p = os.path.join(root, "f%05d.bin" % i)
            with open(p, "wb") as f:
                f.write(rng.integers(0, 256, mib << 20, dtype=np.uint8).tobytes() if i % 2 == 0 else text)
            paths.append(p)
        total = n * (mib << 20)
        ent = [dict(relpath="f%05d.bin" % i, kind=M.KIND_FILE, mode=0o644, uid=0, gid=0, size=mib << 20,
                    mtime_sec=1, link_target=None) for i in range(n)]
        for name, level in (("no gzip", M.GZIP_OFF), ("gzip 1 (speed)", 1), ("gzip default", M.GZIP_DEFAULT)):
            if level != M.GZIP_OFF and total > (1 << 30):
                sub = max(1, n * (1 << 30) // total)          # gzip is slow: one GiB is enough
            else:
                sub = n
            fd = os.open(os.devnull, os.O_WRONLY)
            t0 = time.perf_counter()
            with M.Layer(out_fd=fd, gzip_level=level) as l:
                for e, p in zip(ent[:sub], paths[:sub]):
                    l.add(e, p)
                r = l.finish()
            dt = time.perf_counter() - t0
            os.close(fd)
            print("mi_layer %-15s %6.2f GB/s of file bytes (%d files, %.1f GB, tar %d B, blob %d B)"
                  % (name, sub * (mib << 20) / dt / 1e9, sub, sub * (mib << 20) / 1e9, r["tar_bytes"], r["gzip_bytes"]))
        try:
            import torch  # noqa: F401
            with M.Engine() as e:
                for rep in range(2):
                    with e.batch(n, total) as b:
                        t0 = time.perf_counter()
                        for p in paths:
                            b.add_path(p, mib << 20)
                        b.run()
                        dt = time.perf_counter() - t0
                        st = e.stats()
                print("GPU content scan of the same files (host-fed, one batch): %.2f GB/s end to end, %d chunks"
                      % (total / dt / 1e9, st["n_chunks"]))
        except Exception as ex:  # noqa: BLE001  (no GPU here)
            print("GPU scan skipped:", type(ex).__name__, str(ex)[:80])
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
