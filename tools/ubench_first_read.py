"""What the kernel charges for the FIRST read of freshly written page-cache files (round 5: a ctx's first staged batch runs at a third of
the steady rate; it is not the GPU, not the arena, not the DMA path -- a single thread's plain pread of files it has just written shows
the same: 0.193 s against 0.116 s for 3.2 GB in /dev/shm).  With and without posix_fadvise(POSIX_FADV_NOREUSE) before reading.
usage: ubench_first_read.py [dir = /dev/shm]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np


def run(base, advise):
    d = tempfile.mkdtemp(dir=base)
    blob = np.random.default_rng(1).integers(0, 256, 128 << 20, dtype=np.uint8)
    ps = []
    for i in range(24):
        p = os.path.join(d, "f%d" % i)
        blob.tofile(p)
        ps.append(p)
    buf = bytearray(8 << 20)
    out = []
    for rnd in range(3):
        t0 = time.perf_counter()
        for p in ps:
            fd = os.open(p, os.O_RDONLY)
            if advise is not None:
                os.posix_fadvise(fd, 0, 0, advise)
            off = 0
            while True:
                n = os.preadv(fd, [buf], off)
                if n <= 0:
                    break
                off += n
            os.close(fd)
        out.append(time.perf_counter() - t0)
    shutil.rmtree(d)
    return out


base = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
print(os.uname().release, base)
for name, adv in (("plain", None), ("FADV_NOREUSE", os.POSIX_FADV_NOREUSE), ("FADV_SEQUENTIAL", os.POSIX_FADV_SEQUENTIAL), ("plain", None)):
    r = run(base, adv)
    print("%-16s 3.2 GB, one thread: first read %.3f s, second %.3f, third %.3f" % (name, r[0], r[1], r[2]))
