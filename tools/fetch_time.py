import sys, time, os
sys.path.insert(0, "/root/repo")
import torch
import makisu_amd
with makisu_amd.Engine() as e:
    with e.batch() as b:
        b.add_synthetic([65536] * 100000, list(range(100000)))
        b.run()
        for rep in range(3):
            b.rerun()
            t0 = time.perf_counter(); ch = b.chunks(); t1 = time.perf_counter(); fl = b.files(); t2 = time.perf_counter()
            print("chunks() %.2f ms (%d rows)  files() %.2f ms (%d rows)" % ((t1 - t0) * 1e3, len(ch), (t2 - t1) * 1e3, len(fl)))
