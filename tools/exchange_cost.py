"""Dev tool: what the job-wide marking costs one rank of an 8-GPU run, measured on ONE GPU.

The xGMI all-gather cannot be run here; everything after it can: the rank's own C2 digest set
is replicated `--world` times (byte 0 of each replica xor'ed with the replica number so the
keys stay distinct, like other ranks' chunks) and marked with mi_dedup_mark, alone and while
the scan pipeline keeps `--inflight` batches running.  Prints ms per marking and the step time
with / without it."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd  # noqa: E402
from makisu_amd import distributed as mdist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--files", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--inflight", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    eng = makisu_amd.Engine(flags=makisu_amd.FLAG_NO_DEDUP)
    batches = []
    for i in range(a.inflight):
        b = eng.batch()
        b.add_synthetic([65536] * a.files, list(range(i * a.files, (i + 1) * a.files)))
        b.run()
        batches.append(b)

    def fake_gather(b):
        loc = mdist.digests_tensor(b, dev)
        reps = []
        for r in range(a.world):
            t = loc.clone()
            t[:, 0] ^= r
            reps.append(t)
        return torch.cat(reps, 0).contiguous()

    glob = fake_gather(batches[0])
    dup = torch.empty(glob.shape[0], dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    n_own = glob.shape[0] // a.world
    for _ in range(3):
        t0 = time.perf_counter()
        nu = eng.dedup_mark(glob.data_ptr(), glob.shape[0], dup.data_ptr())
        dt = time.perf_counter() - t0
        print("full marking: %d rows -> %d unique, %.3f ms (kernel %.3f)" %
              (glob.shape[0], nu, dt * 1e3, eng.stats()["ms_dedup"]))
    for r in (0, a.world // 2, a.world - 1):
        t0 = time.perf_counter()
        nf = eng.dedup_mark_range(glob.data_ptr(), glob.shape[0], r * n_own, n_own, dup.data_ptr())
        dt = time.perf_counter() - t0
        print("range marking as rank %d: %d own rows, %d first, %.3f ms (kernel %.3f)" %
              (r, n_own, nf, dt * 1e3, eng.stats()["ms_dedup"]))

    def loop(with_mark):
        pending = []
        t0 = time.perf_counter()
        for k in range(a.steps):
            if len(pending) == a.inflight:
                b = pending.pop(0)
                b.wait()
                if with_mark:
                    g = fake_gather(b)
                    torch.cuda.current_stream(dev).synchronize()
                    if with_mark == "full":
                        eng.dedup_mark(g.data_ptr(), g.shape[0], dup.data_ptr())
                    else:                      # as the LAST rank: the most probes
                        eng.dedup_mark_range(g.data_ptr(), g.shape[0], (a.world - 1) * n_own, n_own,
                                             dup.data_ptr())
                    b.set_global_dedup(dup.data_ptr(), 0)
            b = batches[k % a.inflight]
            b.submit()
            pending.append(b)
        for b in pending:
            b.wait()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    for rep in range(2):
        print("step without marking: %.3f ms   with %d-rank full marking: %.3f ms   range marking (last rank): %.3f ms" %
              (loop(False), a.world, loop("full"), loop("range")))
    eng.close()


if __name__ == "__main__":
    main()
