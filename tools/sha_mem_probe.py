"""Is a process's SHA chunk-pass speed a property of WHERE its arena lies?  One process, one ctx; before
every round a dummy device allocation of a different size shifts where the next arena lands; prints the
serial SHA launch time (median of 12) per round.  python tools/sha_mem_probe.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import makisu_amd as M  # noqa: E402
from makisu_amd import workloads as W  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sh = W.c2(0, 1)
out = []
with M.Engine() as e:
    keep = []
    for r in range(rounds):
        gib = [0, 3, 11, 40, 1, 90, 7, 0][r % 8]
        dummy = torch.empty(gib << 30, dtype=torch.uint8, device="cuda") if gib else None   # stays while the arena is made
        b = e.batch(sh.n_files, sh.n_bytes)
        b.add_synthetic(sh.sizes, sh.cids, seed=sh.seed)
        b.run()
        ms = []
        for i in range(12):
            b.rerun()
            ms.append(e.stats()["ms_sha_chunks"])
        out.append((gib, round(float(np.median(ms)), 3), round(min(ms), 3)))
        b.free()
        del dummy
        torch.cuda.empty_cache()
print("dummy GiB, sha median ms, min ms per round:", out)
