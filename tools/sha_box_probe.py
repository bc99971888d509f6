"""Why do boxes of the pool hash C2 at different rates with the same VALU roof?  One process, one C2 batch
per load scheme: serial SHA chunk-pass launch times (lane-owned vs quad-cooperative loads), the same-run
VALU roof, sclk/power while it runs.  python tools/sha_box_probe.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import makisu_amd as M  # noqa: E402
from makisu_amd import workloads as W  # noqa: E402
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
sh = W.c2(0, 1)
out = {}
for name, scheme in (("lane", M.SHA_LOADS_LANE), ("coop", M.SHA_LOADS_COOP), ("lane_again", M.SHA_LOADS_LANE)):
    with M.Engine(sha_load_scheme=scheme) as e:
        b = e.batch(sh.n_files, sh.n_bytes)
        b.add_synthetic(sh.sizes, sh.cids, seed=sh.seed)
        b.run()
        s = bench.ClockSampler(None)
        s.start()
        ms, cdc = [], []
        for i in range(reps):
            b.rerun()
            st = e.stats()
            ms.append(st["ms_sha_chunks"])
            cdc.append(st["ms_cdc"])
        clk = s.stop()
        roof = e.sha_valu_roof() / 1e9
        alg = st["bytes_in"] + 52 * st["n_chunks"]
        out[name] = {"sha_ms_min": round(min(ms), 3), "sha_ms_median": round(float(np.median(ms)), 3), "sha_ms_max": round(max(ms), 3),
                     "cdc_ms_median": round(float(np.median(cdc)), 3), "valu_roof_GBps": round(roof, 1),
                     "frac_of_roof_median": round(alg / (float(np.median(ms)) * 1e-3) / 1e9 / roof, 3), "clocks": clk}
        b.free()
print(json.dumps(out))
