"""Long form of tests/test_gpu_staging.py::test_host_fed_soak: the host-fed
4 x 1 GiB + 300-small mix for many rounds, with and without MI_FLAG_VERIFY_STAGING, including the
first copies into fresh VRAM right after a >= 100 GB arena was freed; prints one JSON line per mode
and the staging rate with / without the flag.   python tools/stage_soak.py [rounds]"""
import json
import os
import pathlib
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from oracle import mi_oracle as O  # noqa: E402
import test_gpu_staging as T  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
O.build()
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    for verify in (False, True):
        t0 = time.time()
        res = T.soak(O, pathlib.Path(d), rounds, verify)
        res["verify"] = verify
        res["seconds"] = round(time.time() - t0, 1)
        print(json.dumps(res), flush=True)
    # cost of the flag: staging rate of the reused batch (no allocation in the timed rounds)
    import makisu_amd
    paths, sizes, extra, blobs = T._mix_inputs(O, pathlib.Path(d))
    nbytes = sum(len(x) for x in blobs)
    for verify in (False, True):
        with makisu_amd.Engine(flags=makisu_amd.FLAG_VERIFY_STAGING if verify else 0) as e, e.batch() as b:
            best = 0.0
            for rep in range(6):
                b.reset()
                t0 = time.time()
                T._fill_mix(b, paths, sizes, extra)
                b.run()
                dt = time.time() - t0
                if rep:
                    best = max(best, nbytes / dt / 1e9)
            ss = b.stage_stats()
            print(json.dumps({"verify": verify, "host_fed_GBps_best_of_5": round(best, 2), "bytes": nbytes,
                              "spans": ss["spans"], "ms_verify_summed_over_threads": round(ss["ms_verify"], 1)}), flush=True)
