"""What is different in a process whose SHA chunk pass runs at 5.2 ms instead of 4.2 (the
per-process effect the workgroup pin does not reach)?  One process = one line: the serial chunk-pass times
of one C2 batch, the memory-side clocks while it runs, and three probes of the memory system that do not
involve this library's kernels at all -- a device-to-device copy (streaming bandwidth), a random 8-byte
gather over the batch's 6.5 GB (translation + latency bound) and the same gather over 64 MiB (latency
alone).  Run it several times in one gpurun call and compare lines.
    python tools/sha_proc_probe.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import makisu_amd as M  # noqa: E402
from makisu_amd import workloads as W  # noqa: E402
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(n):
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


sh = W.c2(0, 1)
out = {"pid": os.getpid()}
with M.Engine() as e:
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
    out["clocks"] = s.stop()
    out["sha_ms"] = [round(min(ms), 3), round(float(np.median(ms)), 3), round(max(ms), 3)]
    out["cdc_ms_median"] = round(float(np.median(cdc)), 3)
    out["valu_roof_GBps"] = round(e.sha_valu_roof() / 1e9, 1)
    # probes that use none of this library's kernels
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    big = torch.empty(sh.n_bytes // 8, dtype=torch.int64, device="cuda")          # 6.5 GB, like the arena
    big.random_(generator=g)
    dst = torch.empty_like(big)
    ms_copy = timed(lambda: dst.copy_(big))
    out["copy_GBps"] = round(2 * big.numel() * 8 / (ms_copy * 1e-3) / 1e9, 1)
    idx = torch.randint(0, big.numel(), (1 << 24,), device="cuda", generator=g)
    ms_g = timed(lambda: big[idx])
    out["gather_6GB_Mps"] = round(idx.numel() / (ms_g * 1e-3) / 1e6, 1)
    small = big[: (64 << 20) // 8]
    idx2 = torch.randint(0, small.numel(), (1 << 24,), device="cuda", generator=g)
    ms_g2 = timed(lambda: small[idx2])
    out["gather_64MB_Mps"] = round(idx2.numel() / (ms_g2 * 1e-3) / 1e6, 1)
    del big, dst, idx, idx2, small
    ms2 = []
    for i in range(reps):
        b.rerun()
        ms2.append(e.stats()["ms_sha_chunks"])
    out["sha_ms_again"] = [round(min(ms2), 3), round(float(np.median(ms2)), 3), round(max(ms2), 3)]
    b.free()
print(json.dumps(out))
