"""Per-wave record of the SHA chunk pass (MI_SHA_WAVE_STATS, csrc/sha256.hip): where every wave of the
persistent grid ran (XCC / SE / CU / SIMD), when it started and ended, how many loop iterations it made
and how many lane-blocks it hashed.  The grid pulls strings from shared queues, so all waves run for the
whole launch and a wave's iteration count IS its pace: this prints the pace by XCC, by how many waves
shared the wave's SIMD and CU, and the slowest / fastest CUs.

    python tools/sha_wave_stats.py            # one process: C2 batch, a few launches, analysis of each
    python tools/sha_wave_stats.py --read F   # analyse a file written earlier
"""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_records(path):
    raw = np.fromfile(path, dtype=np.uint32)
    out, p = [], 0
    while p + 4 <= raw.size:
        grid, wpw, coop, n = (int(x) for x in raw[p:p + 4])
        nw = grid * wpw
        body = raw[p + 4:p + 4 + nw * 8].reshape(nw, 8)
        out.append({"grid": grid, "waves_per_wg": wpw, "coop": coop, "n": n, "w": body})
        p += 4 + nw * 8
    return out


def analyse(rec):
    w = rec["w"]
    hw, xcc, role = w[:, 0], w[:, 1] & 0xF, (w[:, 1] >> 8) & 0xFF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    t0 = w[:, 2].astype(np.uint64)                     # low word of the 100 MHz clock: launches last milliseconds
    t0 = t0 + np.where(t0 < t0.max() // 2 if t0.max() - t0.min() > (1 << 31) else False, np.uint64(1 << 32), np.uint64(0))
    wait_ticks = w[:, 3].astype(np.float64)
    t1 = t0 + w[:, 4].astype(np.uint64)
    iters, lane_blocks, fast = w[:, 5].astype(np.float64), w[:, 6].astype(np.float64), w[:, 7].astype(np.float64)
    span_us = float(t1.max() - t0.min()) / 100.0
    dur_us = (t1 - t0).astype(np.float64) / 100.0
    cu_key = (xcc.astype(np.int64) << 12) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu
    simd_key = (cu_key << 2) | simd
    _, cu_inv, cu_cnt = np.unique(cu_key, return_inverse=True, return_counts=True)
    _, sd_inv, sd_cnt = np.unique(simd_key, return_inverse=True, return_counts=True)
    waves_on_my_cu, waves_on_my_simd = cu_cnt[cu_inv], sd_cnt[sd_inv]
    us_per_iter = dur_us / np.maximum(iters, 1)
    out = {
        "waves": int(w.shape[0]), "coop": rec["coop"], "strings": rec["n"],
        "span_ms": round(span_us / 1e3, 3),
        "start_skew_us_p50_max": [round(float(np.percentile((t0 - t0.min()).astype(np.float64) / 100.0, 50)), 1),
                                   round(float((t0.max() - t0.min())) / 100.0, 1)],
        "end_skew_us_p50_max": [round(float(np.percentile((t1.max() - t1).astype(np.float64) / 100.0, 50)), 1),
                                 round(float((t1.max() - t1.min())) / 100.0, 1)],
        "us_per_iteration_p5_p50_p95": [round(float(np.percentile(us_per_iter, q)), 3) for q in (5, 50, 95)],
        "lane_utilisation": round(float(lane_blocks.sum() / (64.0 * iters.sum())), 4),
        "cus_used": int(cu_cnt.size),
        "cus_by_waves_held": {int(k): int((cu_cnt == k).sum()) for k in np.unique(cu_cnt)},
        "simds_by_waves_held": {int(k): int((sd_cnt == k).sum()) for k in np.unique(sd_cnt)},
        "us_per_iteration_by_waves_on_simd": {int(k): round(float(np.median(us_per_iter[waves_on_my_simd == k])), 3)
                                               for k in np.unique(waves_on_my_simd)},
        "us_per_iteration_by_waves_on_cu": {int(k): round(float(np.median(us_per_iter[waves_on_my_cu == k])), 3)
                                             for k in np.unique(waves_on_my_cu)},
        "all_lanes_mid_string_iteration_share": round(float(fast.sum() / iters.sum()), 4),
        "all_lanes_mid_string_iteration_share_by_role": {int(k): round(float(fast[role == k].sum() / iters[role == k].sum()), 4) for k in np.unique(role)},
        "loop_top_wait_share_of_wave_time_by_role": {int(k): round(float(wait_ticks[role == k].sum() / (dur_us[role == k].sum() * 100.0)), 4)
                                                     for k in np.unique(role)},
        "waves_by_role": {int(k): int((role == k).sum()) for k in np.unique(role)},
        "us_per_iteration_by_role": {int(k): round(float(np.median(us_per_iter[role == k])), 3) for k in np.unique(role)},
        "lane_blocks_share_by_role": {int(k): round(float(lane_blocks[role == k].sum() / lane_blocks.sum()), 3) for k in np.unique(role)},
        "end_before_last_us_p50_by_role": {int(k): round(float(np.median((t1.max() - t1[role == k]).astype(np.float64)) / 100.0), 1)
                                           for k in np.unique(role)},
        "us_per_iteration_by_xcc": {int(k): round(float(np.median(us_per_iter[xcc == k])), 3) for k in np.unique(xcc)},
        "waves_by_xcc": {int(k): int((xcc == k).sum()) for k in np.unique(xcc)},
    }
    return out


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        for r in read_records(sys.argv[2]):
            print(json.dumps(analyse(r)))
        return
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    path = os.path.join(tempfile.mkdtemp(), "waves.bin")
    os.environ["MI_SHA_WAVE_STATS"] = path
    import makisu_amd as M
    from makisu_amd import workloads as W
    sh = W.c2(0, 1)
    ms = []
    with M.Engine() as e:
        b = e.batch(sh.n_files, sh.n_bytes)
        b.add_synthetic(sh.sizes, sh.cids, seed=sh.seed)
        b.run()
        for _ in range(launches):
            b.rerun()
            ms.append(round(e.stats()["ms_sha_chunks"], 3))
        b.free()
    recs = read_records(path)
    print(json.dumps({"pid": os.getpid(), "sha_ms_by_launch_incl_first": ms, "records": len(recs)}))
    for r in recs[-launches:]:
        print(json.dumps(analyse(r)))


if __name__ == "__main__":
    main()
