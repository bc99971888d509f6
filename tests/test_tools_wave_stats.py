"""tools/sha_wave_stats.py reads the per-wave records the chunk pass writes under MI_SHA_WAVE_STATS (csrc/sha256.hip:
a 4-word header {grid, waves per workgroup, coop, n}, then 8 words per wave).  No GPU here: a synthetic file in that
layout, two launches, and the numbers the analysis must get out of it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _record(grid, coop, n, t_base, slow_second_wave):
    """grid workgroups of 4 waves; workgroup g on XCC g % 8, SE 0, CU g // 8, waves on SIMD 0..3; on every SIMD the
    first workgroup's wave is role 0, a second one (same CU) role 1 and three times slower."""
    words = [grid, 4, coop, n]
    seen = {}
    for g in range(grid):
        xcc, cu = g % 8, (g // 8) % 16
        for w in range(4):
            key = (xcc, cu, w)
            role = seen.get(key, 0)
            seen[key] = role + 1
            hw = (w << 4) | (cu << 8)                       # SIMD 5:4, CU 11:8
            iters = 300 if role and slow_second_wave else 900
            dur = 420000                                    # 4.2 ms in 100 MHz ticks
            words += [hw, xcc | (role << 8), (t_base + g) & 0xFFFFFFFF, 5000, dur - g, iters, iters * 63, iters * 9 // 10]
    return np.array(words, dtype=np.uint32)


def test_wave_stats_reader_and_analysis(tmp_path):
    import sha_wave_stats as WS
    path = tmp_path / "waves.bin"
    a = _record(256, 0, 700000, 1000, True)                 # 2 workgroups per CU on 128 CUs
    b = _record(64, 1, 1234, 0xFFFFFFE0, False)             # the clock's low word wraps inside the launch
    np.concatenate([a, b]).tofile(path)
    recs = WS.read_records(str(path))
    assert [r["grid"] for r in recs] == [256, 64] and [r["coop"] for r in recs] == [0, 1]
    assert recs[0]["w"].shape == (1024, 8) and recs[1]["n"] == 1234
    x = WS.analyse(recs[0])
    assert x["waves"] == 1024 and x["cus_used"] == 128 and x["cus_by_waves_held"] == {8: 128}
    assert x["simds_by_waves_held"] == {2: 512} and x["waves_by_role"] == {0: 512, 1: 512}
    # 900 iterations against 300 in the same 4.2 ms: 4.67 us against 14 us per iteration, three quarters of the work
    assert abs(x["us_per_iteration_by_role"][0] - 4.667) < 0.01 and abs(x["us_per_iteration_by_role"][1] - 14.0) < 0.01
    assert abs(x["lane_blocks_share_by_role"][0] - 0.75) < 1e-3
    assert abs(x["lane_utilisation"] - 63 / 64) < 1e-3 and abs(x["all_lanes_mid_string_iteration_share"] - 0.9) < 1e-3
    assert 4.19 < x["span_ms"] < 4.21 and abs(x["loop_top_wait_share_of_wave_time_by_role"][0] - 5000 / 420000) < 1e-3
    y = WS.analyse(recs[1])
    assert y["waves"] == 256 and y["waves_by_role"] == {0: 256} and 4.19 < y["span_ms"] < 4.21
