#!/bin/bash
# round 6, GPU call 31: every batch on an arena of pieces hashed with the cooperative loads (MI_SHA_COOP_MIN_GIB_PIECES=0: the library's
# choice from 1 GiB on, here from the first byte) -- the gpu suite, the commit soak in six modes, the split soak
mkdir -p gpurun_out/c31
export MI_SHA_COOP_MIN_GIB_PIECES=0
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/c31/r06_gputests_coop_on_every_piecewise_arena.txt
cat gpurun_out/c31/r06_gputests_coop_on_every_piecewise_arena.txt
{
timeout 200 python tools/commit_soak.py 1700 40
MI_COMMIT_PIPELINE=0 timeout 200 python tools/commit_soak.py 1740 40
MI_SOAK_TRUST=1 timeout 200 python tools/commit_soak.py 1780 40
MI_COMMIT_FORCE_WINDOWS=1 MI_COMMIT_WINDOW_MB=1 timeout 200 python tools/commit_soak.py 1820 40
MI_SOAK_N_CTXS=2 timeout 200 python tools/commit_soak.py 1900 20
MI_SOAK_N_CTXS=8 MI_SOAK_TRUST=1 timeout 300 python tools/commit_soak.py 1920 20
} 2>&1 | grep -v "^$" | tail -30 > gpurun_out/c31/r06_commit_soak_coop.txt
cat gpurun_out/c31/r06_commit_soak_coop.txt
timeout 600 python tools/split_soak.py 2000 40 > gpurun_out/c31/r06_split_soak_coop.txt 2>&1
tail -4 gpurun_out/c31/r06_split_soak_coop.txt
