#!/bin/bash
# round 6, GPU call 9: the round's profiles (core part) + the commit soak in seven modes + small-batch latency + the stage soak
mkdir -p gpurun_out/soak
PARTS=core bash tools/round_profiles.sh r06 > gpurun_out/round_profiles.log 2>&1
tail -3 gpurun_out/round_profiles.log
{
timeout 200 python tools/commit_soak.py 700 40
MI_COMMIT_PIPELINE=0 timeout 200 python tools/commit_soak.py 740 40
MI_SOAK_TRUST=1 timeout 200 python tools/commit_soak.py 780 40
MI_COMMIT_FORCE_WINDOWS=1 MI_COMMIT_WINDOW_MB=1 timeout 200 python tools/commit_soak.py 820 40
MI_SOAK_TRUST=1 MI_COMMIT_FORCE_WINDOWS=1 MI_COMMIT_WINDOW_MB=1 timeout 200 python tools/commit_soak.py 860 40
MI_SOAK_N_CTXS=2 timeout 200 python tools/commit_soak.py 900 20
MI_SOAK_N_CTXS=8 MI_SOAK_TRUST=1 timeout 300 python tools/commit_soak.py 920 20
} 2>&1 | grep -v "^$" | tail -30 > gpurun_out/soak/r06_commit_soak.txt
cat gpurun_out/soak/r06_commit_soak.txt
(timeout 120 python tools/small_batch_latency.py; echo "## MI_ARENA=malloc"; MI_ARENA=malloc timeout 120 python tools/small_batch_latency.py) > gpurun_out/soak/r06_small_batch_latency.txt 2>&1
cat gpurun_out/soak/r06_small_batch_latency.txt
timeout 400 python tools/stage_soak.py 60 > gpurun_out/soak/r06_stage_soak.txt 2>&1
tail -5 gpurun_out/soak/r06_stage_soak.txt
