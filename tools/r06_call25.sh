#!/bin/bash
# round 6, call 25: the commit soak again on the round's FINAL code (split files, both exchange forms, the prefetch fix), new seeds,
# eight modes: pipelined, phase by phase, TRUST_CTIME, forced windows, both, 2 ctxs, 3 ctxs phase by phase, 8 ctxs + TRUST_CTIME
mkdir -p gpurun_out/call25
{
timeout 400 python tools/commit_soak.py 3000 60
MI_COMMIT_PIPELINE=0 timeout 400 python tools/commit_soak.py 3100 60
MI_SOAK_TRUST=1 timeout 400 python tools/commit_soak.py 3200 60
MI_COMMIT_FORCE_WINDOWS=1 MI_COMMIT_WINDOW_MB=1 timeout 400 python tools/commit_soak.py 3300 60
MI_SOAK_TRUST=1 MI_COMMIT_FORCE_WINDOWS=1 MI_COMMIT_WINDOW_MB=1 timeout 400 python tools/commit_soak.py 3400 60
MI_SOAK_N_CTXS=2 timeout 400 python tools/commit_soak.py 3500 40
MI_SOAK_N_CTXS=3 MI_COMMIT_PIPELINE=0 timeout 400 python tools/commit_soak.py 3600 40
MI_SOAK_N_CTXS=8 MI_SOAK_TRUST=1 timeout 600 python tools/commit_soak.py 3700 40
} 2>&1 | grep -v "^$" | tail -40 > gpurun_out/call25/r06_commit_soak_final_code.txt
cat gpurun_out/call25/r06_commit_soak_final_code.txt
