#!/bin/bash
# the round's last GPU call: the whole GPU suite on the final code, then the commit soak in its four modes
mkdir -p gpurun_out/last
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -4 > gpurun_out/last/r05_gputests_final.txt
cat gpurun_out/last/r05_gputests_final.txt
{
timeout 200 python tools/commit_soak.py 500 40
MI_COMMIT_PIPELINE=0 timeout 200 python tools/commit_soak.py 540 20
MI_SOAK_TRUST=1 timeout 200 python tools/commit_soak.py 560 40
MI_COMMIT_FORCE_WINDOWS=1 MI_COMMIT_WINDOW_MB=1 timeout 200 python tools/commit_soak.py 600 30
MI_SOAK_TRUST=1 MI_COMMIT_FORCE_WINDOWS=1 MI_COMMIT_WINDOW_MB=1 timeout 200 python tools/commit_soak.py 630 20
} 2>&1 | grep -v "^$" | tail -30 > gpurun_out/last/r05_commit_soak_modes.txt
cat gpurun_out/last/r05_commit_soak_modes.txt
