#!/bin/bash
# round 6, GPU call 29: does SHA-256's round, ordered into one run of 4-pass and one of 2-pass VALU instructions, hash faster?
mkdir -p gpurun_out
timeout 300 tools/bin/ubench_sha_runs > gpurun_out/r06_ubench_sha_runs.txt 2>&1
cat gpurun_out/r06_ubench_sha_runs.txt
