#!/bin/bash
mkdir -p gpurun_out/c14
(timeout 1200 python -m pytest tests/test_gpu_commit.py -m gpu -q -k "several_ctxs or 600_mib" 2>&1 | tail -30) > gpurun_out/c14/split.txt
tail -30 gpurun_out/c14/split.txt | cut -c1-500
