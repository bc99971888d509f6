#!/bin/bash
# round 6, GPU call 36: the gzip leg at the reference's default level with and without the probe that stores blocks which will not compress --
# the image's /usr (what a real layer is made of) and the synthetic 48 x 128 MiB tree (random bytes: all of it incompressible)
mkdir -p gpurun_out/c36
MI_REAL_WARM=1 MI_REAL_GZIP=-1 MI_GZIP_PROBE=0 timeout 2400 python tools/real_tree_commit.py /usr 100 > gpurun_out/c36/r06_real_tree_gzip_probe_off.txt 2>&1
tail -12 gpurun_out/c36/r06_real_tree_gzip_probe_off.txt | cut -c1-230
MI_REAL_GZIP=-1 timeout 2400 python tools/real_tree_commit.py /usr 100 > gpurun_out/c36/r06_real_tree_gzip_probe_on.txt 2>&1
tail -12 gpurun_out/c36/r06_real_tree_gzip_probe_on.txt | cut -c1-230
(MI_BENCH_GZIP=-1 timeout 900 python tools/commit_layer_bench.py 48 134217728 2>&1 | grep -E "^  all new" -A3) > gpurun_out/c36/r06_commit_gzip_default_probe_on.txt 2>&1
cut -c1-200 gpurun_out/c36/r06_commit_gzip_default_probe_on.txt
