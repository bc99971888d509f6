#!/bin/bash
# round 6, GPU call 13: the kernels of a COMMIT under rocprofv3 (its arena is the piecewise kind), and the commit table with the gzip leg on
mkdir -p gpurun_out/c13
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/c13/kt -o kt -- python tools/commit_layer_bench.py 48 134217728 > gpurun_out/c13/kt.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/commit_layer_bench.py 48 134217728   (nine commits of a 6.4 GB tree, six of them with a ctx: the kernels of a commit, on an arena mapped in 32 MiB pieces)"; db=$(find gpurun_out/c13/kt -name "*_results.db" | head -1); python tools/prof_summary.py $db; } > gpurun_out/c13/r06_kernel_trace_stats_commit.txt 2>&1
head -14 gpurun_out/c13/r06_kernel_trace_stats_commit.txt | cut -c1-150
rm -rf gpurun_out/c13/kt
(MI_BENCH_GZIP=-1 timeout 900 python tools/commit_layer_bench.py 48 134217728 2>&1 | grep -E "^  |^    |gzip"; MI_BENCH_GZIP=-1 timeout 600 python tools/commit_layer_bench.py 100000 4096 2>&1 | grep -E "^  all new" -A3) > gpurun_out/c13/r06_commit_gzip_default.txt 2>&1
cut -c1-210 gpurun_out/c13/r06_commit_gzip_default.txt | head -16
