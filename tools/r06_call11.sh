#!/bin/bash
# round 6, GPU call 11: what the AVX2 sums buy the reader threads; the commit tables again
mkdir -p gpurun_out
(for i in 1 2; do MI_FEED_MODES=add_path,add_path,add_path timeout 300 python tools/host_feed_bench.py 48 128 | tail -2; MI_FEED_SUMS=1 MI_FEED_MODES=add_path,add_path,add_path timeout 300 python tools/host_feed_bench.py 48 128 | tail -2 | sed 's/^/sums: /'; done) > gpurun_out/r06_feed_sums_avx2.txt 2>&1
cat gpurun_out/r06_feed_sums_avx2.txt
(MI_LAYER_TIMING=1 timeout 300 python tools/commit_layer_bench.py 48 134217728 2>&1 | grep -E "mi_layer: 6442|^  all|^  noth|^    (gpu|cpu)" | head -12; timeout 300 python tools/commit_layer_bench.py 100000 4096 2>&1 | grep -E "^  |^    ") > gpurun_out/r06_commit_tables_avx2.txt 2>&1
cat gpurun_out/r06_commit_tables_avx2.txt | cut -c1-220
