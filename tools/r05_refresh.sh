#!/bin/bash
# after the last host changes of round 5: the bench line, the commit tables (both trees, pipelined and phase by phase, and 10^6 files), the GPU commit tests
o=gpurun_out/refresh; mkdir -p $o
timeout 200 python bench.py 2> $o/bench.err | grep "^{" | tail -1 > $o/r05_bench_n1.json; tail -c 200 $o/r05_bench_n1.json; echo
timeout 200 python -m pytest tests/test_gpu_commit.py -q 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $o/gpu_commit_tests.txt
{
  echo "# tools/commit_layer_bench.py on the MI355X box (page-cache files in /dev/shm, gzip leg off, a fresh ctx per run: the first commit of a"
  echo "# run pays the ctx's first use -- reader threads, 68 ms per GiB of fresh device memory; bench.py's commit_e2e runs on a warm ctx)"
  for args in "100000 4096" "48 134217728"; do
    echo "## pipelined (default): $args"; timeout 200 python tools/commit_layer_bench.py $args 2>&1 | tail -10
    echo "## MI_COMMIT_PIPELINE=0 (one phase after the other): $args"; MI_COMMIT_PIPELINE=0 timeout 200 python tools/commit_layer_bench.py $args 2>&1 | tail -10
  done
} > $o/r05_commit_e2e.txt 2>&1
{ echo "# tools/commit_layer_bench.py 1000000 4096 on the MI355X box: the commit table at ten times the bench line's file count (a fresh ctx)"; timeout 300 python tools/commit_layer_bench.py 1000000 4096 2>/dev/null; } > $o/r05_commit_e2e_1m.txt
tail -4 $o/r05_commit_e2e_1m.txt | cut -c1-140
