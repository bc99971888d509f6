#!/bin/bash
# Round 5, the end-of-round GPU call: everything the round's profiles/ and tables come from, on one box.
#   gpurun --timeout 2400 -- 'tools/round5_final_call.sh'
#   1. tools/gpu_round_check.sh          the GPU suite, smoke, both N > 1 launch forms on the RCCL double, the forced exchange  (~6 min)
#   2. PARTS=core tools/round_profiles.sh r05   the bench line, kernel traces, FETCH / WRITE, SQ counters, C3 / C4 / C5 lines      (~7 min)
#   3. the commit tables: tools/commit_layer_bench.py (two trees, pipelined and phase by phase), tools/one_read_vs_two.py,
#      tools/first_use_probe.py, tools/r05_host_rows.sh                                                                            (~3 min)
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/final
mkdir -p $out
echo "== 1. round check"; timeout 900 tools/gpu_round_check.sh 2>&1 | tee $out/round_check.txt | tail -25
echo "== 2. core profiles"; PARTS=core timeout 900 tools/round_profiles.sh r05 2>&1 | tee $out/round_profiles.txt | tail -12
echo "== 3. commit tables"
{
  echo "# tools/commit_layer_bench.py on the MI355X box (page-cache files in /dev/shm, gzip leg off, a fresh ctx per run: the first commit of a"
  echo "# run pays the ctx's first use -- reader threads, 68 ms per GiB of fresh device memory; bench.py's commit_e2e runs on a warm ctx)"
  for args in "100000 4096" "48 134217728"; do
    echo "## pipelined (default): $args"; timeout 200 python tools/commit_layer_bench.py $args 2>&1 | tail -10
    echo "## MI_COMMIT_PIPELINE=0 (one phase after the other): $args"; MI_COMMIT_PIPELINE=0 timeout 200 python tools/commit_layer_bench.py $args 2>&1 | tail -10
  done
} > $out/r05_commit_e2e.txt 2>&1
tail -12 $out/r05_commit_e2e.txt
timeout 300 python tools/one_read_vs_two.py /tmp > $out/r05_one_read_vs_two.txt 2>&1; tail -10 $out/r05_one_read_vs_two.txt
timeout 200 python tools/first_use_probe.py > $out/r05_first_use_probe.txt 2>&1; tail -9 $out/r05_first_use_probe.txt
timeout 300 tools/r05_host_rows.sh > $out/r05_host_rows.txt 2>&1; tail -8 $out/r05_host_rows.txt
timeout 400 python tools/host_scale_bench.py 10000000 > $out/r05_host_scale_1e7.txt 2>&1; tail -10 $out/r05_host_scale_1e7.txt
