#!/bin/bash
# after the tree's per-depth places and the trust predicate's (round 5, last session): the GPU commit tests, the host rows, the commit tables
mkdir -p gpurun_out/places
o=gpurun_out/places
timeout 300 python -m pytest tests/test_gpu_commit.py -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/gpu_commit_tests.txt; cat $o/gpu_commit_tests.txt
timeout 200 tools/r05_host_rows.sh > $o/r05_host_rows.txt 2>&1; grep -E "merge|scan" $o/r05_host_rows.txt
timeout 400 python tools/host_scale_bench.py 10000000 > $o/r05_host_scale_py.txt 2>&1; cat $o/r05_host_scale_py.txt
timeout 200 python tools/commit_layer_bench.py 100000 4096 > $o/r05_commit_e2e_100k.txt 2>/dev/null
timeout 300 python tools/commit_layer_bench.py 1000000 4096 > $o/r05_commit_e2e_1m.txt 2>/dev/null
grep -A3 -E "nothing changed|rewritten" $o/r05_commit_e2e_100k.txt $o/r05_commit_e2e_1m.txt | cut -c1-150
