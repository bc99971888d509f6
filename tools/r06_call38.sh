#!/bin/bash
# round 6, GPU call 38: the commit table's small tree "all new" as each side's best of three fresh handles -- the test that holds the
# summary, then the default bench twice (what the ratio does from run to run)
mkdir -p gpurun_out/c38
(timeout 900 python -m pytest tests/test_gpu_commit.py -m gpu -q -k "commit_table or bench_line" 2>&1 | grep -E "passed|failed|Error" | tail -3) > gpurun_out/c38/tests.txt; cat gpurun_out/c38/tests.txt
for i in 1 2; do
timeout 600 python bench.py 2>gpurun_out/c38/bench$i.err | tail -1 > gpurun_out/c38/r06_bench_n1_rounds_$i.json
python -c "
import json; d=json.load(open('gpurun_out/c38/r06_bench_n1_rounds_$i.json')); c=d['cpu_baseline']['commit_s']; print(d['value'], d['config']['bench_wall_s'], c['all_new'], c['all_new_gpu_over_header_only'], json.dumps(d['commit_e2e']['small_files']['commits'][0]['all_new_rounds_s']))"
done
