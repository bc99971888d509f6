#!/bin/bash
# round 6, GPU call 37: the gpu suite as the driver runs it (-x) on the library whose gzip leg stores incompressible blocks; smoke(); the bench
mkdir -p gpurun_out/c37
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/c37/gputests_full.log 2>&1
grep -E "passed|failed|error" gpurun_out/c37/gputests_full.log | tail -3 > gpurun_out/c37/r06_gputests_head.txt
cat gpurun_out/c37/r06_gputests_head.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/c37/r06_bench_n1_head.json
python -c "
import json; d=json.load(open('gpurun_out/c37/r06_bench_n1_head.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_of_valu_roof'], d['config']['bench_wall_s'], json.dumps(d['cpu_baseline']['commit_s']['all_new_gpu_over_header_only']))"
