#!/bin/bash
# round 6, GPU call 32: the push-digest test with its self-calibrated limit (its child's line), then the default bench with its wall time
mkdir -p gpurun_out/c32
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "long" -s 2>&1 | grep -E "eight 128|passed|failed" | tail -5) > gpurun_out/c32/r06_long_strings.txt
cat gpurun_out/c32/r06_long_strings.txt
( time timeout 900 python bench.py > gpurun_out/c32/r06_bench_n1_call32.json 2> gpurun_out/c32/bench.err ) 2> gpurun_out/c32/bench_time.txt
cat gpurun_out/c32/bench_time.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c32/r06_bench_n1_call32.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("bench_wall_s"))
PY
