#!/bin/bash
# round 6, the final GPU call: the gpu suite, smoke(), the bench line as the driver runs it, the fault soak on real DMA
mkdir -p gpurun_out/final
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/final/r06_gputests_final.txt
cat gpurun_out/final/r06_gputests_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/r06_smoke.txt 2>&1; tail -2 gpurun_out/final/r06_smoke.txt
timeout 900 python bench.py > gpurun_out/final/r06_bench_n1_final_code.json 2> gpurun_out/final/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final/r06_bench_n1_final_code.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"].get("with_rows_ratio"))
print(json.dumps(d["cpu_baseline"].get("commit_s")))
print(d.get("commit_e2e", {}).get("first_use_s"))
PY
(timeout 600 python tools/verify_fault_soak.py 16; MI_COMMIT_PIPELINE=0 timeout 600 python tools/verify_fault_soak.py 8) > gpurun_out/final/r06_verify_fault_soak.txt 2>&1
tail -14 gpurun_out/final/r06_verify_fault_soak.txt
