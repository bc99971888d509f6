#!/bin/bash
# round 6, GPU call 30: on the library whose walk-fed batches hash with the cooperative loads -- smoke(), the bench line as the driver
# runs it (its commit legs are walk-fed: 6.4 GB of pieces), the real 15 GB tree by four handles with the oracle's roots
mkdir -p gpurun_out/c30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c30/r06_smoke.txt 2>&1; tail -2 gpurun_out/c30/r06_smoke.txt
timeout 900 python bench.py > gpurun_out/c30/r06_bench_n1_call30.json 2> gpurun_out/c30/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c30/r06_bench_n1_call30.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"].get("with_rows_ratio"))
print(json.dumps(d["cpu_baseline"].get("commit_s")))
PY
MI_REAL_WARM=1 MI_REAL_N_CTXS=2 timeout 2400 python tools/real_tree_commit.py /usr 300 > gpurun_out/c30/r06_real_tree_commit_call30.txt 2>&1
tail -16 gpurun_out/c30/r06_real_tree_commit_call30.txt | cut -c1-260
