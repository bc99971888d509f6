#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2k
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2k/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2k/pytest.log
tail -3 gpurun_out/r2k/pytest.log
timeout 600 python tools/exchange_cost.py --files 1250000 --world 8 --inflight 1 --steps 2 > gpurun_out/r2k/exchange_c4.log 2>&1
grep -E "marking|step" gpurun_out/r2k/exchange_c4.log
timeout 300 python bench.py --config c4 --files 1250000 --no-cpu-baseline --steps 3 > gpurun_out/r2k/bench_c4_1gpu.json 2> gpurun_out/r2k/bench_c4.err
python - <<'PY'
import json
txt=[l for l in open('gpurun_out/r2k/bench_c4_1gpu.json') if l.startswith('{')]
if txt:
    j=json.loads(txt[-1]); print('c4 1 gpu', j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('dedup_check'), j['serial_phase_ms'])
else:
    print(open('gpurun_out/r2k/bench_c4.err').read()[-2000:])
PY
