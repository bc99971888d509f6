#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
O=gpurun_out/r2d
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 600 python bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r2d/bench_c3.json'))
print(j['value'], j['ms_per_step'], {k:v for k,v in j['config'].items() if 'host' in k})
PY
