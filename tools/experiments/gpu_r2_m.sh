#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2m; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 200 python tools/quick_bench.py --files 4 --size 4294967296 --steps 3 2>&1 | grep inflight | tail -1
timeout 200 python tools/quick_bench.py --files 1 --size 17179869184 --steps 3 2>&1 | grep inflight | tail -1
timeout 300 python tools/quick_bench.py --files 500 --size 134217728 --steps 2 2>&1 | grep inflight | tail -1
for cfg in c5 c3 c2; do
  timeout 400 python bench.py --config $cfg --no-cpu-baseline --no-host-fed 2>/dev/null | grep "^{" | tail -1 > $O/$cfg.json
  python -c "
import json; j=json.load(open('$O/$cfg.json')); print('$cfg', j['value'], j['ms_per_step'], j['serial_phase_ms'], j.get('dedup_check'))"
done
