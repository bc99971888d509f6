#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for w in 0 786432 393216 1572864; do
  echo "== window rows $w"
  MI_SHA_WINDOW_ROWS=$w timeout 200 python tools/quick_bench.py --files 4 --size 4294967296 --steps 3 2>&1 | grep inflight | tail -1
  MI_SHA_WINDOW_ROWS=$w timeout 300 python tools/quick_bench.py --files 500 --size 134217728 --steps 2 2>&1 | grep inflight | tail -1
  MI_SHA_WINDOW_ROWS=$w timeout 300 python tools/quick_bench.py --files 1250000 --size 65536 --steps 2 2>&1 | grep inflight | tail -1
done
for w in 0 786432; do
  MI_SHA_WINDOW_ROWS=$w timeout 300 python bench.py --config c5 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > $O/c5_$w.json
  python -c "
import json; j=json.load(open('$O/c5_$w.json')); print('c5 window $w', j['value'], j['ms_per_step'], j['serial_phase_ms'])"
done
