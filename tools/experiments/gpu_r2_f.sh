#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
O=gpurun_out/r2f
run() { # label, env..., args
  label=$1; shift
  env "$@" > /dev/null 2>&1
}
for bpc in 2 3; do for inf in 1 2 3; do
  MI_SHA_BLOCKS_PER_CU=$bpc timeout 120 python bench.py --no-cpu-baseline --inflight $inf --steps 40 > $O/b_${bpc}_${inf}.json 2>/dev/null
  python - <<PY
import json
j=json.load(open('$O/b_${bpc}_${inf}.json'))
print('bpc $bpc inflight $inf:', j['value'], 'GiB/s', j['ms_per_step'], 'ms/step sha', j['roofline']['avg_launch_ms'])
PY
done; done
# clocks and power while the GPU runs serial steps, then two batches in flight
( for i in $(seq 1 120); do date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power"; sleep 0.1; done ) > $O/smi_serial.log 2>&1 &
SM=$!
timeout 120 python bench.py --no-cpu-baseline --inflight 1 --steps 1500 > $O/long_serial.json 2>/dev/null
kill $SM 2>/dev/null; wait $SM 2>/dev/null
( for i in $(seq 1 120); do date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power"; sleep 0.1; done ) > $O/smi_overlap.log 2>&1 &
SM=$!
timeout 120 python bench.py --no-cpu-baseline --inflight 2 --steps 1500 > $O/long_overlap.json 2>/dev/null
kill $SM 2>/dev/null; wait $SM 2>/dev/null
grep -c sclk $O/smi_serial.log; grep -E "sclk|ower" $O/smi_serial.log | sort | uniq -c | sort -rn | head -12
echo ---; grep -E "sclk|ower" $O/smi_overlap.log | sort | uniq -c | sort -rn | head -12
python - <<PY
import json
for n in ('long_serial','long_overlap'):
    j=json.load(open('$O/%s.json'%n)); print(n, j['value'], j['ms_per_step'])
PY
