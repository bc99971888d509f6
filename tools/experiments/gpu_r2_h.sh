#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
O=gpurun_out/r2h
for rep in 1 2; do
for v in default geartree; do
  if [ $v = default ]; then L=""; else L="--lib tools/bin/libmi_$v.so"; fi
  echo "== $v serial"; timeout 120 python tools/quick_bench.py $L --steps 20 --inflight 1 2>&1 | grep inflight | tail -1
  echo "== $v inflight2"; timeout 120 python tools/quick_bench.py $L --steps 40 --inflight 2 2>&1 | grep inflight | tail -1
done; done
timeout 300 python -X faulthandler bench.py --config c2 --force-exchange --exchange torch --no-cpu-baseline --steps 3 > $O/x_torch.json 2> $O/x_torch.err; echo "torch rc=$?"; tail -25 $O/x_torch.err
timeout 300 python -X faulthandler bench.py --config c2 --force-exchange --exchange native --no-cpu-baseline --steps 3 > $O/x_native.json 2> $O/x_native.err; echo "native rc=$?"; tail -25 $O/x_native.err
timeout 600 python bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r2h/bench_c3.json'))
print('c3', j['value'], j['ms_per_step'], {k:v for k,v in j['config'].items() if 'host' in k})
PY
