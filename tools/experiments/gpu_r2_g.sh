#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
O=gpurun_out/r2g
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline > $O/bench_c2_$i.json 2> $O/bench_c2_$i.err; done
python - <<'PY'
import json
for i in (1,2):
    j=json.load(open('gpurun_out/r2g/bench_c2_%d.json'%i))
    print('c2', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['path_frac'], j['serial_phase_ms'])
PY
timeout 600 python bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
timeout 600 python bench.py --config c2 --force-exchange --exchange native --no-cpu-baseline > $O/bench_c2_native.json 2> $O/bench_c2_native.err
timeout 600 python bench.py --config c2 --force-exchange --exchange torch --no-cpu-baseline > $O/bench_c2_torchx.json 2> $O/bench_c2_torchx.err
python - <<'PY'
import json
for n in ('c3','c5','c2_native','c2_torchx'):
    try:
        j=json.load(open('gpurun_out/r2g/bench_%s.json'%n))
        print(n, j['value'], j['ms_per_step'], {k:v for k,v in j['config'].items() if 'host' in k or k=='exchange'}, j.get('dedup_check'))
    except Exception as e:
        print(n, 'failed', e); print(open('gpurun_out/r2g/bench_%s.err'%n).read()[-1500:])
PY
