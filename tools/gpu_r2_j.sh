#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2j
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2j/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2j/pytest.log
tail -4 gpurun_out/r2j/pytest.log
bash tools/round_profiles.sh r02 > gpurun_out/r2j/round.log 2>&1
tail -30 gpurun_out/r2j/round.log
python - <<'PY'
import json
for n in ('n1','c3','c5','c2_native_exchange','c4_one_shard'):
    try:
        txt=[l for l in open('gpurun_out/round/r02_bench_%s.json'%n) if l.startswith('{')][-1]
        j=json.loads(txt)
        print(n, j['value'], j['ms_per_step'], j['roofline']['frac'], {k:v for k,v in j['config'].items() if 'host' in k and 'sample' not in k}, j.get('dedup_check'), (j.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(n, 'failed', e)
PY
cat gpurun_out/round/traffic_latest.json; cat gpurun_out/round/r02_large_files.txt; cat gpurun_out/round/r02_sha_schemes_32gb.txt
