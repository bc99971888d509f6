#!/bin/bash
# round-2 GPU call B: whole GPU suite (new staging, index fix) + host-fed rates + host CPU facts
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
O=gpurun_out/r2b
{ nproc; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())";
  cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null;
  lscpu | head -25; free -g; df -h /dev/shm /tmp; } > $O/host.txt 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for t in 2 4 8 16; do MI_STAGE_THREADS=$t timeout 300 python tools/host_feed_bench.py 32 128 >> $O/host_feed.log 2>&1; done
cat $O/host_feed.log | grep threads
timeout 600 python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r2b/bench_c3.json'))
print(j['value'], j['ms_per_step'], j['config'].get('host_fed_GBps'), j.get('cpu_baseline'))
PY
cat $O/host.txt | head -40
