#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
O=gpurun_out/r2e
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -14 $O/pytest.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline > $O/bench_c2_$i.json 2> $O/bench_c2_$i.err; done
python - <<'PY'
import json
for i in (1,2):
    j=json.load(open('gpurun_out/r2e/bench_c2_%d.json'%i))
    print('c2', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['path_frac'], j['phase_ms_avg'], j['serial_phase_ms'])
PY
MI_STAGE_THREADS=16 MI_FEED_MODES=add_path,add_path,add_path timeout 300 python tools/host_feed_bench.py 48 128 > $O/feed_notorch.log 2>&1
MI_FEED_TORCH=1 MI_STAGE_THREADS=16 MI_FEED_MODES=add_path,add_path,add_path timeout 300 python tools/host_feed_bench.py 48 128 > $O/feed_torch.log 2>&1
grep threads $O/feed_notorch.log $O/feed_torch.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 > $GRAFT_REPO_ROOT/$O/prof_run.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats*" | head; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -30 "$f"
