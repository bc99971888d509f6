#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
O=gpurun_out/r2c
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for t in 8 16; do MI_STAGE_THREADS=$t timeout 300 python tools/host_feed_bench.py 32 128 >> $O/host_feed.log 2>&1; done
grep threads $O/host_feed.log
MI_STAGE_THREADS=16 MI_FEED_MODES=add_path,add_path,add_path timeout 300 python tools/host_feed_bench.py 2000 1 >> $O/host_feed_small.log 2>&1
grep threads $O/host_feed_small.log
