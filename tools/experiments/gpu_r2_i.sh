#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
O=gpurun_out/r2i
M=add_path,add_path,add_path
echo "--- plain"; MI_FEED_MODES=$M timeout 300 python tools/host_feed_bench.py 48 128 2>&1 | grep -E "threads|torch"
echo "--- torch import"; MI_FEED_TORCH=1 MI_FEED_MODES=$M timeout 300 python tools/host_feed_bench.py 48 128 2>&1 | grep -E "threads|torch"
echo "--- torch cuda"; MI_FEED_TORCH=cuda MI_FEED_MODES=$M timeout 300 python tools/host_feed_bench.py 48 128 2>&1 | grep -E "threads|torch"
echo "--- torch cuda 8 threads"; MI_STAGE_THREADS=8 MI_FEED_TORCH=cuda MI_FEED_MODES=$M timeout 300 python tools/host_feed_bench.py 48 128 2>&1 | grep -E "threads|torch"
echo "--- torch cuda OMP=1"; OMP_NUM_THREADS=1 MI_FEED_TORCH=cuda MI_FEED_MODES=$M timeout 300 python tools/host_feed_bench.py 48 128 2>&1 | grep -E "threads|torch"
