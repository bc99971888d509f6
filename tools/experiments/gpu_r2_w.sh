#!/bin/bash
# round 2, call w: cooperative SHA loads -- full GPU suite with the scheme forced, then the crossover
mkdir -p gpurun_out/r2w
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r2w
MI_SHA_COOP_MIN_GIB=0 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $out/pytest_coop.txt
for files in 100000 200000 400000 800000; do
  for g in 1000 0; do
    MI_SHA_COOP_MIN_GIB=$g python tools/quick_bench.py --files $files --size 65536 --steps 6 2>&1 | tail -1 | sed "s/^/files $files coop_min $g /" | tee -a $out/cross.txt
  done
done
for g in 1000 0; do
  MI_SHA_COOP_MIN_GIB=$g python tools/quick_bench.py --files 200000 --size 65536 --steps 10 --inflight 2 2>&1 | tail -1 | sed "s/^/files 200000 coop_min $g /" | tee -a $out/cross.txt
  MI_SHA_COOP_MIN_GIB=$g python tools/quick_bench.py --files 100000 --size 65536 --steps 20 --inflight 2 2>&1 | tail -1 | sed "s/^/files 100000 coop_min $g /" | tee -a $out/cross.txt
done
