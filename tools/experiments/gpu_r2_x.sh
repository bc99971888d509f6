#!/bin/bash
# round 2, call x: cooperative SHA loads x workgroups per CU
mkdir -p gpurun_out/r2x
cd /root/repo
out=gpurun_out/r2x
for files in 100000 400000; do
  for cfg in "1000 2" "0 2" "0 3" "1000 3"; do
    set -- $cfg
    MI_SHA_COOP_MIN_GIB=$1 MI_SHA_BLOCKS_PER_CU=$2 python tools/quick_bench.py --files $files --size 65536 --steps 8 2>&1 | tail -1 | sed "s/^/files $files coop_min $1 wg_per_cu $2 /" | tee -a $out/occ.txt
  done
done
for cfg in "1000 2" "0 2" "0 3"; do
  set -- $cfg
  MI_SHA_COOP_MIN_GIB=$1 MI_SHA_BLOCKS_PER_CU=$2 python tools/quick_bench.py --files 100000 --size 65536 --steps 20 --inflight 2 2>&1 | tail -1 | sed "s/^/files 100000 coop_min $1 wg_per_cu $2 /" | tee -a $out/occ.txt
done
