#!/bin/bash
# round 2, call z: 16 GiB batch, cooperative loads with 2 or 3 workgroups per CU, and the old loads
cd /root/repo
for cfg in "1000 2" "0 2" "0 3"; do
  set -- $cfg
  MI_SHA_COOP_MIN_GIB=$1 MI_SHA_COOP_BLOCKS_PER_CU=$2 python tools/quick_bench.py --files 4 --size 4294967296 --steps 4 2>&1 | tail -1 | sed "s/^/4x4GiB coop_min $1 wg $2 /"
  MI_SHA_COOP_MIN_GIB=$1 MI_SHA_COOP_BLOCKS_PER_CU=$2 python tools/quick_bench.py --files 250000 --size 65536 --steps 4 2>&1 | tail -1 | sed "s/^/250k x 64K coop_min $1 wg $2 /"
done
