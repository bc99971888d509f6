#!/bin/bash
# (round 6: the -D knobs these variants use live in tools/experiments/gear_cdc_experiments.patch -- apply it to a copy of the tree first)
# Round 4, second look at marking-under-hashing: is the shared resource the address translation (the lane-owned hashing
# loads touch 64 pages per instruction)?  The cooperative hashing loads ask 15x less of it; at 161 VGPRs two of its waves
# leave room for ONE marking wave per SIMD (256-thread marking workgroups, -DMI_GEAR_FAST_COPIES=16).
out=gpurun_out/overlap_ab2
mkdir -p $out; : > $out/log.txt
run() { label=$1; lib=$2; inflight=$3; shift 3
    echo "== $label" >> $out/log.txt
    env "$@" timeout 60 python tools/quick_bench.py --steps 24 --inflight $inflight ${lib:+--lib $lib} 2>&1 | grep "^inflight" | tail -2 >> $out/log.txt; }
C16=tools/bin/libmi_gprio3c16.so; C16P0=tools/bin/libmi_gc16.so
COOP="MI_SHA_COOP_MIN_GIB=0 MI_SHA_COOP_BLOCKS_PER_CU=2"
run "512-thread marking, cooperative hashing 2/CU, one at a time"   ""     1 $COOP
run "512-thread marking, cooperative hashing 2/CU, 2 in flight"      ""     2 $COOP
run "256-thread marking prio 0, cooperative hashing 2/CU, one at a time" $C16P0 1 $COOP
run "256-thread marking prio 0, cooperative hashing 2/CU, 2 in flight"   $C16P0 2 $COOP
run "256-thread marking prio 3, cooperative hashing 2/CU, 2 in flight"   $C16   2 $COOP
run "256-thread marking prio 3, cooperative hashing 2/CU, 3 in flight"   $C16   3 $COOP
run "256-thread marking prio 0, lane-owned hashing, one at a time"       $C16P0 1 X=0
run "256-thread marking prio 0, lane-owned hashing, 2 in flight"         $C16P0 2 X=0
run "512-thread marking, lane-owned hashing, 2 in flight (reference)"    ""     2 X=0
cat $out/log.txt
