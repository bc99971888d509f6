#!/bin/bash
# (round 6: the -D knobs these variants use live in tools/experiments/gear_cdc_experiments.patch -- apply it to a copy of the tree first)
# Round 4, last experiment: does the Gear marking of the NEXT batch run under the hashing of the current one when its
# waves are raised above the hashing's (s_setprio in the marking kernels, -DMI_GEAR_PRIO), and what does the step gain?
# The marking's 512-thread workgroup needs 2 x 120 VGPRs on every SIMD of a CU; two hashing waves hold 2 x 136: 512 together.
# Libraries: tools/build_variants.sh gprio3 "-DMI_GEAR_PRIO=3" makisu_amd/csrc/gear_cdc.hip  (and gprio1, gprio3c16 with
# -DMI_GEAR_FAST_COPIES=16: 256-thread workgroups).  Output: gpurun_out/overlap_ab/log.txt
out=gpurun_out/overlap_ab
mkdir -p $out
run() {   # label, lib, inflight, env...
    label=$1; lib=$2; inflight=$3; shift 3
    echo "== $label" >> $out/log.txt
    env "$@" timeout 60 python tools/quick_bench.py --steps 24 --inflight $inflight ${lib:+--lib $lib} 2>&1 | grep "^inflight" | tail -2 >> $out/log.txt
}
: > $out/log.txt
B=""; P3=tools/bin/libmi_gprio3.so; P1=tools/bin/libmi_gprio1.so; C16=tools/bin/libmi_gprio3c16.so
run "base, one at a time"                      "$B"  1 X=0
run "base, 2 in flight"                        "$B"  2 X=0
run "marking prio 3, 2 in flight"              $P3   2 X=0
run "marking prio 1, 2 in flight"              $P1   2 X=0
run "marking prio 3, 3 in flight"              $P3   3 X=0
run "marking prio 3, 2 in flight, hashing 1 workgroup/CU"  $P3 2 MI_SHA_BLOCKS_PER_CU=1
run "marking prio 3, 3 in flight, hashing 1 workgroup/CU"  $P3 3 MI_SHA_BLOCKS_PER_CU=1
run "base, 3 in flight, hashing 1 workgroup/CU"            "$B" 3 MI_SHA_BLOCKS_PER_CU=1
run "marking prio 3 + 256-thread workgroups, 2 in flight"  $C16 2 X=0
run "marking prio 3 + 256-thread workgroups, 3 in flight, hashing 1 workgroup/CU"  $C16 3 MI_SHA_BLOCKS_PER_CU=1
run "base, 2 in flight, cooperative hashing loads 1 workgroup/CU"  "$B" 2 MI_SHA_COOP_MIN_GIB=0 MI_SHA_COOP_BLOCKS_PER_CU=1
run "marking prio 3, 2 in flight, cooperative hashing loads 1 workgroup/CU"  $P3 2 MI_SHA_COOP_MIN_GIB=0 MI_SHA_COOP_BLOCKS_PER_CU=1
run "base, 2 in flight (again)"                "$B"  2 X=0
cat $out/log.txt
