#!/bin/bash
# (round 6: the -D knobs these variants use live in tools/experiments/gear_cdc_experiments.patch -- apply it to a copy of the tree first)
# Round 4: the candidate test of the marking as 7 v_min3_u32 + 1 v_min_u32 (depth 3) instead of hipcc's 8 v_min + 4 v_min3
# (tools/build_variants.sh gm3 "-DMI_GEAR_MIN3_TREE" makisu_amd/csrc/gear_cdc.hip).  Same box, alternating.
out=gpurun_out/gear_min3_ab
mkdir -p $out; : > $out/log.txt
for rep in 1 2 3; do
  for v in base gm3; do
    lib=""; [ $v = gm3 ] && lib="--lib tools/bin/libmi_gm3.so"
    for inflight in 1 2; do
      echo "== $v, inflight $inflight (rep $rep)" >> $out/log.txt
      timeout 60 python tools/quick_bench.py --steps 24 --inflight $inflight $lib 2>&1 | grep "^inflight" | tail -1 >> $out/log.txt
    done
  done
done
cat $out/log.txt
