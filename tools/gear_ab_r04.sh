# (round 6: the -D knobs these variants use live in tools/experiments/gear_cdc_experiments.patch -- apply it to a copy of the tree first)
out=gpurun_out/r04_gear_ab.txt
export TMPDIR=/tmp
summ() { db=$(find $1 -name "*_results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py $db; }
{
echo "# Round 4, Gear marking: what could a scheme WITHOUT the warm-up cache line gain, and what may it cost?  Same box, C2 serial steps"
echo "# (tools/quick_bench.py --steps 20; 'cdc' of the last step).  Variants of csrc/gear_cdc.hip (tools/build_variants.sh):"
echo "#   default     the shipped kernel: 64-byte warm-up read in front of every 1 KiB lane run (a second cache line per run)"
echo "#   nowarm      MEASUREMENT ONLY, wrong cuts: the warm-up bytes taken from the lane's own first line -- same instructions, same"
echo "#               registers, no extra line: the upper bound of what getting the neighbour's tail for free could save"
echo "#   wg384       the shipped kernel in 384-thread workgroups: 12 waves per CU = 3 per SIMD, the occupancy a kernel with 16 more"
echo "#               VGPRs (132 instead of 116: the lane's own first 64 bytes kept for the end of the run) would have"
echo "#   nowarm384   both: the best case of the DPP scheme (no extra line, three waves per SIMD)"
for v in default nowarm wg384 nowarm384 default nowarm384; do
  lib=$PWD/tools/bin/libmi_$v.so; [ $v = default ] && lib=$PWD/makisu_amd/libmakisu_mi.so
  echo -n "$v: "; MAKISU_MI_LIB=$lib timeout 100 python tools/quick_bench.py --steps 20 2>&1 | grep inflight | tail -1 | sed 's/.*last step: //'
done
for v in default nowarm; do
  lib=$PWD/tools/bin/libmi_$v.so; [ $v = default ] && lib=$PWD/makisu_amd/libmakisu_mi.so
  rm -rf gpurun_out/gear_pmc
  MAKISU_MI_LIB=$lib timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/gear_pmc -o p -- python tools/quick_bench.py --steps 3 > gpurun_out/gear_pmc.log 2>&1
  echo "## $v: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/quick_bench.py --steps 3 (KiB per dispatch, x2 on gfx950)"
  summ gpurun_out/gear_pmc | grep -E "gear_cdc_small_fast_kernel" | head -3
done
rm -rf gpurun_out/gear_pmc
} > $out 2>&1
cat $out
