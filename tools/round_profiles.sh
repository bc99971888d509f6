#!/bin/bash
# Regenerates everything under profiles/ that comes from a GPU run (run on the MI355X box from the
# repo root; results land in gpurun_out/round/, copy what is to be judged into profiles/).
#   tools/round_profiles.sh r03
set -u
tag=${1:-rXX}
out=gpurun_out/round
mkdir -p $out
export TMPDIR=/tmp
T="timeout 170"
summ() { db=$(find $1 -name "*_results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py $db; }

# ---- C2 (the config the metric is quoted on): bench line, kernel traces, HBM traffic, SQ counters ----
$T python bench.py 2> $out/bench.err | grep "^{" | tail -1 > $out/${tag}_bench_n1.json
tail -c 300 $out/${tag}_bench_n1.json; echo
# one batch at a time (every kernel has the GPU to itself: the durations `roofline` is computed from -- the bench takes them
# from its own one-at-a-time steps), and the bench's default form (two batches in flight in the timed region, then serial steps)
$T rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 40 --warmup 2 --inflight 1 --no-cpu-baseline --no-with-rows --no-commit-e2e > $out/kt.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $out/kts -o kts -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-with-rows --no-commit-e2e > $out/kts.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 40 --warmup 2 --inflight 1 --no-cpu-baseline --no-with-rows --no-commit-e2e   (one batch at a time: a launch's duration is the kernel's own -- what roofline.avg_launch_ms measures)"; summ $out/kt; } > $out/${tag}_kernel_trace_stats.txt 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-with-rows --no-commit-e2e   (the default command: 12 steps with two batches in flight -- launches overlap --, then 12 one at a time)"; summ $out/kts; } > $out/${tag}_kernel_trace_stats_default.txt 2>&1
# HBM traffic: one --pmc pass per counter, kernel trace only
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o fetch -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-with-rows --no-commit-e2e > $out/fetch.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o write -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-with-rows --no-commit-e2e > $out/write.log 2>&1
summ $out/fetch > $out/${tag}_fetch.txt 2>&1
summ $out/write > $out/${tag}_write.txt 2>&1
cp $out/${tag}_fetch.txt $out/${tag}_pmc_fetch_size.txt; cp $out/${tag}_write.txt $out/${tag}_pmc_write_size.txt
python tools/make_traffic_json.py $tag $out > $out/traffic_latest.json 2> $out/traffic.err || cat $out/traffic.err
# SQ counters of both Gear kernels and both SHA load schemes (C2 serial; the cooperative scheme forced in
# the second pair of passes).  Two passes of <= 8 counters each per scheme.
{
for scheme in 1000 0; do
  echo "## MI_SHA_COOP_MIN_GIB=$scheme (1000 = lane-owned loads, 0 = quad-cooperative loads), bench.py --inflight 1, C2"
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT"; do
    rm -rf $out/sq
    MI_SHA_COOP_MIN_GIB=$scheme $T rocprofv3 --pmc $set --kernel-trace -d $out/sq -o sq -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-with-rows --no-commit-e2e > $out/sq.log 2>&1
    summ $out/sq 2>&1 | grep -E "gear_|sha256_items_kernel<0" | grep -v "^void mi::sha256_items_kernel<0.*FETCH"
  done
done
} > $out/${tag}_sq_counters.txt

# ---- per-wave records of the chunk pass (three processes) and the optional per-file passes ----
{ for i in 1 2 3; do timeout 100 python tools/sha_wave_stats.py 3 2>/dev/null; done; } > $out/${tag}_sha_wave_stats_final.txt
{ echo "# tools/quick_bench.py --steps 10 --flags F on C2 (F = 0 none, 2 MI_FLAG_FILE_CRC32): the CRC pass = roots(2) - roots(0)";
  for f in 0 2; do echo -n "C2 flags=$f: "; timeout 100 python tools/quick_bench.py --steps 10 --flags $f 2>&1 | grep inflight | tail -1; done; } > $out/${tag}_crc_pass.txt

# ---- the other BASELINE configs: bench lines; kernel trace + traffic for C3 and C5 ----
$T python bench.py --config c3 --no-cpu-baseline --no-with-rows --no-commit-e2e > $out/${tag}_bench_c3.json 2> $out/bench_c3.err
$T python bench.py --config c5 --no-cpu-baseline --no-with-rows --no-commit-e2e > $out/${tag}_bench_c5.json 2> $out/bench_c5.err
$T python bench.py --config c5u --no-cpu-baseline --no-with-rows --no-commit-e2e > $out/${tag}_bench_c5u.json 2> $out/bench_c5u.err
$T python bench.py --config c2 --force-exchange --no-cpu-baseline --no-with-rows --no-commit-e2e > $out/${tag}_bench_c2_native_exchange.json 2> $out/bench_x.err
# one rank's C4 shard (1.25 M files, 76 GiB) on one GPU
$T python bench.py --config c4 --no-cpu-baseline --no-with-rows --no-commit-e2e --steps 4 --warmup 1 > $out/${tag}_bench_c4_one_shard.json 2> $out/bench_c4.err
for cfg in c3 c5; do
  $T rocprofv3 --kernel-trace --stats -d $out/kt_$cfg -o kt -- python bench.py --config $cfg --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-with-rows --no-commit-e2e --no-host-fed > $out/kt_$cfg.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $cfg --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-with-rows --no-commit-e2e --no-host-fed"; summ $out/kt_$cfg; } > $out/${tag}_kernel_trace_stats_$cfg.txt 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmc_$cfg
    $T rocprofv3 --pmc $ctr --kernel-trace -d $out/pmc_$cfg -o p -- python bench.py --config $cfg --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline --no-with-rows --no-commit-e2e --no-host-fed > $out/pmc_$cfg.log 2>&1
    { echo "# rocprofv3 --pmc $ctr --kernel-trace -- python bench.py --config $cfg --steps 1 --warmup 1 --inflight 1 (counter averages per dispatch, KiB; FETCH_SIZE x2 on gfx950)"; summ $out/pmc_$cfg | grep -E "$ctr" | head -12; } >> $out/${tag}_pmc_$cfg.txt 2>&1
  done
  rm -rf $out/kt_$cfg $out/pmc_$cfg
done
[ "${PARTS:-all}" = core ] && { ls -la $out; exit 0; }
{
for f in 4:4294967296 1:17179869184; do
  timeout 100 python tools/quick_bench.py --files ${f%%:*} --size ${f##*:} --steps 3 2>&1 | grep inflight | tail -1
done
echo "# two batches in flight"
timeout 100 python tools/quick_bench.py --files 4 --size 4294967296 --steps 4 --inflight 2 2>&1 | grep inflight | tail -1
} > $out/${tag}_large_files.txt

# ---- host-fed: soak with / without MI_FLAG_VERIFY_STAGING, many small files ----
timeout 400 python tools/stage_soak.py ${SOAK_ROUNDS:-100} > $out/${tag}_stage_soak.txt 2> $out/soak.err
{ for kib in 4 64; do timeout 200 python tools/many_small_files.py $((kib == 4 ? 200 : 100)) 500 $kib 2>&1 | grep -E "add_tree|add_paths"; done; echo "# MI_WALK_THREADS=1 (the sequential walker)"; MI_WALK_THREADS=1 timeout 100 python tools/many_small_files.py 200 500 4 2>&1 | grep add_tree | tail -1; } > $out/${tag}_many_small_files.txt

# ---- Gear marking variants, same box (tools/build_variants.sh built them into tools/bin/) ----
{
echo "# C2 serial steps (tools/quick_bench.py --steps 20), then 48 x 128 MiB, per variant of csrc/gear_cdc.hip:"
echo "#  default = no LDS bitmap, 16 table copies, lane-owned loads; coal64 / coal128 = the same with the coalesced fetch +"
echo "#  LDS exchange (MI_GEAR_COAL_BYTES); the round-2 kernel (LDS bitmap, 8 copies) measured 1.46-1.50 ms on C2"
for v in default coal64 coal128; do
  lib=$PWD/tools/bin/libmi_$v.so; [ $v = default ] && lib=$PWD/makisu_amd/libmakisu_mi.so
  [ -f $lib ] || continue
  for rep in 1 2; do echo -n "$v C2: "; MAKISU_MI_LIB=$lib timeout 100 python tools/quick_bench.py --steps 20 2>&1 | grep inflight | tail -1; done
  echo -n "$v 48 x 128 MiB: "; MAKISU_MI_LIB=$lib timeout 100 python tools/quick_bench.py --files 48 --size 134217728 --steps 5 2>&1 | grep inflight | tail -1
done
} > $out/${tag}_gear_ab.txt
# ---- SHA workgroup placement: serial launch times with and without the LDS pin, this box ----
{ for pin in 0 1 0 1; do echo "## MI_SHA_PIN_BLOCKS=$pin"; MI_SHA_PIN_BLOCKS=$pin timeout 120 python tools/sha_box_probe.py 30 2>/dev/null | tail -1; done; } > $out/${tag}_sha_placement.txt
ls -la $out
