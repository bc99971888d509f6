#!/bin/bash
# Regenerates everything under profiles/ that comes from a GPU run (run on the MI355X box from the
# repo root; results land in gpurun_out/round/, copy what is to be judged into profiles/).
#   tools/round_profiles.sh r01
set -u
tag=${1:-rXX}
out=gpurun_out/round
mkdir -p $out
export TMPDIR=/tmp
T="timeout 170"
$T python bench.py 2> $out/bench.err | grep "^{" | tail -1 > $out/${tag}_bench_n1.json
tail -c 400 $out/${tag}_bench_n1.json; echo
# same command as the bench (batches in flight) and the serial form
$T rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/kt.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $out/kts -o kts -- python bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline > $out/kts.log 2>&1
# HBM traffic: one --pmc pass per counter, kernel trace only
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o fetch -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > $out/fetch.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o write -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > $out/write.log 2>&1
for n in kt kts fetch write; do
  db=$(find $out/$n -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/prof_summary.py $db > $out/${tag}_$n.txt 2>&1
done
python tools/make_traffic_json.py $tag $out > $out/traffic_latest.json 2> $out/traffic.err || cat $out/traffic.err
# the other BASELINE configs (bench lines only) and the host-fed leg
$T python bench.py --config c3 --no-cpu-baseline > $out/${tag}_bench_c3.json 2> $out/bench_c3.err
$T python bench.py --config c5 --no-cpu-baseline > $out/${tag}_bench_c5.json 2> $out/bench_c5.err
$T python bench.py --config c2 --force-exchange --exchange native --no-cpu-baseline > $out/${tag}_bench_c2_native_exchange.json 2> $out/bench_x.err
# one rank's C4 shard (1.25 M files, 76 GiB) on one GPU
$T python bench.py --config c4 --no-cpu-baseline --steps 4 --warmup 1 > $out/${tag}_bench_c4_one_shard.json 2> $out/bench_c4.err
{
for f in 4:4294967296 1:17179869184; do
  timeout 100 python tools/quick_bench.py --files ${f%%:*} --size ${f##*:} --steps 3 2>&1 | grep inflight | tail -1
done
echo "# two batches in flight"
timeout 100 python tools/quick_bench.py --files 4 --size 4294967296 --steps 4 --inflight 2 2>&1 | grep inflight | tail -1
} > $out/${tag}_large_files.txt
# the chunk pass's two load schemes on a 32 GB arena: kernel time + per-CU TLB counters
for g in 1000 0; do
  MI_SHA_COOP_MIN_GIB=$g $T rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --kernel-trace -d $out/utcl_$g -o p -- python tools/quick_bench.py --files 240 --size 134217728 --steps 2 > $out/utcl_$g.log 2>&1
  db=$(find $out/utcl_$g -name "*_results.db" | head -1)
  echo "## MI_SHA_COOP_MIN_GIB=$g (1000 = lane-owned byte-aligned loads, 0 = quad-cooperative), 240 x 128 MiB"
  [ -n "$db" ] && python tools/prof_summary.py $db 2>&1 | grep "sha256_items_kernel<0"
  rm -rf $out/utcl_$g
  MI_SHA_COOP_MIN_GIB=$g timeout 100 python tools/quick_bench.py --files 240 --size 134217728 --steps 4 2>&1 | grep inflight | tail -1
  MI_SHA_COOP_MIN_GIB=$g $T rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fs_$g -o p -- python tools/quick_bench.py --files 240 --size 134217728 --steps 2 > $out/fs_$g.log 2>&1
  db=$(find $out/fs_$g -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/prof_summary.py $db 2>&1 | grep "sha256_items_kernel<0.*FETCH_SIZE"
  rm -rf $out/fs_$g
done > $out/${tag}_sha_schemes_32gb.txt
ls -la $out
