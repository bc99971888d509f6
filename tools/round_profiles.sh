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
for f in 4:4294967296 1:17179869184; do
  timeout 100 python tools/quick_bench.py --files ${f%%:*} --size ${f##*:} --steps 3 2>&1 | grep inflight | tail -1
done > $out/${tag}_large_files.txt
ls -la $out
