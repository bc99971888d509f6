#!/bin/bash
# round 6, call 19: the device work of ONE rank in either exchange form at C4's scale, alone on the GPU (one rank on the real
# RCCL, the exchange forced: with one rank every row is its own owner's -- the split, the marking of 9 M received rows, the
# answers and the scatter are all there, the wire is a device copy); then the exchange tests of both forms again (the scan of
# the split is now one workgroup per owner).
out=gpurun_out/call19; mkdir -p $out
for form in allgather alltoall; do
  timeout 600 python bench.py --config c4 --files 1250000 --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline --force-exchange --exchange-form $form > $out/bench_c4_n1_$form.json 2> $out/bench_c4_n1_$form.err
  python - $out/bench_c4_n1_$form.json <<'P'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(j["config"].get("exchange_form"), j["value"], j["ms_per_step"], {k: v for k, v in j.items() if "exchange" in k or "marking" in k or "gather" in k})
P
done
export TMPDIR=/tmp
summ() { db=$(find $1 -name "*_results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py $db; }
cmd="python bench.py --config c4 --files 1250000 --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --force-exchange --exchange-form alltoall"
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- $cmd > $out/kt.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $cmd   (one rank, 9.02 M rows, alone on the GPU)"; summ $out/kt; } > $out/r06_kernel_trace_stats_alltoall_c4_one_rank.txt 2>&1
rm -rf $out/kt
grep -i -E "part_|answer|dedup" $out/r06_kernel_trace_stats_alltoall_c4_one_rank.txt | cut -c1-160
timeout 900 python -m pytest tests/test_gpu_native_exchange.py tests/test_gpu_parity.py -m gpu -x -q -k "native_exchange or rccl or plain_c" > $out/tests.txt 2>&1
grep -E "passed|failed" $out/tests.txt | tail -2
