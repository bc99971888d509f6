#!/bin/bash
# round 6, call 18: the kernels of the two exchange forms under rocprofv3 (8 ctxs x 250 000 files on this one GPU, the double for
# the wire): split (part_hist / part_scan / part_scatter), the owner's marking, answer / answer_scatter against the all-gather
# form's insert / tag / probe / finish.
out=gpurun_out/call18; mkdir -p $out
export TMPDIR=/tmp MI_BENCH_FORCE_DEVICE=0 MI_RCCL_LIB=$PWD/tests/rccl_stub/libmi_rccl_stub.so
summ() { db=$(find $1 -name "*_results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py $db; }
for form in allgather alltoall; do
  cmd="python bench.py --gpus 8 --files 250000 --inflight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-n1 --exchange-form $form"
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt_$form -o kt -- $cmd > $out/kt_$form.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- $cmd   (8 ctxs on ONE GPU: kernels of different ranks overlap)"; summ $out/kt_$form; } > $out/r06_kernel_trace_stats_exchange_$form.txt 2>&1
  rm -rf $out/kt_$form
  grep -i -E "part_|answer|dedup" $out/r06_kernel_trace_stats_exchange_$form.txt | cut -c1-200
done
