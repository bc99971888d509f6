#!/bin/bash
# round 6, call 17: the two forms of the digest exchange at 8 ranks x 250 000 files (1.8 M rows per rank, 14.4 M job-wide; 8 ctxs
# on this one GPU through the double: the WIRE is host copies here and says nothing, the device work of either form is real):
# per-rank device times of the last exchange (mi_comm_exchange_ms) and the kernels under rocprofv3.
out=gpurun_out/call17; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.txt 2>&1
# (the tests of both forms: tools/r06_call16.sh and the round-end suite)

export MI_BENCH_FORCE_DEVICE=0 MI_RCCL_LIB=$PWD/tests/rccl_stub/libmi_rccl_stub.so
for form in allgather alltoall; do
  timeout 900 python bench.py --gpus 8 --files 250000 --inflight 1 --steps 4 --warmup 1 --no-cpu-baseline --exchange-form $form > $out/bench8_250k_$form.json 2> $out/bench8_250k_$form.err
  python - $out/bench8_250k_$form.json <<'P'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(j["config"]["exchange_form"], j["value"], j["ms_per_step"], "wire(double)", j["per_rank"]["exchange_gather_ms"], "device work", j["per_rank"]["marking_ms"], j["dedup_check"]["ok"], sum(j["config"]["chunks_per_rank_last_batch"]))
P
done
cd /tmp && export TMPDIR=/tmp
for form in allgather alltoall; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_$form -o x -- python $GRAFT_REPO_ROOT/bench.py --gpus 8 --files 250000 --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-n1 --exchange-form $form > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$out/prof_$form -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$out/kernel_stats_$form.csv
  rm -rf $GRAFT_REPO_ROOT/$out/prof_$form
done
