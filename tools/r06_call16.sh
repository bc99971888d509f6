#!/bin/bash
# round 6, call 16: the hash-partitioned exchange (mi_dedup_alltoall[_all]) on the GPU: the native-exchange tests in both
# forms (2 / 3 / 8 ranks on the double, ragged / C4 / C5), the single-rank tests on the real RCCL, and bench.py --gpus 8
# bare in both forms (8 ctxs on this one GPU: the marking and host times compare, the wire does not exist here).
out=gpurun_out/call16; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_native_exchange.py tests/test_gpu_parity.py -m gpu -x -q -k "native or rccl" > $out/tests.txt 2>&1
tail -5 $out/tests.txt
export MI_BENCH_FORCE_DEVICE=0 MI_RCCL_LIB=$PWD/tests/rccl_stub/libmi_rccl_stub.so
for form in allgather alltoall; do
  timeout 600 python bench.py --gpus 8 --files 100000 --steps 5 --warmup 2 --no-cpu-baseline --exchange-form $form > $out/bench8_$form.json 2> $out/bench8_$form.err
  python - $out/bench8_$form.json <<'P'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(j["config"]["exchange_form"], j["value"], j["ms_per_step"], "gather", j["per_rank"]["exchange_gather_ms"], "marking", j["per_rank"]["marking_ms"], j["dedup_check"]["ok"])
P
done
