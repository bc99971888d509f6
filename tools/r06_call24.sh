#!/bin/bash
# round 6, call 24: mutants of the all-to-all exchange's HOST code (csrc/mi_comm.hip: the split's launch, the plan from the job's
# matrix, the grouped sends / receives, the owner's marking, the two entry points) on the GPU box against the exchange tests of
# that form -- 3 and 8 ctxs of one process (ragged with an empty rank, C5 with split files), 3 processes, the plain-C driver.
out=gpurun_out/call24; mkdir -p $out
P=tests/test_gpu_native_exchange.py
T="$P::test_native_exchange_n_ctxs_in_one_process[3-ragged-alltoall] $P::test_native_exchange_n_ctxs_in_one_process[8-ragged-alltoall] $P::test_native_exchange_n_ctxs_in_one_process[8-c5-alltoall] $P::test_native_exchange_n_processes_on_one_gpu[3-ragged-alltoall] $P::test_plain_c_exchange[alltoall]"
for r in 396-566 757-858; do
  timeout 2400 python tools/mutate_host.py makisu_amd/csrc/mi_comm.hip --tests $T --marker gpu --n 30 --jobs 6 --seed 8 --lines $r --timeout 240 \
      --work /tmp/mi_mut_gpu --out $out/mut_$r.txt > $out/mut_$r.log 2>&1
  tail -1 $out/mut_$r.log
done
cat $out/mut_*.txt > $out/r06_mutation_alltoall_gpu.txt
grep SURVIVED $out/r06_mutation_alltoall_gpu.txt | cut -c1-250
