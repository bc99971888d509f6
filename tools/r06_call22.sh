#!/bin/bash
# round 6, call 22: mutation testing ON THE GPU BOX of the host code that only runs with real kernels -- files split over the ctxs
# of a commit as parts (csrc/mi_api.hip: the split in add_paths, group_resolve_parts, the roots of split rows, reading a split
# row, a split row's chunk sums) -- against tests/test_gpu_commit.py's commits over several ctxs (7 parameter sets, 3 of them with
# MI_COMMIT_SPLIT_MIB lowered so that files are split).  On the HIP double these lines cannot be reached (no cut records).
out=gpurun_out/call22; mkdir -p $out
T="tests/test_gpu_commit.py::test_the_commit_over_several_ctxs_on_the_gpu"
for r in 1000-1085 1525-1572 1770-1800 1853-1875 2010-2075; do
  timeout 1500 python tools/mutate_host.py makisu_amd/csrc/mi_api.hip --tests $T --marker gpu --n 12 --jobs 6 --seed 7 --lines $r --timeout 500 \
      --work /tmp/mi_mut_gpu --out $out/mut_$r.txt > $out/mut_$r.log 2>&1
  tail -1 $out/mut_$r.log
done
cat $out/mut_*.txt > $out/r06_mutation_split_files_gpu.txt
grep -c SURVIVED $out/r06_mutation_split_files_gpu.txt
