#!/bin/bash
# The host rows on the GPU box's host (round 5): what a first touch of fresh memory costs there, the layer merge and the scan at
# 10^6 and 10^7 entries straight through the C ABI.   tools/r05_host_rows.sh > gpurun_out/r05_host_rows.txt
cd "$(dirname "$0")/.."
gcc -O2 tools/ubench_page_faults.c -o /tmp/ubench_page_faults && { /tmp/ubench_page_faults 2; /tmp/ubench_page_faults 24; /tmp/ubench_page_faults 24 huge; }
gcc -O2 -I include tools/merge_scale.c -o /tmp/merge_scale -L makisu_amd -lmakisu_mi -Wl,-rpath,$PWD/makisu_amd || exit 1
for n in 10000 100000; do ( time MI_MOUNTS_FILE=/dev/null /tmp/merge_scale $n ) 2>&1 | grep -v "^$"; done
