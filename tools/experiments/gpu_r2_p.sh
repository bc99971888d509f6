#!/bin/bash
# round 2, call p: UTCL1 (TLB) counters of the SHA chunk pass, C2-sized vs 32 GiB batch
mkdir -p gpurun_out/r2p
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r2p
for shape in "c2 --files 100000 --size 65536" "big --files 240 --size 134217728"; do
  set -- $shape; name=$1; shift
  for ctr in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_THRASHING_STALL TCP_UTCL1_STALL_INFLIGHT_MAX GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
    tag=${name}_$(echo $ctr | cut -c1-30 | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $out/$tag -o p -- python tools/quick_bench.py "$@" --steps 2 > $out/$tag.log 2>&1
    db=$(find $out/$tag -name "*_results.db" | head -1)
    [ -n "$db" ] && python tools/prof_summary.py $db > $out/$tag.txt 2>&1
    rm -rf $out/$tag
    grep -i "sha256_items_kernel<0>\|UTCL\|GRBM" $out/$tag.txt | head -12
  done
done
