#!/bin/bash
# round 2, call r: same-box A/B of the SHA load variants (old = byte-aligned dwordx4, new = dword-aligned + perm)
mkdir -p gpurun_out/r2r
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r2r
for rep in 1 2 3; do
  for v in old new; do
    python tools/quick_bench.py --lib tools/bin/libmi_$v.so --files 100000 --size 65536 --steps 20 2>&1 | tail -1 | sed "s/^/$v serial   /" | tee -a $out/ab.txt
    python tools/quick_bench.py --lib tools/bin/libmi_$v.so --files 100000 --size 65536 --steps 20 --inflight 2 2>&1 | tail -1 | sed "s/^/$v inflight2 /" | tee -a $out/ab.txt
  done
done
for v in old new; do
  python tools/quick_bench.py --lib tools/bin/libmi_$v.so --files 240 --size 134217728 --steps 4 2>&1 | tail -1 | sed "s/^/$v big /" | tee -a $out/ab.txt
done
for v in old new; do
  timeout 300 rocprofv3 --kernel-trace -d $out/kt_$v -o p -- python tools/quick_bench.py --lib tools/bin/libmi_$v.so --files 100000 --size 65536 --steps 5 > $out/kt_$v.log 2>&1
  db=$(find $out/kt_$v -name "*_results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py $db > $out/kt_$v.txt 2>&1; rm -rf $out/kt_$v
  grep "sha256_items_kernel<0>" $out/kt_$v.txt | head -2 | sed "s/^/$v kernel-trace /"
done
