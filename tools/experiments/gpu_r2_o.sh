#!/bin/bash
# round 2, call o: (1) large files with two batches in flight; (2) which TLB counters exist;
# (3) clocks/power during a long SHA launch (C3-shaped batch) vs C2
mkdir -p gpurun_out/r2o
cd /root/repo
export TMPDIR=/tmp
python tools/quick_bench.py --files 4 --size 4294967296 --steps 4 --inflight 2 > gpurun_out/r2o/large4_inflight2.txt 2>&1
tail -2 gpurun_out/r2o/large4_inflight2.txt
(rocprofv3 --list-avail 2>&1 || rocprofv3 -L 2>&1) > gpurun_out/r2o/avail.txt
grep -i -c "" gpurun_out/r2o/avail.txt
grep -i "utcl\|tlb\|xnack\|translat" gpurun_out/r2o/avail.txt | head -40
# clocks during a long C3-shaped serial run
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics" ; sleep 0.1; done ) > gpurun_out/r2o/smi_c3.txt &
SMI=$!
python tools/quick_bench.py --files 480 --size 134217728 --steps 40 > gpurun_out/r2o/c3_serial.txt 2>&1
wait $SMI
tail -1 gpurun_out/r2o/c3_serial.txt
sort gpurun_out/r2o/smi_c3.txt | uniq -c | sort -rn | head -12
