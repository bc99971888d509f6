#!/bin/bash
# round 2, call ag: where does the large-file Gear path spend its time? (kernel trace + VALU instruction counts)
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r2ag; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace -d $out/kt -o p -- python tools/quick_bench.py --files 240 --size 134217728 --steps 3 > $out/kt.log 2>&1
db=$(find $out/kt -name "*_results.db" | head -1); python tools/prof_summary.py $db > $out/kt.txt 2>&1; rm -rf $out/kt
grep "gear\|sha256_items_kernel<0\|scan\|compact" $out/kt.txt | head
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $out/pmc -o p -- python tools/quick_bench.py --files 240 --size 134217728 --steps 2 > $out/pmc.log 2>&1
db=$(find $out/pmc -name "*_results.db" | head -1); python tools/prof_summary.py $db > $out/pmc.txt 2>&1; rm -rf $out/pmc
grep "gear_group_mark\|gear_file_fix\|gear_group_validate" $out/pmc.txt | head -30
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $out/pmc2 -o p -- python tools/quick_bench.py --files 100000 --size 65536 --steps 2 > $out/pmc2.log 2>&1
db=$(find $out/pmc2 -name "*_results.db" | head -1); python tools/prof_summary.py $db > $out/pmc2.txt 2>&1; rm -rf $out/pmc2
grep "gear_cdc_small" $out/pmc2.txt | head -12
