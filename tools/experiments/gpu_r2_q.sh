#!/bin/bash
# round 2, call q: dword-aligned SHA loads -- parity, speed (C2 / 32 GiB batch), UTCL1 requests
mkdir -p gpurun_out/r2q
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r2q
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_files.py -x -q 2>&1 | tail -5 > $out/pytest.txt
cat $out/pytest.txt
python tools/quick_bench.py --files 100000 --size 65536 --steps 10 > $out/c2_serial.txt 2>&1; tail -2 $out/c2_serial.txt
python tools/quick_bench.py --files 100000 --size 65536 --steps 10 --inflight 2 > $out/c2_inflight2.txt 2>&1; tail -1 $out/c2_inflight2.txt
python tools/quick_bench.py --files 240 --size 134217728 --steps 4 > $out/big_serial.txt 2>&1; tail -1 $out/big_serial.txt
python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; grep '^{' $out/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
for shape in "c2 --files 100000 --size 65536"; do
  set -- $shape; name=$1; shift
  ctr="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"
  tag=${name}_utcl1
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $out/$tag -o p -- python tools/quick_bench.py "$@" --steps 2 > $out/$tag.log 2>&1
  db=$(find $out/$tag -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/prof_summary.py $db > $out/$tag.txt 2>&1
  rm -rf $out/$tag
  grep -i "sha256_items_kernel<0>" $out/$tag.txt | head -6
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o p -- python tools/quick_bench.py --files 100000 --size 65536 --steps 2 > $out/fetch.log 2>&1
db=$(find $out/fetch -name "*_results.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py $db > $out/fetch.txt 2>&1; rm -rf $out/fetch
grep -i "sha256_items_kernel<0>\|gear_cdc_small" $out/fetch.txt | head -6
