#!/bin/bash
# round 2, call y: bench lines of the large configs with the cooperative chunk pass
mkdir -p gpurun_out/r2y
cd /root/repo
out=gpurun_out/r2y
python bench.py --config c3 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err; grep '^{' $out/bench_c3.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'], {k:v for k,v in d['config'].items() if 'host' in k or 'fed' in k})"
python bench.py --config c5 --no-cpu-baseline > $out/bench_c5.json 2> $out/bench_c5.err; grep '^{' $out/bench_c5.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', d['value'], d['ms_per_step'])"
python bench.py --config c4 --no-cpu-baseline --steps 4 --warmup 1 > $out/bench_c4.json 2> $out/bench_c4.err; grep '^{' $out/bench_c4.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', d['value'], d['ms_per_step'])"
python tools/quick_bench.py --files 4 --size 4294967296 --steps 4 2>&1 | tail -1
python tools/quick_bench.py --files 4 --size 4294967296 --steps 4 --inflight 2 2>&1 | tail -1
python bench.py --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err; grep '^{' $out/bench_c2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['ms_per_step'])"
