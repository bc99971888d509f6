#!/bin/bash
# round 2, call ah: v_readlane instead of ds_bpermute in cut selection -- parity + speed
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_files.py tests/test_gpu_parts.py tests/test_gpu_configs.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
for i in 1 2; do
python tools/quick_bench.py --files 100000 --size 65536 --steps 20 2>&1 | tail -1
python tools/quick_bench.py --files 240 --size 134217728 --steps 4 2>&1 | tail -1
done
python tools/quick_bench.py --files 100000 --size 65536 --steps 30 --inflight 2 2>&1 | tail -1
python tools/quick_bench.py --files 4 --size 4294967296 --steps 4 2>&1 | tail -1
