#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r2ae
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2ae/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2ae/pytest.log; grep -E "passed|failed|error" gpurun_out/r2ae/pytest.log | tail -3
MI_SHA_COOP_MIN_GIB=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_files.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2; do
python tools/quick_bench.py --files 100000 --size 65536 --steps 20 2>&1 | tail -1
python tools/quick_bench.py --files 100000 --size 65536 --steps 30 --inflight 2 2>&1 | tail -1
done
python tools/quick_bench.py --files 400000 --size 65536 --steps 6 2>&1 | tail -1
