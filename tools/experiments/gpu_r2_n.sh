#!/bin/bash
# round 2, call n: parts (split files) + regression of the large-file path
mkdir -p gpurun_out/r2n
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parts.py tests/test_gpu_large_files.py -x -q 2>&1 | tail -30 > gpurun_out/r2n/pytest_parts.txt
cat gpurun_out/r2n/pytest_parts.txt
python tools/quick_bench.py --files 4 --size 4294967296 --steps 3 > gpurun_out/r2n/large4.txt 2>&1
tail -3 gpurun_out/r2n/large4.txt
