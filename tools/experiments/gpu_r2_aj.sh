#!/bin/bash
# round 2, call aj: large files marked one wave per tile + separate speculation kernel -- parity + speed
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_large_files.py tests/test_gpu_parts.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_staging.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
python tools/quick_bench.py --files 240 --size 134217728 --steps 4 2>&1 | tail -1
python tools/quick_bench.py --files 4 --size 4294967296 --steps 4 2>&1 | tail -1
python tools/quick_bench.py --files 20000 --size 200000 --steps 6 2>&1 | tail -1
python tools/quick_bench.py --lib tools/bin/libmi_base.so --files 20000 --size 200000 --steps 6 2>&1 | tail -1
python tools/quick_bench.py --lib tools/bin/libmi_base.so --files 240 --size 134217728 --steps 4 2>&1 | tail -1
