#!/bin/bash
# round 2, call v: quad-cooperative SHA loads: parity, then same-box A/B (old = MI_SHA_COOP 0)
mkdir -p gpurun_out/r2v
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r2v
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_files.py tests/test_gpu_configs.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 | tee $out/pytest.txt
for rep in 1 2; do
  for v in old new; do
    python tools/quick_bench.py --lib tools/bin/libmi_$v.so --files 100000 --size 65536 --steps 20 2>&1 | tail -1 | sed "s/^/$v serial   /" | tee -a $out/ab.txt
    python tools/quick_bench.py --lib tools/bin/libmi_$v.so --files 100000 --size 65536 --steps 20 --inflight 2 2>&1 | tail -1 | sed "s/^/$v inflight2 /" | tee -a $out/ab.txt
  done
done
for v in old new; do
  python tools/quick_bench.py --lib tools/bin/libmi_$v.so --files 240 --size 134217728 --steps 4 2>&1 | tail -1 | sed "s/^/$v big /" | tee -a $out/ab.txt
done
