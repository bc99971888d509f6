#!/bin/bash
# round 2, call ai: EXPERIMENT (not parity-safe for dense tiles): 16 table copies in LDS, bitmaps in global scratch
cd /root/repo
for i in 1 2; do
for v in base gexp16; do
python tools/quick_bench.py --lib tools/bin/libmi_$v.so --files 100000 --size 65536 --steps 20 2>&1 | tail -1 | sed "s/^/$v /"
python tools/quick_bench.py --lib tools/bin/libmi_$v.so --files 100000 --size 65536 --steps 30 --inflight 2 2>&1 | tail -1 | sed "s/^/$v /"
done
done
