#!/bin/bash
# Round 5, the first GPU call, everything in one (a call's acquisition costs minutes; its commands do not):
#   1. tools/gpu_round_check.sh      the suite, smoke, both N > 1 launch forms on the RCCL double, the forced exchange  (~4 min)
#   2. PARTS=core tools/round_profiles.sh r05   bench line, kernel traces, FETCH / WRITE, SQ counters for C2          (~2 min)
#   3. (was: the CU partition probe -- run once in round 5, profiles/r05_cu_partition_probe.txt: no split gains, the knob is gone)
#   4. the host rows where system calls and page faults cost what they cost in production                             (~1 min)
# Everything lands under gpurun_out/first_call/; nothing here decides anything -- read, then act (tools/experiments/README.md).
#   gpurun --timeout 900 -- 'tools/round5_first_call.sh'
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/first_call
mkdir -p $out
[ -f tests/rccl_stub/libmi_rccl_stub.so ] || python -c "import __graft_entry__ as g; g.build()" > $out/build.txt 2>&1   # (build() makes the doubles too)
echo "== 1. round check"; timeout 420 tools/gpu_round_check.sh 2>&1 | tee $out/round_check.txt | tail -25
echo "== 2. core profiles"; PARTS=core timeout 240 tools/round_profiles.sh r05 2>&1 | tee $out/round_profiles.txt | tail -15
echo "== 4. host rows"
{ MI_WALK_TIMING=1 timeout 170 python tools/many_small_files.py 200 500 4; timeout 170 python tools/commit_layer_bench.py 100000 4096; timeout 170 python tools/host_scale_bench.py 1000000; } 2>&1 | tee $out/host_rows.txt | tail -30
