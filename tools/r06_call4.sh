# round 6, GPU call 4: the gpu suite on the piecewise-mapped arena; the headline bench over piece sizes against one hipMalloc
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/r06_gputests_vmm3.txt
tail -3 gpurun_out/r06_gputests_vmm3.txt
: > gpurun_out/r06_arena_ab.txt
for round in 1 2; do
  for mode in malloc 2 32 256 1024; do
    if [ $mode = malloc ]; then export MI_ARENA=malloc; unset MI_ARENA_PIECE_MB; else unset MI_ARENA; export MI_ARENA_PIECE_MB=$mode; fi
    timeout 300 python bench.py --no-cpu-baseline --no-with-rows --no-commit-e2e 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('round $round arena %-7s value %8.2f GiB/s  ms/step %.4f  sha serial %.4f ms  inflight span %.4f  other scheme serial %.4f' % ('$mode', d['value'], d['ms_per_step'], r['avg_launch_ms'], r.get('inflight_span_ms', 0), r['other_load_scheme_serial']['serial_launch_ms']))" >> gpurun_out/r06_arena_ab.txt
  done
done
unset MI_ARENA MI_ARENA_PIECE_MB
cat gpurun_out/r06_arena_ab.txt
T=/usr/local/lib/python3.10/dist-packages/torch/lib
(for i in 1 2 3; do LD_PRELOAD=$T/libamdhip64.so:$T/libhsa-runtime64.so timeout 200 tools/bin/ubench_vmm 6 | grep "at once"; done) > gpurun_out/r06_ubench_vmm_torch_runtime.txt 2>&1
cat gpurun_out/r06_ubench_vmm_torch_runtime.txt
