# round 6, GPU call 7: who pays a process's first large commit (the order of the sides), the host's CPU topology, the long-string test
mkdir -p gpurun_out
(echo "nproc $(nproc)"; taskset -p $$; lscpu | grep -E "Model name|Thread|Core|Socket|NUMA node\(s\)"; for c in $(taskset -pc $$ | sed 's/.*: //' | tr ',' ' '); do :; done; cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list) > gpurun_out/r06_host_cpus.txt 2>&1
python - >> gpurun_out/r06_host_cpus.txt <<'PY'
import os
cpus = sorted(os.sched_getaffinity(0))
sib = {}
for c in cpus:
    try:
        s = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
    except OSError:
        s = "?"
    sib.setdefault(s, []).append(c)
print("allowed cpus:", cpus)
print("sibling groups among them:", sib)
PY
cat gpurun_out/r06_host_cpus.txt
for order in gpu_first cpu_first gpu_first cpu_first; do
  echo "## MI_BENCH_ORDER=$order"
  MI_BENCH_ORDER=$order MI_LAYER_TIMING=1 timeout 300 python tools/commit_layer_bench.py 48 134217728 2>&1 | grep -E "mi_layer: 6442|^  all new|^    (gpu|cpu)" | head -7
done > gpurun_out/r06_first_commit_order.txt 2>&1
cat gpurun_out/r06_first_commit_order.txt
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_commit.py -m gpu -q -k "long_strings or driver_keeps or bench_line" 2>&1 | tail -15) > gpurun_out/r06_gputests_item3.txt
tail -5 gpurun_out/r06_gputests_item3.txt
