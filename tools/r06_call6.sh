# round 6, GPU call 6: the end-to-end sums on hardware (both hops), what they cost, long strings on host streams, the gpu suite
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/r06_gputests_sums.txt
grep -E "passed|failed" gpurun_out/r06_gputests_sums.txt
for v in 1 0; do
  echo "## MI_COMMIT_VERIFY=$v"
  MI_COMMIT_VERIFY=$v MI_LAYER_TIMING=1 timeout 300 python tools/commit_layer_bench.py 48 134217728 2>&1 | grep -E "all new|mi_layer: 6442|^    (gpu|cpu)" | head -8
  MI_COMMIT_VERIFY=$v timeout 300 python tools/commit_layer_bench.py 100000 4096 2>&1 | grep -E "all new" -A3
done > gpurun_out/r06_verify_cost.txt 2>&1
cat gpurun_out/r06_verify_cost.txt
(MI_FEED_MODES=add_path,add_path,add_path timeout 300 python tools/host_feed_bench.py 48 128; MI_FEED_SUMS=1 MI_FEED_MODES=add_path,add_path,add_path timeout 300 python tools/host_feed_bench.py 48 128) > gpurun_out/r06_feed_sums.txt 2>&1
cat gpurun_out/r06_feed_sums.txt
