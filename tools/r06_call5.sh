# round 6, GPU call 5: hybrid arena + two read-back windows: the commit tables (system runtime, and PyTorch's), the gpu suite, the bench line
mkdir -p gpurun_out
MI_LAYER_TIMING=1 MI_ARENA_TRACE=1 timeout 300 python tools/commit_layer_bench.py 48 134217728 > gpurun_out/r06_commit_large_2win.txt 2>&1
timeout 300 python -c "import torch, runpy, sys; sys.argv=['commit_layer_bench.py','48','134217728']; runpy.run_path('tools/commit_layer_bench.py', run_name='__main__')" > gpurun_out/r06_commit_large_2win_torch.txt 2>&1
timeout 300 python -c "import torch, runpy, sys; sys.argv=['commit_layer_bench.py','100000','4096']; runpy.run_path('tools/commit_layer_bench.py', run_name='__main__')" > gpurun_out/r06_commit_small_2win_torch.txt 2>&1
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/r06_gputests_hybrid.txt
timeout 600 python bench.py > gpurun_out/r06_bench_hybrid.json 2> gpurun_out/r06_bench_hybrid.err
grep -h "all new" -A3 gpurun_out/r06_commit_large_2win.txt gpurun_out/r06_commit_large_2win_torch.txt gpurun_out/r06_commit_small_2win_torch.txt
tail -3 gpurun_out/r06_gputests_hybrid.txt
