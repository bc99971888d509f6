timeout 400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== two ranks on one GPU over gloo (control flow of the N>1 bench)"
MI_BENCH_FORCE_DEVICE=0 timeout 170 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --exchange torch --backend gloo --files 20000 --steps 4 --warmup 1 2>&1 | tail -2 | cut -c1-600
echo "== one rank, forced exchange over RCCL"
timeout 170 python bench.py --force-exchange --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-420
tools/round_profiles.sh r01 2>&1 | tail -25
