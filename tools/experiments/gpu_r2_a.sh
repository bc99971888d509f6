#!/bin/bash
# round-2 GPU call A: parity of the new large-file CDC path + bench lines for every config
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
O=gpurun_out/r2a
timeout 1500 python -m pytest tests/test_gpu_large_files.py tests/test_gpu_parity.py tests/test_gpu_layer_tar.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
timeout 400 python bench.py --config c3 --no-cpu-baseline --no-host-fed > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
timeout 300 python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
timeout 200 python tools/quick_bench.py --files 4 --size 4294967296 --steps 2 > $O/four_4g.log 2>&1; echo "4x4g rc=$?"
timeout 200 python tools/quick_bench.py --files 1 --size 17179869184 --steps 2 > $O/one_16g.log 2>&1; echo "1x16g rc=$?"
tail -3 $O/four_4g.log $O/one_16g.log
for f in c2 c3 c5; do tail -c 1500 $O/bench_$f.json; echo; tail -3 $O/bench_$f.err; done
