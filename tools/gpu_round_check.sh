#!/bin/bash
# The round's GPU check, from the repo root on the MI355X box: the GPU suite, smoke, the N > 1 bench in both launch forms
# on the RCCL double, the forced exchange on real RCCL.  Output that matters lands in gpurun_out/round_check/.
# A process that dies with "Memory access fault" / an HSA error leaves nothing but that line: the suite is then run ONCE
# MORE from the test that died, with the runtime naming every kernel and copy it launches, serialized
# (AMD_LOG_LEVEL=3, AMD_SERIALIZE_KERNEL=3), and the last 200 lines of THAT are kept -- kernel name and address.
out=gpurun_out/round_check
mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1
grep -E "passed|failed|error" $out/gputests.txt | tail -3
if grep -qE "Memory access fault|HSA_STATUS_ERROR|Aborted|core dumped" $out/gputests.txt; then
    last=$(grep -oE "tests/test_[a-z_]+\.py" $out/gputests.txt | tail -1)
    echo "== a process died on the GPU; once more with the runtime's launch log: ${last:-the whole suite}"
    AMD_LOG_LEVEL=3 AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout 900 python -m pytest ${last:-tests} -x -q -m gpu 2>&1 |
        grep -E "ShaderName|hipLaunchKernel|hipMemcpy|fault|HSA_STATUS|PASSED|FAILED|::test_" | tail -200 > $out/fault_launch_log.txt
    tail -5 $out/fault_launch_log.txt
fi
echo "== smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
STUB=$PWD/tests/rccl_stub/libmi_rccl_stub.so
echo "== bench.py --gpus 8 launched bare: one process, eight ctxs on this GPU, the library's exchange on the RCCL double"
MI_BENCH_FORCE_DEVICE=0 MI_RCCL_LIB=$STUB timeout 200 python bench.py --gpus 8 --files 20000 --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-500
echo "== the driver's launch line, two ranks on this GPU: native exchange, torch ships the id (gloo)"
MI_BENCH_FORCE_DEVICE=0 MI_RCCL_LIB=$STUB timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --files 20000 --steps 4 --warmup 1 2>&1 | tail -2 | cut -c1-500
echo "== one rank, forced exchange over real RCCL"
timeout 170 python bench.py --force-exchange --steps 10 --warmup 2 --no-cpu-baseline --no-with-rows --no-commit-e2e 2>&1 | tail -1 | cut -c1-420
