#!/bin/bash
# round 6, call 27: how many address ranges of an arena's size a process can reserve (they are never given back), on both runtimes
out=gpurun_out/call27; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/vmm_va_probe.hip -o /tmp/vmm_va_probe
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
{
echo "# tools/vmm_va_probe.hip (tools/r06_call27.sh): the system's runtime, then the one PyTorch bundles (LD_PRELOAD of its libamdhip64 + libhsa-runtime64)"
timeout 300 /tmp/vmm_va_probe 32
timeout 300 /tmp/vmm_va_probe 32 1
LD_PRELOAD=$TL/libamdhip64.so:$TL/libhsa-runtime64.so timeout 300 /tmp/vmm_va_probe 32
LD_PRELOAD=$TL/libamdhip64.so:$TL/libhsa-runtime64.so timeout 300 /tmp/vmm_va_probe 32 1
} > $out/r06_vmm_va_probe.txt 2>&1
cat $out/r06_vmm_va_probe.txt
