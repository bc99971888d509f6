#!/bin/bash
# round 6, GPU call 33: the randomized differential soak with every arena in pieces (2 MiB and 32 MiB) AND hashed with the cooperative
# loads from the first byte -- the combination the library now chooses from 1 GiB on, on shapes no fixed case enumerates
mkdir -p gpurun_out/c33
{
MI_ARENA=pieces MI_ARENA_PIECE_MB=2 MI_SHA_COOP_MIN_GIB_PIECES=0 timeout 400 python tools/gpu_fuzz.py 240 3301 2>&1 | tail -2
MI_ARENA=pieces MI_SHA_COOP_MIN_GIB_PIECES=0 timeout 400 python tools/gpu_fuzz.py 240 3302 2>&1 | tail -2
timeout 300 python tools/gpu_fuzz.py 120 3303 2>&1 | tail -2
} > gpurun_out/c33/r06_gpu_fuzz_coop_on_pieces.txt
cat gpurun_out/c33/r06_gpu_fuzz_coop_on_pieces.txt
