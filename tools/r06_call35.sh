#!/bin/bash
# round 6, GPU call 35: the real 15 GB tree with the fourth handle over EIGHT ctxs on the one device (the nine files of 256 MiB and more
# split into parts over eight ctxs, 82 126 files to the least-loaded of eight)
mkdir -p gpurun_out/c35
MI_REAL_WARM=1 MI_REAL_N_CTXS=8 timeout 2400 python tools/real_tree_commit.py /usr 300 > gpurun_out/c35/r06_real_tree_commit_8ctxs.txt 2>&1
tail -18 gpurun_out/c35/r06_real_tree_commit_8ctxs.txt | cut -c1-260
