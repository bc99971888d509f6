#!/bin/bash
# round 6, GPU call 28: which load scheme the chunk pass wants on an arena of small pieces, by footprint; then the gpu suite on the
# library that chooses by it
mkdir -p gpurun_out
timeout 1200 python tools/pieces_scheme_ab.py > gpurun_out/r06_pieces_scheme_ab.txt 2> gpurun_out/r06_pieces_scheme_ab.err
cat gpurun_out/r06_pieces_scheme_ab.txt; tail -3 gpurun_out/r06_pieces_scheme_ab.err
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/r06_gputests_call28.txt
cat gpurun_out/r06_gputests_call28.txt
