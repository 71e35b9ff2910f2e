#!/bin/bash
# round 5, call W: the sporadic multi-second LSTM steps - with and without the L2 run-ahead helpers, allocator numbers beside
set -u
out=$PWD/gpurun_out/r05w; mkdir -p "$out"
PK_REC_HELPER=0 STEPS=160 timeout 100 python tools/diag_slow_steps.py 2>&1 | grep -v amdgpu.ids | tee -a "$out/diag.txt"
STEPS=160 timeout 100 python tools/diag_slow_steps.py 2>&1 | grep -v amdgpu.ids | tee -a "$out/diag.txt"
