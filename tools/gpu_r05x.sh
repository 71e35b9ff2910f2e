#!/bin/bash
# round 5, call X: LSTM then GRU in one process, host running ahead (as bench.py's other_configs), helpers off
set -u
out=$PWD/gpurun_out/r05x; mkdir -p "$out"
PK_REC_HELPER=0 SYNC=0 STEPS=200 RECIPES=timit_lstm,libri_gru timeout 150 python tools/diag_slow_steps.py 2>&1 | grep -v amdgpu.ids | tee -a "$out/diag.txt"
