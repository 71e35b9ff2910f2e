#!/bin/bash
# round 5, call Y: LSTM then GRU in one process, host running ahead, default switches (helpers on) - with the step fence
set -u
out=$PWD/gpurun_out/r05y; mkdir -p "$out"
SYNC=0 STEPS=200 RECIPES=timit_lstm,libri_gru timeout 110 python tools/diag_slow_steps.py 2>&1 | grep -v amdgpu.ids | tee -a "$out/diag.txt"
