#!/bin/bash
set -u
out=gpurun_out/r06i; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 1800 python -m pytest tests/test_gpu_dp_run_nn.py -q -m gpu -s > "$out/pytest_dp_run_nn.txt" 2>&1; echo "dp_run_nn rc=$?"; grep -E "run_nn_dp on|passed|failed|Error|assert" "$out/pytest_dp_run_nn.txt" | tail -12
