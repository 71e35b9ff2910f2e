#!/bin/bash
set -u
out=gpurun_out/r06h; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 600 python -m pytest tests/test_gpu_round6.py -q -m gpu > "$out/pytest_round6.txt" 2>&1; echo "round6 rc=$?"; tail -3 "$out/pytest_round6.txt"
timeout 1800 python -m pytest tests/test_gpu_dp_run_nn.py -x -q -m gpu -s > "$out/pytest_dp_run_nn.txt" 2>&1; echo "dp_run_nn rc=$?"; grep -E "run_nn_dp on|passed|failed|Error" "$out/pytest_dp_run_nn.txt" | tail -8
timeout 600 python tools/step_ops_profile.py timit_mlp > "$out/ops_timit_mlp.txt" 2> "$out/ops_timit_mlp.err"; tail -45 "$out/ops_timit_mlp.txt"
