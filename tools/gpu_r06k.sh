#!/bin/bash
# round 6, call k: per-step LayerNorm in the fourth-generation fp32 recurrences (LSTM / GRU / minimalGRU): the parity file
set -u
out=gpurun_out/r06k; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "layernorm or laynorm or ln" > "$out/pytest_ln.txt" 2>&1; echo "ln rc=$?"; tail -15 "$out/pytest_ln.txt"
timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > "$out/pytest_parity.txt" 2>&1; echo "parity rc=$?"; tail -5 "$out/pytest_parity.txt"
