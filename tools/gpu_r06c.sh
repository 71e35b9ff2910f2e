#!/bin/bash
set -u
out=gpurun_out/r06c; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 900 python tools/bench_rec4.py > "$out/rec4_diag.json" 2> "$out/rec4_diag.err"; cat "$out/rec4_diag.json"; tail -3 "$out/rec4_diag.err"
