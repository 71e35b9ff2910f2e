#!/bin/bash
set -u
out=gpurun_out/r06e; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 600 python tools/trace_rec4.py > "$out/trace.json" 2> "$out/trace.err"; cat "$out/trace.json"; tail -3 "$out/trace.err"
