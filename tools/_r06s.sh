#!/bin/bash
set -u
out=gpurun_out/r06s; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
for d in 0 1 2 4 6; do
  PK_EXPERIMENT=f32_dbg=$d SPLITS=1 timeout 200 python tools/bench_gemm_f32.py 2>&1 | head -6
done | tee "$out/f32_dbg.txt"
