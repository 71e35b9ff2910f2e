#!/bin/bash
# does the row pitch of the fp32 output matter? (1100 floats = 4400 B rows start in the middle of 128-byte lines)
mkdir -p gpurun_out/gemm_sq
for al in 1 32 64; do
SHAPES="64000:1100:1104:1:1;64000:1100:1100:1:0;64000:1938:1100:1:1" REPS=20 LDC_ALIGN=$al \
  timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/gemm_sq/out_ldc.txt
