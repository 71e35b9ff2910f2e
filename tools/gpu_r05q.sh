#!/bin/bash
# round 5, call Q: the exact-fp32 GEMM per k-tile depth (BK = 16 / 32) on the reference-precision mode's shapes
set -u
out=$PWD/gpurun_out/r05q; mkdir -p "$out"
for arm in f32_bk=16 f32_bk=32; do
  PK_EXPERIMENT=$arm timeout 300 python tools/bench_gemm_f32.py 2>&1 | tee -a "$out/gemm_f32.txt" | tail -22
done
