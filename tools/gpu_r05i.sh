#!/bin/bash
# round 5, call I: the 256-tile GEMM's epilogue transposed across lanes (eight consecutive lanes = 128 contiguous bytes of a row)
set -u
out=$PWD/gpurun_out/r05i
mkdir -p "$out"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "gemm or projection or linear" > "$out/pytest_gemm.log" 2>&1; echo "gemm tests rc=$? $(tail -1 $out/pytest_gemm.log)"; grep -E "^FAILED|^E  " "$out/pytest_gemm.log" | head -8 | cut -c1-300
export SHAPES="64000:1100:1104:1:1;64000:1100:1100:1:0;64000:1938:1100:1:1;1100:1104:64000:0:0;1938:1100:64000:0:0" TILES=256 REPS=30
for v in "gemm_epi=0" "gemm_epi=1"; do
  for la in 1 32; do
  echo "== $v ldc align $la" | tee -a "$out/gemm.txt"
  LDC_ALIGN=$la PK_EXPERIMENT=$v timeout 120 python tools/bench_gemm.py 2>&1 | grep TFLOP | tee -a "$out/gemm.txt"
  done
done
unset SHAPES TILES REPS
for i in 1 2 3; do for v in "gemm_epi=0" "gemm_epi=1"; do
  ms=$(PK_EXPERIMENT=$v timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "$v headline $ms" | tee -a "$out/ab.txt"
done; done
