#!/bin/bash
# round 5, call H: what the epilogue stores of the row-streaming GEMMs cost, and whether staggered start phases spread them;
# + the re-check of call G (step-wise split-K limited to fp32)
set -u
out=$PWD/gpurun_out/r05h
mkdir -p "$out"
export SHAPES="64000:1100:1104:1:1;64000:1100:1100:1:0;64000:1938:1100:1:1" TILES=256 LDC_ALIGN=32 REPS=30
for v in "x=0" "gemm_nostore=1" "gemm_stagger=2" "gemm_stagger=4" "gemm_stagger=6" "gemm_stagger=9" "gemm_stagger=12"; do
  echo "== $v" | tee -a "$out/gemm.txt"
  PK_EXPERIMENT=$v timeout 120 python tools/bench_gemm.py 2>&1 | grep TFLOP | tee -a "$out/gemm.txt"
done
unset SHAPES TILES LDC_ALIGN REPS
for i in 1 2; do for v in "x=0" "gemm_stagger=4" "gemm_stagger=9"; do
  ms=$(PK_EXPERIMENT=$v timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "$v headline $ms" | tee -a "$out/ab.txt"
done; done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lstm_waves.py -q -m gpu > "$out/pytest_parity.log" 2>&1; echo "parity rc=$? $(tail -1 $out/pytest_parity.log)"; grep -E "^FAILED|^E  " "$out/pytest_parity.log" | head -8 | cut -c1-300
ms=$(timeout 600 python bench.py --recipe libri_gru --prec fp32 --no-extras --no-cpu-baseline --steps 2 --warmup 1 --prewarm-s 0 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step); echo "libri_gru fp32 $ms"
