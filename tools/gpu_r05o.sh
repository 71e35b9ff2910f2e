#!/bin/bash
# round 5, call O: the exact-fp32 GEMM - items mapped XCD by XCD, weight-gradient products split over 1024 slots
set -u
out=$PWD/gpurun_out/r05o; mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -x -k "oracle_parity or golden or gemm or linear" > "$out/pytest_f32.log" 2>&1; echo "fp32 tests rc=$? $(tail -1 $out/pytest_f32.log)"; grep -E "^FAILED|^E  " "$out/pytest_f32.log" | head -8 | cut -c1-300
for arm in "gemm_f32_flat=1,f32_splitk_slots=256" "gemm_f32_flat=0,f32_splitk_slots=256" "gemm_f32_flat=1,f32_splitk_slots=1024" "gemm_f32_flat=0,f32_splitk_slots=1024"; do
  for r in timit_ligru timit_lstm libri_gru; do
    ms=$(PK_EXPERIMENT=$arm timeout 300 python bench.py --recipe $r --prec fp32 --no-extras --no-cpu-baseline --steps 4 --warmup 2 --prewarm-s 0 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
    echo "$arm $r fp32 $ms" | tee -a "$out/ab.txt"
  done
done
