#!/bin/bash
# round 5, call S: fp32 GEMM with the fragment reads one k-pair ahead of the MFMAs - A/B against the previous library
# (pytorch-kaldi_amd/lib/libpk_amd_old.so, built from the commit before) on one box
set -u
out=$PWD/gpurun_out/r05s; mkdir -p "$out"
L=pytorch-kaldi_amd/lib
cp $L/libpk_amd.so $L/libpk_amd_new.so
for arm in old new old new; do
  cp $L/libpk_amd_$arm.so $L/libpk_amd.so
  echo "== $arm" | tee -a "$out/gemm_f32.txt"
  SPLITS=6,12 timeout 300 python tools/bench_gemm_f32.py 2>&1 | grep -v amdgpu.ids | tee -a "$out/gemm_f32.txt"
done
for arm in old new; do
  cp $L/libpk_amd_$arm.so $L/libpk_amd.so
  for r in timit_ligru libri_gru; do
    ms=$(timeout 300 python bench.py --recipe $r --prec fp32 --no-extras --no-cpu-baseline --steps 4 --warmup 2 --prewarm-s 0 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
    echo "$arm $r fp32 $ms" | tee -a "$out/ab.txt"
  done
done
cp $L/libpk_amd_new.so $L/libpk_amd.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -x -k "oracle_parity or golden or gemm or linear" > "$out/pytest_f32.log" 2>&1; echo "fp32 tests rc=$? $(tail -1 $out/pytest_f32.log)"
