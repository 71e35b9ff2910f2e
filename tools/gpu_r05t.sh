#!/bin/bash
# round 5, call T: (1) the reference's mask stream drawn IN LINE: stability of the LSTM / GRU / Li-GRU steps against the
# device RNG (the side-stream form made whole regions of LSTM / GRU steps 5-15 x slower, call R); (2) its tests;
# (3) the fp32 GEMM with fragment reads one k-pair ahead, A/B against the previous library on this box
set -u
out=$PWD/gpurun_out/r05t; mkdir -p "$out"
for r in timit_lstm libri_gru timit_ligru; do
  for v in device reference; do
    timeout 300 python bench.py --recipe $r --mask-rng $v --no-extras --no-cpu-baseline --steps 40 --warmup 3 --step-trace > "$out/line_${r}_$v.json" 2>/dev/null
    echo "$r mask-rng $v $(python tools/jget.py $out/line_${r}_$v.json ms_per_step step_ms.median step_ms.max)" | tee -a "$out/ab_masks.txt"
  done
done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "reference_mask" > "$out/pytest_rng.log" 2>&1; echo "rng tests rc=$? $(tail -1 $out/pytest_rng.log)"; grep -E "^FAILED|^E  " "$out/pytest_rng.log" | head -6 | cut -c1-300
L=pytorch-kaldi_amd/lib
cp $L/libpk_amd.so $L/libpk_amd_new.so
for arm in old new old new; do
  cp $L/libpk_amd_$arm.so $L/libpk_amd.so
  echo "== $arm" | tee -a "$out/gemm_f32.txt"
  SPLITS=6,12 timeout 300 python tools/bench_gemm_f32.py 2>&1 | grep -v amdgpu.ids | tee -a "$out/gemm_f32.txt" | cut -c1-110
done
for arm in old new; do
  cp $L/libpk_amd_$arm.so $L/libpk_amd.so
  for r in timit_ligru libri_gru; do
    ms=$(timeout 300 python bench.py --recipe $r --prec fp32 --no-extras --no-cpu-baseline --steps 4 --warmup 2 --prewarm-s 0 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
    echo "$arm $r fp32 $ms" | tee -a "$out/ab_f32.txt"
  done
done
cp $L/libpk_amd_new.so $L/libpk_amd.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -x -k "oracle_parity or golden or gemm or linear" > "$out/pytest_f32.log" 2>&1; echo "fp32 tests rc=$? $(tail -1 $out/pytest_f32.log)"
