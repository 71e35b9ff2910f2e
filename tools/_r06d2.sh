#!/bin/bash
set -u
out=gpurun_out/r06d2; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bn_bwd_f32 or batchnorm or gemm" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | head -12
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -k "not bf16" 2>&1 | tail -3
for i in 1 2; do
for rcp in libri_gru; do
  for d in "f32_dp_pitch=0,bn_f32_vec=0" "f32_dp_pitch=0" "f32_dp_pitch=1"; do
    PK_EXPERIMENT=$d timeout 600 python bench.py --recipe $rcp --prec fp32 --steps 12 --warmup 3 --no-cpu-baseline --no-extras > "$out/b.json" 2> "$out/b.err"
    echo "$rcp $d: $(python tools/jget.py "$out/b.json" ms_per_step loss_final 2>/dev/null)"
  done
done; done | tee $out/ab.txt
