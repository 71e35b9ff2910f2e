#!/bin/bash
set -u
out=gpurun_out/r06u; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
for rcp in timit_ligru timit_lstm libri_gru; do
  for d in "f32_dma=0" "f32_dma=1"; do
    PK_EXPERIMENT=$d timeout 600 python bench.py --recipe $rcp --prec fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$out/bench_${rcp}_$d.json" 2> "$out/bench_${rcp}_$d.err"
    echo "$rcp $d: $(python tools/jget.py "$out/bench_${rcp}_$d.json" ms_per_step 2>/dev/null)"
  done
done
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -k "not bf16" > "$out/pytest_parity.txt" 2>&1; echo "parity rc=$?"; tail -3 "$out/pytest_parity.txt"
