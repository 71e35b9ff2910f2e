#!/bin/bash
# round 6, call m: second-tile B fragments of the fourth-generation fp32 recurrences bound to AGPRs (inline-asm MFMAs):
# parity of everything that runs these kernels, launch times, the fp32 rows
set -u
out=gpurun_out/r06m; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -k "not bf16" > "$out/pytest_parity.txt" 2>&1; echo "parity rc=$?"; tail -4 "$out/pytest_parity.txt"
timeout 600 python tools/bench_rec4.py > "$out/bench_rec4.txt" 2>&1; tail -12 "$out/bench_rec4.txt"
for rcp in timit_lstm libri_gru; do
  timeout 600 python bench.py --recipe $rcp --prec fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$out/bench_${rcp}.json" 2> "$out/bench_${rcp}.err"
  echo "$rcp: $(python tools/jget.py "$out/bench_${rcp}.json" ms_per_step 2>/dev/null)"
done
