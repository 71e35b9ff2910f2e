#!/bin/bash
# round 6, call n: B fragments bound to AGPRs in the second-generation fp32 kernels too (Li-GRU / RNN): parity, launch times, fp32 rows
set -u
out=gpurun_out/r06n; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -k "not bf16" > "$out/pytest_parity.txt" 2>&1; echo "parity rc=$?"; tail -4 "$out/pytest_parity.txt"
timeout 600 python tools/bench_rec4.py > "$out/bench_rec4.txt" 2>&1; grep -A3 "liGRU flags=0\|LSTM flags=0\|GRU flags=0" "$out/bench_rec4.txt" | head -20
for rcp in timit_ligru timit_lstm libri_gru; do
  timeout 600 python bench.py --recipe $rcp --prec fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$out/bench_${rcp}.json" 2> "$out/bench_${rcp}.err"
  echo "$rcp: $(python tools/jget.py "$out/bench_${rcp}.json" ms_per_step 2>/dev/null)"
done
