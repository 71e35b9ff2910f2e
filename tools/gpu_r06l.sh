#!/bin/bash
# round 6, call l: self-filling exchange in the fourth-generation fp32 recurrences - parity (twice: stale mailboxes of the first
# pass are what the second one starts from), then the A/B on the fp32 rows
set -u
out=gpurun_out/r06l; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu > "$out/pytest_parity.txt" 2>&1; echo "parity rc=$?"; tail -4 "$out/pytest_parity.txt"
timeout 1200 python -m pytest tests/test_gpu_full_shape.py tests/test_gpu_round6.py -q -m gpu > "$out/pytest_full.txt" 2>&1; echo "full shape rc=$?"; tail -3 "$out/pytest_full.txt"
for rnd in 1 2; do
  for rcp in timit_lstm libri_gru; do
    for sw in 1 0; do
      PK_EXPERIMENT="rec4_self_fill=$sw" timeout 600 python bench.py --recipe $rcp --prec fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$out/bench_${rcp}_fill${sw}_$rnd.json" 2> "$out/bench_${rcp}_fill${sw}_$rnd.err"
      echo "$rcp self_fill=$sw round $rnd: $(python tools/jget.py "$out/bench_${rcp}_fill${sw}_$rnd.json" ms_per_step 2>/dev/null)"
    done
  done
done
