#!/bin/bash
set -u
out=gpurun_out/r06d; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "oracle_parity_h550" > "$out/pytest_h550.txt" 2>&1; echo "h550 rc=$?"; tail -4 "$out/pytest_h550.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "golden_module_persistent or (full_geometry and fp32)" -s > "$out/pytest_persist.txt" 2>&1; echo "persist rc=$?"; grep -E "full geometry|passed|failed|FAILED|Error" "$out/pytest_persist.txt" | tail -20
timeout 900 python tools/bench_rec4.py --cells LSTM,GRU --flags 0,1,3 > "$out/rec4_diag.json" 2> "$out/rec4_diag.err"; cat "$out/rec4_diag.json"; tail -3 "$out/rec4_diag.err"
for r in timit_lstm libri_gru; do
    timeout 600 python3 bench.py --recipe $r --prec fp32 --steps 6 --warmup 2 --no-extras --no-cpu-baseline > "$out/${r}_fp32.json" 2> "$out/${r}_fp32.err"
    echo "$r fp32: $(python3 tools/jget.py "$out/${r}_fp32.json" ms_per_step entry_points_ms_per_step)"
done
