#!/bin/bash
# round 6, call b: the fourth-generation fp32 recurrences (pk_rec_persist4_f32.hip) - parity first, then the fp32 recipes' step times
set -u
out=gpurun_out/r06b; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "oracle_parity_h550" > "$out/pytest_h550.txt" 2>&1; echo "h550 rc=$?"; tail -4 "$out/pytest_h550.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "golden_module_persistent or (full_geometry and fp32)" -s > "$out/pytest_persist.txt" 2>&1; echo "persist rc=$?"; grep -E "full geometry|passed|failed|FAILED|Error" "$out/pytest_persist.txt" | tail -20
timeout 900 python -m pytest tests/test_gpu_reference_pins.py -q -m gpu -k "recipe_scale and (lstm or gru)" -s > "$out/pytest_scale.txt" 2>&1; echo "scale rc=$?"; grep -E "passed|failed|FAILED|Error|scale" "$out/pytest_scale.txt" | tail -12
for r in timit_lstm libri_gru; do
  for g4 in 1 0; do
    PK_EXPERIMENT=rec_f32_gen4=$g4 timeout 600 python3 bench.py --recipe $r --prec fp32 --steps 6 --warmup 2 --no-extras --no-cpu-baseline > "$out/${r}_fp32_gen4_$g4.json" 2> "$out/${r}_fp32_gen4_$g4.err"
    echo "$r fp32 gen4=$g4: $(python3 tools/jget.py "$out/${r}_fp32_gen4_$g4.json" ms_per_step)"
  done
done
