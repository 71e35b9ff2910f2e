#!/bin/bash
set -u
out=gpurun_out/r02j
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16 or full_size or side_stream" > $out/parity.log 2>&1
echo "parity rc=$? $(tail -1 $out/parity.log)"
JSON_OUT=$out/trace_full.json timeout 120 python tools/trace_rec2.py > $out/trace_full.log 2>&1; grep -E "cycles/step|launch ms|mean" $out/trace_full.log | head -20
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 60 > $out/bench.json 2> $out/bench.err
python - <<PY
import json
d=json.loads(open('$out/bench.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['entry_points_ms_per_step'])
PY
