#!/bin/bash
set -u
out=gpurun_out/r02h
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "persistent or oracle_parity or e2e or full_size_properties" > $out/parity.log 2>&1
echo "parity rc=$? $(tail -1 $out/parity.log)"
timeout 600 python -m pytest tests/test_gpu_reference_pins.py -q -m gpu -k "fp32" -s > $out/pins.log 2>&1
echo "pins rc=$? $(tail -1 $out/pins.log)"; grep -E "config-scale" $out/pins.log | cut -c1-300
timeout 300 python bench.py --prec fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_fp32.json 2> $out/bench_fp32.err
echo "bench rc=$? $(cut -c1-250 $out/bench_fp32.json)"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02h/bench_fp32.json').read().strip().split('\n')[-1])
print(d['ms_per_step'], d['entry_points_ms_per_step'])
PY
