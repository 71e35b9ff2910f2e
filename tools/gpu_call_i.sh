#!/bin/bash
set -u
out=gpurun_out/r02i
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lstm_waves.py -q -m gpu -x > $out/parity.log 2>&1
echo "parity rc=$? $(tail -1 $out/parity.log)"
JSON_OUT=$out/trace_full.json timeout 120 python tools/trace_rec2.py > $out/trace_full.log 2>&1; grep -E "cycles/step|launch ms" $out/trace_full.log
for t in 0 128; do
PK_GEMM_TILE=$t timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 60 > $out/bench_tile$t.json 2> $out/bench_tile$t.err
echo "tile=$t rc=$? $(python - <<PY
import json
d=json.loads(open('$out/bench_tile$t.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['entry_points_ms_per_step'])
PY
)"
done
for r in timit_lstm libri_gru; do timeout 200 python bench.py --recipe $r --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_$r.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('$out/bench_$r.json').read().strip().split('\n')[-1]); print('$r', d['ms_per_step'], list(d['entry_points_ms_per_step'].items())[:3])
PY
done
timeout 200 python bench.py --prec fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_fp32.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('$out/bench_fp32.json').read().strip().split('\n')[-1]); print('fp32', d['ms_per_step'], list(d['entry_points_ms_per_step'].items())[:3])
PY
