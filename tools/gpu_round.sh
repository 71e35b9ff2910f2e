#!/bin/bash
# One GPU-box pass for a round: parity suite, the bench line of every BASELINE recipe, and the rocprofv3 kernel trace of
# the default bench command.  Run through gpurun from the repo root; everything lands under gpurun_out/<tag>/ (copy what
# is to be judged into profiles/).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02'
# Arguments: <tag> [skip-tests]
set -u
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p "$out"
if [ "${2:-}" != "skip-tests" ]; then
    timeout 900 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.log" 2>&1
    echo "pytest rc=$? $(tail -1 "$out/pytest_gpu.log")"
fi
timeout 300 python bench.py > "$out/bench_bf16.json" 2> "$out/bench_bf16.err"
echo "bench rc=$? $(cut -c1-200 "$out/bench_bf16.json")"
for r in timit_lstm libri_gru timit_mlp timit_sincnet; do
    timeout 200 python bench.py --recipe $r --steps 8 --warmup 3 --no-cpu-baseline > "$out/bench_$r.json" 2> "$out/bench_$r.err"
    echo "$r rc=$? $(python - "$out/bench_$r.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(d["ms_per_step"], "ms", d["value"], d["unit"], list(d["entry_points_ms_per_step"].items())[:3])
except Exception as e:
    print("unreadable:", e)
PY
)"
done
# kernel trace of the default command (no counters in this pass: gpurun refuses --pmc together with other traces)
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -- \
    python "$OLDPWD/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > "$OLDPWD/$out/prof_bench.log" 2>&1 )
echo "rocprofv3 rc=$?"
db=$(find "$out/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then
    python tools/rocpd_stats.py "$db" "$out/kernel_stats.csv" > /dev/null 2> "$out/kernel_stats.err" || true
    head -12 "$out/kernel_stats.csv" 2>/dev/null
fi
