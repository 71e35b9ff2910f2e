#!/bin/bash
# Mid-round check of a changed tree on the GPU box: the whole GPU suite, the headline line without its side records, a
# kernel trace of it, and one short line per other recipe.   gpurun --timeout 1500 -- 'bash tools/gpu_verify.sh <tag>'
set -u
tag=${1:-rXXv}
out=gpurun_out/$tag
mkdir -p "$out"
if ! timeout 90 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
timeout 1200 python -m pytest tests -q -m gpu --maxfail=8 > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$? $(tail -1 "$out/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)" "$out/pytest_gpu.log" | head -10
PK_BENCH_VERBOSE=1 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > "$out/headline.json" 2> "$out/headline.err"
echo "headline rc=$? $(python tools/jget.py "$out/headline.json" ms_per_step step_ms 2>/dev/null | cut -c1-300)"
for r in timit_lstm libri_gru timit_mlp timit_sincnet; do
    timeout 300 python bench.py --recipe $r --steps 30 --warmup 5 --no-cpu-baseline --no-extras > "$out/$r.json" 2> "$out/$r.err"
    echo "$r rc=$? $(python tools/jget.py "$out/$r.json" ms_per_step 2>/dev/null | cut -c1-120)"
done
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -- \
    python "$OLDPWD/bench.py" --steps 8 --warmup 2 --prewarm-s 0 --no-cpu-baseline --no-extras > "$OLDPWD/$out/prof_bench.log" 2>&1 )
echo "rocprofv3 rc=$?"
db=$(find "$out/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then
    python tools/rocpd_stats.py "$db" "$out/kernel_stats.csv" > /dev/null 2> "$out/kernel_stats.err" || true
    python tools/rocpd_dump.py "$db" "$out/tail.csv" 900 > /dev/null 2>&1 || true
    head -6 "$out/kernel_stats.csv" 2>/dev/null
    rm -rf "$out/prof"
fi
