#!/bin/bash
# One GPU-box pass for a round: the whole GPU suite, smoke(), the default bench line (headline + parity_mode +
# other_configs + cpu_baseline) and the rocprofv3 kernel trace of the default command.  Run through gpurun from the repo
# root; everything lands under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r02'
# Arguments: <tag> [skip-tests]
set -u
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p "$out"
# a box whose GPU faults on its first touch (seen in round 3: 'Memory access fault' inside the first .cuda() of a process,
# before any kernel of this library ran) burns minutes on core dumps: stop at once
if ! timeout 90 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
if [ "${2:-}" != "skip-tests" ]; then
    timeout 1500 python -m pytest tests -q -m gpu > "$out/pytest_gpu.log" 2>&1
    echo "pytest rc=$? $(tail -1 "$out/pytest_gpu.log")"
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1
    echo "smoke rc=$? $(tail -2 "$out/smoke.log" | tr '\n' ' ')"
fi
PK_BENCH_VERBOSE=1 timeout 900 python bench.py > "$out/bench_bf16.json" 2> "$out/bench_bf16.err"
echo "bench rc=$? $(cut -c1-200 "$out/bench_bf16.json")"
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof" -- \
    python "$OLDPWD/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-extras > "$OLDPWD/$out/prof_bench.log" 2>&1 )
echo "rocprofv3 rc=$?"
db=$(find "$out/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then
    python tools/rocpd_stats.py "$db" "$out/kernel_stats.csv" > /dev/null 2> "$out/kernel_stats.err" || true
    head -8 "$out/kernel_stats.csv" 2>/dev/null
    rm -rf "$out/prof"
fi
