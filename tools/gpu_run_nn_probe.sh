#!/bin/bash
# the chunk-loop record of bench.py on its own: five chunks in a fresh process, then again behind an fp32 step
set -u
out=gpurun_out/${1:-r04s}
mkdir -p "$out"
timeout 200 python -c "
import sys, json
sys.argv = ['bench.py']
import bench
a = bench.parse()
print(json.dumps(bench.through_run_nn(a, reps=6)))
a.prec = 'fp32'
rec, tr = bench.measure(a, 0, 1, 3, 1)
bench.release(tr)
a.prec = 'bf16'
print(json.dumps(bench.through_run_nn(a, reps=6)))
" > "$out/run_nn_probe.txt" 2> "$out/run_nn_probe.err"
echo "rc=$?"; grep -o '"ms_per_step": [0-9.]*\|"chunks_s": \[[^]]*\]' "$out/run_nn_probe.txt"
tail -3 "$out/run_nn_probe.err"
