#!/bin/bash
# SincNet: bf16 convolutions (PK_CONV_BF16) graded on the recipe-scale fixture, A/B of the step, kernel trace.
set -u
tag=${1:-r04o}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
for m in 1 2; do
  PK_CONV_BF16=$m timeout 600 python -m pytest tests/test_gpu_reference_pins.py -q -m gpu -x -s -k "sincnet and bf16" > "$out/pytest_conv$m.log" 2>&1
  echo "PK_CONV_BF16=$m pytest rc=$? $(tail -1 "$out/pytest_conv$m.log")"; grep -E "^FAILED|^E  |assert" "$out/pytest_conv$m.log" | head -6 | cut -c1-300
done
for i in 1 2; do for m in 0 1 2; do
  ms=$(PK_CONV_BF16=$m timeout 200 python bench.py --recipe timit_sincnet --steps 100 --warmup 5 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step regions_ms_per_step)
  echo "PK_CONV_BF16=$m timit_sincnet $ms" | tee -a "$out/ab.txt"
done; done
R=$PWD
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$out/prof" -- python "$R/bench.py" --recipe timit_sincnet --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$out/prof.log" 2>&1 )
db=$(find "$out/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" "$out/r04_timit_sincnet_kernel_stats.csv" > /dev/null 2> "$out/kstats.err"; head -25 "$out/r04_timit_sincnet_kernel_stats.csv" | cut -c1-160; rm -rf "$out/prof"; fi
