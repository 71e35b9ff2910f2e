#!/bin/bash
# Full arithmetic with doctored I/O (role-split kernels, NP = 0): EMPTY=6 cache-hit loads, 7 cache-hit loads and no stores;
# 0 = the real step.  What a step could cost if the HBM traffic did not run on the recurrence's own CUs.
set -u
tag=${1:-r04n}
out=gpurun_out/$tag
mkdir -p "$out"
for e in 0 6 7; do
  PK_REC_GEN=5 PK_SPLIT_POLLERS=0 EMPTY=$e JSON_OUT="$out/trace_np0_e$e.json" timeout 120 python tools/trace_rec2.py > "$out/trace_np0_e$e.log" 2>&1
  echo "EMPTY=$e: $(grep -vE 'amdgpu' "$out/trace_np0_e$e.log" | tr '\n' ' ' | tr -s ' ' | cut -c1-900)"
done
JSON_OUT="$out/trace_default.json" timeout 120 python tools/trace_rec2.py > "$out/trace_default.log" 2>&1
echo "default gens: $(grep -E 'cycles/step|launch ms' "$out/trace_default.log" | tr '\n' ' ' | cut -c1-300)"
