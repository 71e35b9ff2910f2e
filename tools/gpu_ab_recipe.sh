#!/bin/bash
# A/B of environment switches on ONE box for a given recipe, round-robin:
#   tools/gpu_ab_recipe.sh <recipe> <rounds> <steps> "ENV=a" "ENV=b" ...
RCP="$1"; R="$2"; ST="$3"; shift 3
mkdir -p gpurun_out/ab
python bench.py --recipe "$RCP" --no-extras --no-cpu-baseline --steps 5 >/dev/null 2>&1
for i in $(seq 1 $R); do
  for v in "$@"; do
    ms=$(env $v python bench.py --recipe "$RCP" --no-extras --no-cpu-baseline --steps "$ST" --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$RCP $v  ms_per_step $ms" | tee -a gpurun_out/ab/log_$RCP.txt
  done
done
