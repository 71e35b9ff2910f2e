#!/bin/bash
# A/B of two environments of the headline bench on ONE box, alternating (boxes differ by +-3 %, so do first runs):
#   tools/gpu_ab.sh "PK_X=0" "PK_X=1" [rounds] [extra bench.py args]
A="$1"; B="$2"; R="${3:-2}"; shift 3
mkdir -p gpurun_out/ab
python bench.py --no-extras --steps 5 "$@" >/dev/null 2>&1   # warm the box
for i in $(seq 1 $R); do
  for v in "$A" "$B"; do
    ms=$(env $v python bench.py --no-extras --steps 40 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$v  ms_per_step $ms" | tee -a gpurun_out/ab/log.txt
  done
done
