#!/bin/bash
# several environments of the headline bench on ONE box, round-robin:  tools/gpu_ab3.sh rounds "ENV=a" "ENV=b" ...
R="$1"; shift
mkdir -p gpurun_out/ab
python bench.py --no-extras --steps 5 >/dev/null 2>&1
for i in $(seq 1 $R); do
  for v in "$@"; do
    ms=$(env $v python bench.py --no-extras --steps 40 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$v  ms_per_step $ms" | tee -a gpurun_out/ab/log.txt
  done
done
