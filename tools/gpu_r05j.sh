#!/bin/bash
# round 5, call J: the reference's drop-mask stream drawn on the device (tests, cost in the headline step)
set -u
out=$PWD/gpurun_out/r05j
mkdir -p "$out"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "reference_mask" > "$out/pytest_rng.log" 2>&1; echo "rng tests rc=$? $(tail -1 $out/pytest_rng.log)"; grep -E "^FAILED|^E  " "$out/pytest_rng.log" | head -8 | cut -c1-300
for v in device reference reference_host; do
  ms=$(timeout 300 python bench.py --mask-rng $v --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "mask-rng $v headline $ms" | tee -a "$out/ab.txt"
done
