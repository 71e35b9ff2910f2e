#!/bin/bash
# round 5, call M: the mask stream no longer waits for the caller's stream
set -u
out=$PWD/gpurun_out/r05m; mkdir -p "$out"
timeout 300 python tools/diag_ref_rng.py 2>&1 | tee "$out/diag.txt" | tail -5
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -k "reference_mask or mask_rng or ref_rng or reference_stream" > "$out/pytest_rng.log" 2>&1; echo "rng tests rc=$? $(tail -1 $out/pytest_rng.log)"; grep -E "^FAILED|^E  " "$out/pytest_rng.log" | head -8 | cut -c1-300
for v in device reference device reference; do
  ms=$(timeout 300 python bench.py --mask-rng $v --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "mask-rng $v headline $ms" | tee -a "$out/ab.txt"
done
