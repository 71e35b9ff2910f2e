#!/bin/bash
# round 5, call N: the tree with the reference's mask stream as the default - whole GPU suite, smoke, the driver's command
set -u
out=$PWD/gpurun_out/r05n; mkdir -p "$out"
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/driver_cmd_line.json" 2> "$out/driver_cmd.err"; echo "driver cmd rc=$? $(python tools/jget.py $out/driver_cmd_line.json ms_per_step value config.mask_rng)"
timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > "$out/pytest_gpu.log" 2>&1; echo "gpu suite rc=$? $(tail -1 $out/pytest_gpu.log)"; grep -E "^FAILED|^E  " "$out/pytest_gpu.log" | head -8 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1; echo "smoke rc=$? $(tail -1 $out/smoke.txt | cut -c1-200)"
for v in device reference; do
  ms=$(timeout 300 python bench.py --mask-rng $v --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "mask-rng $v headline $ms" | tee -a "$out/ab.txt"
done
