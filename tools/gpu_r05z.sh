#!/bin/bash
# round 5, closing check on the FINAL tree (the evidence pass ran one commit earlier: the step-wise split-K was then limited to
# the fp32 mode): the driver's command as the first process, the whole GPU suite, smoke
set -u
out=$PWD/gpurun_out/r05z
mkdir -p "$out"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/driver_exact_cmd_line.json" 2> "$out/driver_exact.err"; echo "driver's exact command: $(python3 tools/jget.py $out/driver_exact_cmd_line.json ms_per_step value)"
PK_FULL_SHAPE_JSON=$out/full_shape.json timeout 1500 python -m pytest tests -q -m gpu > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$? $(tail -1 "$out/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)" "$out/pytest_gpu.log" | head -10
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1
echo "smoke rc=$? $(tail -2 "$out/smoke.log" | tr '\n' ' ')"
