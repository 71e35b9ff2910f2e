#!/bin/bash
# round 3, final evidence pass: the whole GPU suite + smoke + the default bench line + kernel trace (gpu_round.sh),
# PMC passes and the other recipes' kernel traces (gpu_profiles_r03.sh), step floor traces, the CPU port at the metric's
# full shape, the reference-stream mask mode
bash tools/gpu_round.sh r03f
out=gpurun_out/floor3; mkdir -p $out
JSON_OUT=$out/trace_full.json timeout 120 python tools/trace_rec2.py > $out/trace_full.log 2>&1
EMPTY=1 JSON_OUT=$out/trace_empty.json timeout 120 python tools/trace_rec2.py > $out/trace_empty.log 2>&1
grep -E "cycles/step" $out/trace_full.log $out/trace_empty.log
python bench.py --mask-rng reference --steps 30 --no-extras --no-cpu-baseline > gpurun_out/r03f/bench_maskref.json 2> gpurun_out/r03f/bench_maskref.err
echo "mask-rng reference: $(python -c "import json;d=json.loads(open('gpurun_out/r03f/bench_maskref.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
bash tools/gpu_profiles_r03.sh > gpurun_out/r03f/profiles.log 2>&1; tail -12 gpurun_out/r03f/profiles.log
timeout 900 python bench.py --cpu-full > gpurun_out/r03f/cpu_full_shape.json 2> gpurun_out/r03f/cpu_full_shape.err; tail -c 600 gpurun_out/r03f/cpu_full_shape.json
