#!/bin/bash
# round 5, call U: closing records on the final tree: the driver's command, the default bench line, the whole GPU suite, smoke
set -u
tag=r05
out=$PWD/gpurun_out/r05u; mkdir -p "$out"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/driver_cmd_line.json" 2> "$out/driver_cmd.err"
echo "driver command, first process: $(python3 tools/jget.py $out/driver_cmd_line.json ms_per_step value config.mask_rng roofline.frac)"
PK_BENCH_VERBOSE=1 timeout 900 python bench.py > "$out/${tag}_bench_bf16.json" 2> "$out/bench_bf16.err"
echo "bench rc=$? $(python3 tools/jget.py $out/${tag}_bench_bf16.json ms_per_step value parity_mode.ms_per_step cpu_baseline.value)"
python3 - "$out/${tag}_bench_bf16.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
for o in d["other_configs"]:
    pm = o.get("parity_mode")
    print(o["recipe"], o["ms_per_step"], o.get("regions_ms_per_step"), pm.get("ms_per_step") if isinstance(pm, dict) else "")
PY
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > "$out/pytest_gpu.log" 2>&1; echo "gpu suite rc=$? $(tail -1 $out/pytest_gpu.log)"; grep -E "^FAILED|^ERROR" "$out/pytest_gpu.log" | head -8 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1; echo "smoke rc=$? $(tail -1 $out/smoke.txt | cut -c1-200)"
