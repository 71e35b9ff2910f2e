#!/bin/bash
# round 5, call V: when do the LSTM / GRU bf16 steps turn slow inside a long run, and what does the loss do there
set -u
out=$PWD/gpurun_out/r05v; mkdir -p "$out"
for r in libri_gru timit_lstm; do
  timeout 200 python bench.py --recipe $r --no-extras --no-cpu-baseline --steps 260 --warmup 3 --step-trace --dump-losses "$out/losses_$r.json" > "$out/line_$r.json" 2>/dev/null
  python - "$out/line_$r.json" "$out/losses_$r.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
st = d.get("step_ms", {})
ls = json.load(open(sys.argv[2]))
steps = st.get("all") or st.get("steps") or []
print(d["config"]["workload"][:20], "ms_per_step", d["ms_per_step"], "prewarm_steps", d["config"].get("prewarm_steps"), "trace keys", list(st.keys()))
if steps:
    slow = [i for i, v in enumerate(steps) if v > 2 * st.get("median", 1e9)]
    print("first slow steps", slow[:5], "count", len(slow), "values", [round(steps[i], 1) for i in slow[:5]])
import math
bad = [i for i, v in enumerate(ls) if not math.isfinite(v)]
print("losses: first", [round(v, 3) for v in ls[:3]], "last", [round(v, 3) for v in ls[-3:]], "non-finite from", bad[:1], "count", len(bad))
print("loss every 20:", [round(v, 2) for v in ls[::20]])
PY
done
