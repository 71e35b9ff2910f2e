#!/usr/bin/env python
"""profiles/<round>_rec_step_floor.json from the two traces tools/gpu_evidence.sh takes with tools/trace_rec2.py (full step / EMPTY=1):
    python tools/make_step_floor.py gpurun_out/floor3 profiles/r03_rec_step_floor.json "<note>"
bench.py reads hop_us.{fwd,bwd} and floor_us_{fwd,bwd} from it (latency record of the roofline object)."""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
full, empty = json.load(open(src + "/trace_full.json")), json.load(open(src + "/trace_empty.json"))


def part(d):
    us = d["us_per_step_hip_events"]
    return {"fwd_us": us["pk_rec_fwd_bf16"], "bwd_us": us["pk_rec_bwd_bf16"], "fwd_cycles": d["fwd"]["cycles_per_step_mean"],
            "bwd_cycles": d["bwd"]["cycles_per_step_mean"], "fwd_phases": d["fwd"]["phases_mean"], "bwd_phases": d["bwd"]["phases_mean"],
            "fwd_poll_retries_per_step": d["fwd"]["poll_retries_per_step"], "bwd_poll_retries_per_step": d["bwd"]["poll_retries_per_step"]}


f, e = part(full), part(empty)
out = {"_note": note, "full_step": f, "empty_step": e,
       # the poll phase of the full step, converted with that launch's own clocks-per-microsecond
       "hop_us": {"fwd": f["fwd_phases"]["poll"] * f["fwd_us"] / f["fwd_cycles"], "bwd": f["bwd_phases"]["poll"] * f["bwd_us"] / f["bwd_cycles"]},
       "floor_us_fwd": e["fwd_us"], "floor_us_bwd": e["bwd_us"]}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hop_us", "floor_us_fwd", "floor_us_bwd")}))
