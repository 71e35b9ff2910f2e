#!/usr/bin/env python
"""profiles/r02_pmc_traffic.json and profiles/r02_pmc_mfma_busy.json from the PMC summaries tools/rocpd_pmc.py wrote
(profiles/r02_pmc_fetch_size.csv, r02_pmc_write_size.csv, r02_pmc_sq.csv):  python tools/pmc_summaries.py [dir]

HBM traffic per launch = 2 x FETCH_SIZE (KB; MI355X_MICROARCH.md: gfx950's rocprofv3 reports half of a 16-byte-per-lane
streaming read) + WRITE_SIZE (as reported; it matches the algorithmic write volume of rec2_fwd_kernel).
Matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)."""
import csv
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r02"  # file-name prefix of the round


def table(name):
    out = {}
    for r in csv.DictReader(open(os.path.join(d, name))):
        out.setdefault(r["Kernel"], {})[r["Counter"]] = (float(r["MeanValue"]), int(r["Dispatches"]), float(r["MeanDurationNs"]))
    return out


RECIPE = sys.argv[3] if len(sys.argv) > 3 else None  # another BASELINE configuration: <TAG>_<recipe>_pmc_{fetch,write}_size.csv
# dominant entry point of each other configuration (bench.py's roofline.kernel) -> prefixes of the kernels behind it
OTHER = {"timit_lstm": {"pk_rec_bwd_bf16": ("rec3l_bwd_kernel", "rec2l_bwd_kernel"), "pk_rec_fwd_bf16": ("rec2l_fwd_kernel",)},
         "libri_gru": {"pk_rec2p_bwd_bf16": ("rec3g_bwd_kernel", "rec2g_bwd_kernel"), "pk_rec2p_fwd_bf16": ("rec2g_fwd_kernel",)},
         "timit_mlp": {"pk_gemm_bf16": ("gemm_bf16_t64_kernel", "gemm_bf16sk_kernel<false>", "gemm_bf16x_kernel", "gemm_bf16s_kernel",
                                        "splitk_reduce_bf_kernel")},
         "timit_sincnet": {"pk_conv1d_pool_bwd": ("conv_tile_kernel<1, 8, false", "conv_bwd_filter_tile_kernel")}}
if RECIPE:
    fetch, write = table("%s_%s_pmc_fetch_size.csv" % (TAG, RECIPE)), table("%s_%s_pmc_write_size.csv" % (TAG, RECIPE))
    out = {"_note": "HBM bytes per kernel dispatch behind the configuration's dominant entry point (dispatch-weighted mean over the "
           "kernels listed), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --recipe %s --steps 2 --warmup 1`; "
           "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on gfx950, WRITE_SIZE as "
           "reported.  whole_step_bytes = the same over EVERY kernel of a step.  Written by tools/pmc_summaries.py." % RECIPE}
    for entry, prefixes in OTHER[RECIPE].items():
        ks = [k for k in fetch if k.startswith(prefixes) and k in write]
        if not ks:
            continue
        disp = sum(fetch[k]["FETCH_SIZE"][1] for k in ks)
        byt = sum((2.0 * fetch[k]["FETCH_SIZE"][0] + write[k]["WRITE_SIZE"][0]) * 1024.0 * fetch[k]["FETCH_SIZE"][1] for k in ks)
        out[entry] = {"kernels": ks, "dispatches": disp, "traffic_bytes": int(round(byt / max(disp, 1)))}
    steps = 3.0  # --steps 2 --warmup 1, plus the two profiled steps of the roofline leg when it runs: reported per dispatch, not per step
    out["all_kernels_bytes_per_run"] = int(sum((2.0 * fetch[k]["FETCH_SIZE"][0] + write.get(k, {"WRITE_SIZE": (0.0,)})["WRITE_SIZE"][0]) * 1024.0 *
                                               fetch[k]["FETCH_SIZE"][1] for k in fetch))
    json.dump(out, open(os.path.join(d, "%s_pmc_traffic_%s.json" % (TAG, RECIPE)), "w"), indent=1)
    print(json.dumps(out, indent=1)[:700])
    sys.exit(0)
fetch, write, sq = table(TAG + "_pmc_fetch_size.csv"), table(TAG + "_pmc_write_size.csv"), table(TAG + "_pmc_sq.csv")
# (round 3: the backward pass runs the third-generation kernel, pk_rec_persist3.hip)
ENTRY = {"pk_rec_bwd_bf16": ("rec3_bwd_kernel<0, 1", "rec2_bwd_kernel<0, 1"), "pk_rec_fwd_bf16": ("rec2_fwd_kernel<0, 1", "rec3_fwd_kernel<0, 1")}
traffic = {"_note": "HBM bytes per launch from rocprofv3 --pmc (separate passes for FETCH_SIZE and WRITE_SIZE, bench.py --steps 2 "
           "--warmup 1; profiles/<round>_pmc_fetch_size.csv, profiles/<round>_pmc_write_size.csv, values in KB). FETCH_SIZE is doubled "
           "as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on gfx950; WRITE_SIZE is used as reported. "
           "Written by tools/pmc_summaries.py."}
for entry, prefixes in ENTRY.items():
    kf = kw = prefix = None
    for prefix in prefixes:
        kf = [k for k in fetch if k.startswith(prefix)]
        kw = [k for k in write if k.startswith(prefix)]
        if kf and kw:
            break
    if not kf or not kw:
        continue
    f, w = fetch[kf[0]]["FETCH_SIZE"][0], write[kw[0]]["WRITE_SIZE"][0]
    traffic[entry] = {"kernel": prefix, "fetch_kb_reported": round(f, 3), "write_kb_reported": round(w, 3),
                      "traffic_bytes": int(round((2.0 * f + w) * 1024.0))}
json.dump(traffic, open(os.path.join(d, TAG + "_pmc_traffic.json"), "w"), indent=1)
busy = {"_note": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
        "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- python bench.py --steps 2 --warmup 1 "
        "(profiles/<round>_pmc_sq.csv). Shares are of SQ_WAVE_CYCLES (quad-cycles): wait_any = parked in s_waitcnt / s_barrier, "
        "wait_inst = issue stalls, active = issuing. mfma_busy_frac_per_simd = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / "
        "(GRBM_GUI_ACTIVE / 8 XCDs): the fraction of the launch during which a SIMD's matrix pipe was busy = MFMA "
        "utilisation. Written by tools/pmc_summaries.py.", "kernels": {}}
rows = sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", (0, 0, 0))[0] * kv[1].get("SQ_WAVE_CYCLES", (0, 1, 0))[1])
for k, c in rows[:14]:
    if "SQ_WAVE_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    wc = c["SQ_WAVE_CYCLES"][0] or 1.0
    busy["kernels"][k] = {
        "dispatches": c["SQ_WAVE_CYCLES"][1], "mean_duration_us": round(c["SQ_WAVE_CYCLES"][2] / 1e3, 1),
        "wait_any_share": round(c.get("SQ_WAIT_ANY", (0,))[0] / wc, 3),
        "wait_inst_share": round(c.get("SQ_WAIT_INST_ANY", (0,))[0] / wc, 3),
        "active_inst_share": round(c.get("SQ_ACTIVE_INST_ANY", (0,))[0] / wc, 3),
        "mfma_busy_frac_per_simd": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0,))[0] / 1024.0 / (c["GRBM_GUI_ACTIVE"][0] / 8.0), 4),
        "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT", (0,))[0]}
json.dump(busy, open(os.path.join(d, TAG + "_pmc_mfma_busy.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1)[:900])
for k, v in list(busy["kernels"].items())[:8]:
    print("%-60s %s" % (k[:60], v))
