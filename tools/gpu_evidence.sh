#!/bin/bash
# The round's closing evidence pass on ONE GPU box (run through gpurun from the repo root; everything lands under
# gpurun_out/<tag>ev/, the summaries to be judged are then copied into profiles/ by hand):
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/gpu_evidence.sh r05'
# Order = what matters most first:
#   1. the driver's exact command as the FIRST process on the fresh box (what BENCH_rNN.json will hold), with the step trace
#   2. the whole GPU suite, smoke()
#   3. the default bench line (headline + parity_mode + other_configs + cpu_baseline)
#   1b. the hand-off / step-floor traces of the recurrence (tools/trace_rec2.py, full and EMPTY) and the PMC passes of the
#       headline (FETCH_SIZE / WRITE_SIZE separately; SQ busy shares): what the bench line's roofline record reads
#   4. rocprofv3 kernel trace of the headline
#   5. kernel traces of the other recipes and of the fp32 mode
#   7. one full-shape bf16 step against the oracle's bf16-operand model (minutes of host time)
set -u
tag=${1:-rXX}
out=$PWD/gpurun_out/${tag}ev
R=$PWD
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
# ---- 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --step-trace > "$out/driver_cmd_first.json" 2> "$out/driver_cmd_first.err"
echo "driver command, first process: $(python3 tools/jget.py "$out/driver_cmd_first.json" ms_per_step value step_ms.first step_ms.median config.prewarm_steps)"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --step-trace > "$out/driver_cmd_second.json" 2> "$out/driver_cmd_second.err"
echo "driver command, second process: $(python3 tools/jget.py "$out/driver_cmd_second.json" ms_per_step value step_ms.first step_ms.median)"
# ---- 1b: what the bench line's roofline record reads (written into profiles/ of THIS copy before bench.py runs)
JSON_OUT=$out/trace_full.json timeout 150 python tools/trace_rec2.py > $out/trace_full.log 2>&1; grep -E "cycles/step|launch ms" $out/trace_full.log | head -4
EMPTY=1 JSON_OUT=$out/trace_empty.json timeout 150 python tools/trace_rec2.py > $out/trace_empty.log 2>&1; grep -E "cycles/step|launch ms" $out/trace_empty.log | head -4
python tools/make_step_floor.py $out $out/${tag}_rec_step_floor.json "round ${tag#r}: tools/trace_rec2.py on the closing tree, full step and EMPTY=1 (no arithmetic)" 2>&1 | tail -1
cp $out/${tag}_rec_step_floor.json profiles/ 2>/dev/null
( cd /tmp && export TMPDIR=/tmp
S="python $R/bench.py --steps 2 --warmup 1 --prewarm-s 0 --no-cpu-baseline --no-extras"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -- $S > $out/pmc_fetch.log 2>&1
python $R/tools/rocpd_pmc.py $(find $out/pmc_fetch -name "*.db" | head -1) $out/${tag}_pmc_fetch_size.csv
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -- $S > $out/pmc_write.log 2>&1
python $R/tools/rocpd_pmc.py $(find $out/pmc_write -name "*.db" | head -1) $out/${tag}_pmc_write_size.csv
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $out/pmc_sq -- $S > $out/pmc_sq.log 2>&1
python $R/tools/rocpd_pmc.py $(find $out/pmc_sq -name "*.db" | head -1) $out/${tag}_pmc_sq.csv
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_sq
python $R/tools/pmc_summaries.py $out $tag > $out/pmc_summaries.log 2>&1; head -12 $out/pmc_summaries.log
# the other BASELINE configurations: HBM bytes behind their dominant entry points (bench.py: other_configs[*].roofline.traffic)
for r in timit_lstm libri_gru timit_mlp timit_sincnet; do
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_f_$r -- $S --recipe $r > $out/pmc_f_$r.log 2>&1
  python $R/tools/rocpd_pmc.py $(find $out/pmc_f_$r -name "*.db" | head -1) $out/${tag}_${r}_pmc_fetch_size.csv
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_w_$r -- $S --recipe $r > $out/pmc_w_$r.log 2>&1
  python $R/tools/rocpd_pmc.py $(find $out/pmc_w_$r -name "*.db" | head -1) $out/${tag}_${r}_pmc_write_size.csv
  rm -rf $out/pmc_f_$r $out/pmc_w_$r
  python $R/tools/pmc_summaries.py $out $tag $r > $out/pmc_summaries_$r.log 2>&1; head -3 $out/pmc_summaries_$r.log
  cp $out/${tag}_pmc_traffic_$r.json $R/profiles/ 2>/dev/null
done
)
cp $out/${tag}_pmc_traffic.json $out/${tag}_pmc_mfma_busy.json profiles/ 2>/dev/null
# ---- 2
PK_FULL_SHAPE_JSON=$out/${tag}_full_shape_parity.json timeout 2400 python -m pytest tests -q -m gpu > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$? $(tail -1 "$out/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)" "$out/pytest_gpu.log" | head -10
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1
echo "smoke rc=$? $(tail -2 "$out/smoke.log" | tr '\n' ' ')"
# ---- 3
# (--cpu-full-in-run: the CPU port's step at the metric's FULL shape timed inside this run - ~4 minutes of host time; the
# driver's default command keeps the bounded sample and quotes this file's figure)
PK_BENCH_VERBOSE=1 timeout 2400 python bench.py --cpu-full-in-run > "$out/${tag}_bench_bf16.json" 2> "$out/bench_bf16.err"
echo "bench rc=$? $(cut -c1-200 "$out/${tag}_bench_bf16.json")"
python3 - "$out/${tag}_bench_bf16.json" "$out/${tag}_cpu_full_shape.json" <<'PY'
import json, sys
try:
    line = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    fs = dict(line["cpu_baseline"]["full_shape"], recipe="timit_ligru", note="measured inside the bench run of the closing pass (bench.py --cpu-full-in-run)")
    json.dump(fs, open(sys.argv[2], "w"), indent=1)
    print("cpu full shape:", fs.get("value"), fs.get("unit"), fs.get("seconds"), "s")
except Exception as e:  # noqa: BLE001
    print("no full-shape record:", e)
PY
# ---- 4, 5
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 8 --warmup 2 --prewarm-s 0 --no-cpu-baseline --no-extras"
S="python $R/bench.py --steps 2 --warmup 1 --prewarm-s 0 --no-cpu-baseline --no-extras"
timeout 400 rocprofv3 --kernel-trace --stats -d $out/kt -- $B > $out/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) $out/${tag}_bench_bf16_kernel_stats.csv > /dev/null 2>&1
python $R/tools/rocpd_dump.py $(find $out/kt -name "*.db" | head -1) $out/${tag}_timeline_tail.csv 1200 > /dev/null 2>&1
python $R/tools/timeline_step.py $out/${tag}_timeline_tail.csv 3 > $out/${tag}_timeline_step.txt 2>/dev/null
rm -rf $out/kt
head -6 $out/${tag}_bench_bf16_kernel_stats.csv
for r in timit_lstm libri_gru timit_mlp timit_sincnet; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $out/kt_$r -- $B --recipe $r > $out/kt_$r.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_$r -name "*.db" | head -1) $out/${tag}_${r}_kernel_stats.csv > /dev/null 2>&1
  rm -rf $out/kt_$r
done
timeout 200 rocprofv3 --kernel-trace --stats -d $out/kt_fp32 -- python $R/bench.py --prec fp32 --steps 3 --warmup 1 --prewarm-s 0 --no-cpu-baseline --no-extras > $out/kt_fp32.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt_fp32 -name "*.db" | head -1) $out/${tag}_bench_fp32_kernel_stats.csv > /dev/null 2>&1
rm -rf $out/kt_fp32
for r in timit_lstm libri_gru; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_fp32_$r -- python $R/bench.py --recipe $r --prec fp32 --steps 3 --warmup 1 --prewarm-s 0 --no-cpu-baseline --no-extras > $out/kt_fp32_$r.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_fp32_$r -name "*.db" | head -1) $out/${tag}_${r}_fp32_kernel_stats.csv > /dev/null 2>&1
  rm -rf $out/kt_fp32_$r
done
cd $R
timeout 300 python tools/trace_rec4.py > $out/${tag}_fp32_gen4_phase_trace.json 2> $out/trace_rec4.err
# ---- 7
# (round 5: the full-shape comparison runs inside the GPU suite above - tests/test_gpu_full_shape.py; PK_FULL_SHAPE_JSON keeps its record)
python3 tools/jget.py $out/${tag}_full_shape_parity.json pass loss_rel_diff model_step_seconds grad_rel_err_worst 2>/dev/null
ls $out | head -60
