#!/bin/bash
# The last call of a round, on the final tree (after gpu_evidence.sh has produced the PMC / trace records earlier): the driver's
# command as the first process, the whole GPU suite, smoke(), the default bench line with the CPU step at the full shape, and
# the kernel traces of the exact-fp32 rows.   gpurun --timeout 3600 -- 'bash tools/gpu_final.sh r06'
set -u
tag=${1:-rXX}
out=$PWD/gpurun_out/${tag}fin
R=$PWD
mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --step-trace > "$out/driver_cmd_first.json" 2> "$out/driver_cmd_first.err"
echo "driver command, first process: $(python3 tools/jget.py "$out/driver_cmd_first.json" ms_per_step value)"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --step-trace > "$out/driver_cmd_second.json" 2> "$out/driver_cmd_second.err"
echo "driver command, second process: $(python3 tools/jget.py "$out/driver_cmd_second.json" ms_per_step value)"
PK_FULL_SHAPE_JSON=$out/${tag}_full_shape_parity.json timeout 2400 python -m pytest tests -q -m gpu > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$? $(tail -1 "$out/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)" "$out/pytest_gpu.log" | head -10
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1
echo "smoke rc=$? $(tail -2 "$out/smoke.log" | tr '\n' ' ')"
PK_BENCH_VERBOSE=1 timeout 2400 python bench.py --cpu-full-in-run > "$out/${tag}_bench_bf16.json" 2> "$out/bench_bf16.err"
echo "bench rc=$? $(cut -c1-160 "$out/${tag}_bench_bf16.json")"
cd /tmp && export TMPDIR=/tmp
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
ls $out
