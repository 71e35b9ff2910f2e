#!/bin/bash
# round 5, call R: the closing records on the final tree (reference mask stream by default, fp32 GEMM change):
# the driver's command, the default bench line, the kernel trace of the headline
set -u
tag=r05
out=$PWD/gpurun_out/r05r; mkdir -p "$out"; R=$PWD
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/driver_cmd_line.json" 2> "$out/driver_cmd.err"
echo "driver command, first process: $(python3 tools/jget.py $out/driver_cmd_line.json ms_per_step value config.mask_rng roofline.frac)"
PK_BENCH_VERBOSE=1 timeout 1200 python bench.py > "$out/${tag}_bench_bf16.json" 2> "$out/bench_bf16.err"
echo "bench rc=$? $(python3 tools/jget.py $out/${tag}_bench_bf16.json ms_per_step value parity_mode.ms_per_step cpu_baseline.value)"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 8 --warmup 2 --prewarm-s 0 --no-cpu-baseline --no-extras"
timeout 400 rocprofv3 --kernel-trace --stats -d $out/kt -- $B > $out/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) $out/${tag}_bench_bf16_kernel_stats.csv > /dev/null 2>&1
python $R/tools/rocpd_dump.py $(find $out/kt -name "*.db" | head -1) $out/${tag}_timeline_tail.csv 1200 > /dev/null 2>&1
python $R/tools/timeline_step.py $out/${tag}_timeline_tail.csv 3 > $out/${tag}_timeline_step.txt 2>/dev/null
rm -rf $out/kt $out/${tag}_timeline_tail.csv
head -8 $out/${tag}_bench_bf16_kernel_stats.csv | cut -c1-160
