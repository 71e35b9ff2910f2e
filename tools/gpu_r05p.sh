#!/bin/bash
# round 5, call P: kernel statistics of the exact-fp32 mode after the GEMM change (Li-GRU headline, GRU, LSTM)
set -u
out=$PWD/gpurun_out/r05p; mkdir -p "$out"; R=$PWD
cd /tmp && export TMPDIR=/tmp
for r in timit_ligru libri_gru timit_lstm; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_$r -- python $R/bench.py --recipe $r --prec fp32 --steps 3 --warmup 1 --prewarm-s 0 --no-cpu-baseline --no-extras > $out/kt_$r.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_$r -name "*.db" | head -1) $out/r05_${r}_fp32_kernel_stats.csv > /dev/null 2>&1
  rm -rf $out/kt_$r
  echo "== $r"; head -9 $out/r05_${r}_fp32_kernel_stats.csv | cut -c1-150
done
