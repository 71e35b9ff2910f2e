#!/bin/bash
set -u
out=$PWD/gpurun_out/r03m
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for v in "PK_GEMM_STAGES=0" "PK_GEMM_STAGES=2 PK_MLP_FUSED=0"; do
  env $v timeout 150 rocprofv3 --kernel-trace -d $out/kt_$i -- python $R/bench.py --recipe timit_mlp --steps 60 --warmup 5 --no-cpu-baseline --no-extras > $out/kt_$i.log 2>&1
  db=$(find $out/kt_$i -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $db $out/mlp_${i}_stats.csv
  rm -rf $out/kt_$i
  echo "== $v"; head -22 $out/mlp_${i}_stats.csv | cut -c1-150
  i=$((i+1))
done
