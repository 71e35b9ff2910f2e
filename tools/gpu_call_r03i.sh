#!/bin/bash
# Is the one-launch MLP layer on the graph-replayed step, and where does the 0.88 ms go?
set -u
out=$PWD/gpurun_out/r03i
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in 1 0; do
  PK_MLP_FUSED=$f timeout 150 rocprofv3 --kernel-trace -d $out/kt_$f -- python $R/bench.py --recipe timit_mlp --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/kt_$f.log 2>&1
  db=$(find $out/kt_$f -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $db $out/mlp_fused${f}_stats.csv
  python $R/tools/rocpd_timeline.py $db 260 > $out/mlp_fused${f}_timeline.txt
  rm -rf $out/kt_$f
  tail -1 $out/kt_$f.log | cut -c1-300
done
head -30 $out/mlp_fused1_stats.csv
