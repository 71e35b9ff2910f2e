#!/bin/bash
# round 3, closing evidence pass on the final tree: whole GPU suite + smoke + default bench line + its kernel trace
# (gpu_round.sh), kernel traces of the other four recipes
bash tools/gpu_round.sh r03z
out=$PWD/gpurun_out/r03z
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
for r in timit_lstm libri_gru timit_mlp timit_sincnet; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $out/kt_$r -- $B --recipe $r > $out/kt_$r.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_$r -name "*.db" | head -1) $out/r03_${r}_kernel_stats.csv
  rm -rf $out/kt_$r
done
ls $out
