#!/bin/bash
out=$PWD/gpurun_out/timeline
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $out/kt -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $out/kt.log 2>&1
python $R/tools/rocpd_dump.py $(find $out/kt -name "*.db" | head -1) $out/tail.csv 2600
rm -rf $out/kt
ls -la $out
