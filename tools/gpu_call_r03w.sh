#!/bin/bash
out=$PWD/gpurun_out/r03w
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_reference_pins.py -q -m gpu -k "sinc or Sinc or cnn" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "^FAILED|^ERROR|Error" $out/pytest.log | head -30
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 150 rocprofv3 --kernel-trace -d $out/kt -- python $R/bench.py --recipe timit_sincnet --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) $out/sinc_stats.csv; rm -rf $out/kt
head -12 $out/sinc_stats.csv | cut -c1-170
