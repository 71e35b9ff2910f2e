#!/bin/bash
set -u
out=gpurun_out/r02g
mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_dp_two_ranks.py > $out/suite.log 2>&1
echo "suite rc=$? $(tail -1 $out/suite.log)"
timeout 600 python -m pytest tests/test_gpu_dp_two_ranks.py -q -m gpu -s > $out/dp.log 2>&1
echo "dp rc=$? $(tail -1 $out/dp.log)"
grep -E "ran over|first-step|parameters after" $out/dp.log | head -20
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$? $(cut -c1-160 $out/bench.json)"
