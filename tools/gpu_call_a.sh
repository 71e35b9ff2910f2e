#!/bin/bash
# r02 first GPU pass: the new reference pins, then the whole GPU suite, then the default bench line.
set -u
out=gpurun_out/r02a
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_reference_pins.py tests/test_core_chunk.py -q -m gpu -s > $out/pins.log 2>&1
echo "pins rc=$? $(tail -1 $out/pins.log)"
timeout 500 python -m pytest tests/test_gpu_dp_two_ranks.py -q -m gpu -s > $out/dp2.log 2>&1
echo "dp2 rc=$? $(tail -1 $out/dp2.log)"
timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_gpu_reference_pins.py --deselect tests/test_gpu_dp_two_ranks.py > $out/suite.log 2>&1
echo "suite rc=$? $(tail -1 $out/suite.log)"
timeout 300 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$? $(cut -c1-300 $out/bench.json)"
