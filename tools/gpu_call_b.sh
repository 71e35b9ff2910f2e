#!/bin/bash
set -u
out=gpurun_out/r02b
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_reference_pins.py tests/test_core_chunk.py tests/test_gpu_dp_two_ranks.py -q -m gpu -s > $out/pins.log 2>&1
echo "pins rc=$? $(tail -1 $out/pins.log)"
timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_gpu_reference_pins.py --deselect tests/test_gpu_dp_two_ranks.py > $out/suite.log 2>&1
echo "suite rc=$? $(tail -1 $out/suite.log)"
PK_BENCH_VERBOSE=1 timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$? $(cut -c1-200 $out/bench.json)"
JSON_OUT=$out/trace_full.json timeout 120 python tools/trace_rec2.py > $out/trace_full.log 2>&1
EMPTY=1 JSON_OUT=$out/trace_empty.json timeout 120 python tools/trace_rec2.py > $out/trace_empty.log 2>&1
echo "trace rc=$?"; tail -3 $out/trace_empty.log
timeout 120 python tools/bench_gemm.py > $out/gemm.log 2>&1; cat $out/gemm.log
