#!/bin/bash
set -u
out=gpurun_out/r02f
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_dp_two_ranks.py tests/test_gpu_parity.py -q -m gpu -k "two_ranks or side_stream or rccl or bf16_mode or full_size" -s > $out/dp.log 2>&1
echo "dp rc=$? $(tail -1 $out/dp.log)"
grep -E "first-step|parameters after" $out/dp.log | head
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 50 > $out/bench.json 2> $out/bench.err
echo "bench rc=$? $(cut -c1-200 $out/bench.json)"
