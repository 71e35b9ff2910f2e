#!/bin/bash
set -u
out=gpurun_out/r02d
mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm or linear" > $out/gemm_tests.log 2>&1
echo "gemm tests rc=$? $(tail -1 $out/gemm_tests.log)"
timeout 200 python tools/bench_gemm.py > $out/gemm.log 2>&1; cat $out/gemm.log
