#!/bin/bash
set -u
out=gpurun_out/r02c
mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" > $out/gemm_tests.log 2>&1
echo "gemm tests rc=$? $(tail -1 $out/gemm_tests.log)"
timeout 200 python tools/bench_gemm.py > $out/gemm.log 2>&1; cat $out/gemm.log
timeout 300 python tools/dp_diag.py > $out/dp_diag.log 2>&1; tail -8 $out/dp_diag.log
