#!/bin/bash
out=$PWD/gpurun_out/r03u
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "conv_pool_bf16" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "^FAILED|^ERROR|Error|assert " $out/pytest.log | head -30
