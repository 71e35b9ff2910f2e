#!/bin/bash
mkdir -p gpurun_out/final2
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/final2/pytest_gpu.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/final2/pytest_gpu.log)"
bash tools/gpu_ab3.sh 2 PK_FLAT_GROUPS=0 PK_FLAT_GROUPS=1
