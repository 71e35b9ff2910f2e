#!/bin/bash
python -m pytest tests/test_gpu_kernels.py -q -x -k "projection_gemm or batchnorm or head_nll or fused_output" 2>&1 | tail -4
python -m pytest tests/test_gpu_reference_pins.py tests/test_gpu_parity.py -q -x -k "bf16" 2>&1 | tail -3
bash tools/gpu_ab.sh PK_GEMM_STATS=0 PK_GEMM_STATS=1 2
