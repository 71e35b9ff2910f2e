#!/bin/bash
out=$PWD/gpurun_out/r03o
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -x \
  -k "gemm or mlp or MLP or linear or head or sincnet or e2e or hip_graph or fused" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "FAILED|Error" $out/pytest.log | head
bash tools/gpu_ab_recipe.sh timit_mlp 2 400 "PK_GEMM_SKINNY=0 PK_MLP_FUSED=0" PK_GEMM_SKINNY=1
bash tools/gpu_ab_recipe.sh timit_sincnet 2 100 "PK_GEMM_SKINNY=0 PK_MLP_FUSED=0" PK_GEMM_SKINNY=1
