#!/bin/bash
out=gpurun_out/r03h
mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -k "mlp or MLP or scale_sincnet or hip_graph or e2e or model_language or fused_output or head_nll or share_their" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
bash tools/gpu_ab_recipe.sh timit_mlp 2 400 PK_MLP_FUSED=0 PK_MLP_FUSED=1
bash tools/gpu_ab_recipe.sh timit_sincnet 2 100 PK_MLP_FUSED=0 PK_MLP_FUSED=1
