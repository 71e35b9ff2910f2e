#!/bin/bash
out=$PWD/gpurun_out/r03t
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -x \
  -k "small_batch or grouped_dropout or mlp or MLP or linear or head or sincnet or e2e or hip_graph or graph or fused or cnn" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "FAILED|Error" $out/pytest.log | head
bash tools/gpu_ab_recipe.sh timit_mlp 2 400 "PK_DIRECT_GRADS=0 PK_MLP_FUSED_BWD=0" PK_DIRECT_GRADS=1
bash tools/gpu_ab_recipe.sh timit_sincnet 2 100 "PK_DIRECT_GRADS=0 PK_MLP_FUSED_BWD=0" PK_DIRECT_GRADS=1
python bench.py --recipe timit_mlp --no-extras --no-cpu-baseline --steps 400 --warmup 5 2>&1 | tail -3 | cut -c1-400
