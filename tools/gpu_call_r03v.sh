#!/bin/bash
out=$PWD/gpurun_out/r03v
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu \
  -k "conv or cnn or CNN or sinc or Sinc" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "^FAILED|^ERROR|Error" $out/pytest.log | head -30
bash tools/gpu_ab_recipe.sh timit_sincnet 2 100 PK_CONV_BF16=0 PK_CONV_BF16=1
