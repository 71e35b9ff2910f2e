#!/bin/bash
out=$PWD/gpurun_out/r03y2
mkdir -p $out
if ! timeout 60 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
  echo "BAD BOX: first GPU touch failed"; exit 0
fi
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -s \
  -k "conv or cnn or CNN or sinc or Sinc" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "^FAILED|^ERROR|Error|\[bf16\]" $out/pytest.log | cut -c1-600 | head -20

