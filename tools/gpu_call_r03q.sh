#!/bin/bash
out=$PWD/gpurun_out/r03q
mkdir -p $out
python bench.py --recipe timit_mlp --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $out/b.log 2>&1; tail -25 $out/b.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -x \
  -k "small_batch or grouped_dropout or mlp or MLP or linear or head or sincnet or e2e or hip_graph or graph or fused or cnn" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "FAILED|Error" $out/pytest.log | head
