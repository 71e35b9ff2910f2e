#!/bin/bash
out=gpurun_out/r03k
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -x \
  -k "dpp_row_sum or layernorm or _ln or ln_ or mixed_stack" -s > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "per-step LayerNorm|FAILED|Error|error" $out/pytest.log | head -40
timeout 300 python tools/time_ln_rec.py > $out/ln_times.json 2> $out/ln_times.err; tail -2 $out/ln_times.json; tail -3 $out/ln_times.err
