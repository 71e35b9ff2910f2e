#!/bin/bash
mkdir -p gpurun_out/head
python -m pytest tests/test_gpu_kernels.py -q -x -k "head_nll or fused_output or logsoftmax" 2>&1 | tail -5
python -m pytest tests/test_gpu_reference_pins.py tests/test_gpu_parity.py tests/test_core_chunk.py tests/test_gpu_dp_two_ranks.py -q -x -m gpu 2>&1 | tail -3
python bench.py --no-extras --steps 40 2>gpurun_out/head/bench2.err | tee gpurun_out/head/bench2.json | cut -c1-300
