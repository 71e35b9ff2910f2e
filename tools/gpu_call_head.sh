#!/bin/bash
mkdir -p gpurun_out/head
python -m pytest tests/test_gpu_kernels.py -q -x -k "fused_output or logsoftmax or linear_autograd" 2>&1 | tail -5
python -m pytest tests/test_gpu_reference_pins.py tests/test_gpu_parity.py -q -x -k "bf16" 2>&1 | tail -3
for rs in 0 1; do
  PK_GEMM_TILE_ROWS=$rs python bench.py --no-extras --steps 30 2>gpurun_out/head/bench_rs$rs.err | tee gpurun_out/head/bench_rs$rs.json | cut -c1-400
done
