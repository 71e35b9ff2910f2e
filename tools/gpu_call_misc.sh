#!/bin/bash
python -m pytest tests/test_gpu_kernels.py -q -x 2>&1 | tail -3
python -m pytest tests/test_gpu_reference_pins.py tests/test_gpu_parity.py tests/test_core_chunk.py -q -x -m gpu -k "bf16 or chunk or trajectory or e2e or scale or golden_module" 2>&1 | tail -3
python bench.py --no-extras --steps 40 2>/dev/null | cut -c1-200
