#!/bin/bash
python -m pytest tests/test_gpu_kernels.py -q -x -k "head_nll or fused_output or logsoftmax or twin" 2>&1 | tail -3
python -m pytest tests/test_gpu_reference_pins.py -q -x 2>&1 | tail -2
for i in 1 2; do python bench.py --no-extras --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['entry_points_ms_per_step'])"; done
