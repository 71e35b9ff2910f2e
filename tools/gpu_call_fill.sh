#!/bin/bash
python -m pytest tests/test_gpu_parity.py -q -x -k "self_filled or persistent_matches" 2>&1 | tail -4
python -m pytest tests/test_gpu_reference_pins.py -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh PK_REC_SELF_FILL=0 PK_REC_SELF_FILL=1 2
