#!/bin/bash
# round 5, call K: the device mirror of the reference's mask stream after the device-normalisation fix
set -u
out=$PWD/gpurun_out/r05k; mkdir -p "$out"
timeout 300 python tools/diag_ref_rng.py 2>&1 | tee "$out/diag.txt" | tail -9
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "reference_mask" 2>&1 | tail -2
