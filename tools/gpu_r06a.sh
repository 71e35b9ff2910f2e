#!/bin/bash
# round 6, call a: the parity items of the round-5 review on hardware + this round's baseline on one box.
set -u
out=gpurun_out/r06a; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
(nproc; lscpu | grep -E "Model name") > "$out/box.txt" 2>&1
export PK_FULL_SHAPE_JSON=$PWD/$out/r06_full_shape_parity.json
timeout 1500 python -m pytest tests/test_gpu_full_shape.py tests/test_gpu_round6.py -x -q -m gpu -s > "$out/pytest_new.txt" 2>&1; echo "new tests rc=$?"; tail -5 "$out/pytest_new.txt"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "head or share or single_hip or output_layers" > "$out/pytest_heads.txt" 2>&1; echo "head tests rc=$?"; tail -3 "$out/pytest_heads.txt"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$out/drv_1.json" 2> "$out/drv_1.err"; echo "drv: $(python3 tools/jget.py "$out/drv_1.json" ms_per_step)"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$out/drv_2.json" 2> "$out/drv_2.err"; echo "drv: $(python3 tools/jget.py "$out/drv_2.json" ms_per_step)"
