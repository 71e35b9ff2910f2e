#!/bin/bash
# round 6, call j: exact-fp32 weight-gradient GEMMs on the side stream - bit-identity with the autograd route, everything that
# trains in fp32 through the fused optimizers, then the A/B on the three recurrent recipes in one process each, round-robin
set -u
out=gpurun_out/r06j; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 900 python -m pytest tests/test_gpu_round6.py -q -m gpu > "$out/pytest_round6.txt" 2>&1; echo "round6 rc=$?"; tail -5 "$out/pytest_round6.txt"
timeout 2400 python -m pytest tests/test_gpu_reference_pins.py tests/test_gpu_dp_two_ranks.py tests/test_gpu_dp_run_nn.py tests/test_gpu_full_shape.py -q -m gpu > "$out/pytest_fp32_users.txt" 2>&1; echo "fp32 users rc=$?"; tail -5 "$out/pytest_fp32_users.txt"
for rnd in 1 2; do
  for rcp in timit_lstm libri_gru timit_ligru; do
    for sw in 1 0; do
      PK_EXPERIMENT="f32_wgrad_side=$sw" timeout 600 python bench.py --recipe $rcp --prec fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$out/bench_${rcp}_side${sw}_$rnd.json" 2> "$out/bench_${rcp}_side${sw}_$rnd.err"
      echo "$rcp side=$sw round $rnd: $(python tools/jget.py "$out/bench_${rcp}_side${sw}_$rnd.json" ms_per_step 2>/dev/null)"
    done
  done
done
