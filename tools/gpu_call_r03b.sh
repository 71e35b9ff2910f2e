#!/bin/bash
# round 3, call b: third-generation recurrences - parity suite, A/B against the second generation, poll-delay sweeps
out=gpurun_out/r03b
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_dp_two_ranks.py > $out/pytest_gpu.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest_gpu.log)"
bash tools/gpu_ab3.sh 2 PK_REC_GEN=2 PK_REC_GEN=3
bash tools/gpu_ab3.sh 1 PK_POLL_DELAY_FWD=0 PK_POLL_DELAY_FWD=1 PK_POLL_DELAY_BWD=0 PK_POLL_DELAY_BWD=1 PK_POLL_DELAY_BWD=3
bash tools/gpu_ab_recipe.sh libri_gru 1 30 PK_POLL_DELAY_BWD=0 PK_POLL_DELAY_BWD=2 PK_POLL_DELAY_FWD=0 PK_POLL_DELAY_FWD=1
bash tools/gpu_ab_recipe.sh timit_lstm 1 30 PK_POLL_DELAY_FWD=0 PK_POLL_DELAY_FWD=1 PK_POLL_DELAY_FWD=2
