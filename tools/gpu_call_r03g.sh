#!/bin/bash
out=gpurun_out/r03g
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lstm_waves.py tests/test_gpu_reference_pins.py -q -m gpu -x -k "LSTM or lstm or GRU or gru or mingru or minimalGRU" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
bash tools/gpu_ab_recipe.sh timit_lstm 2 30 PK_LSTM_BWD_GEN=2 PK_LSTM_BWD_GEN=3
bash tools/gpu_ab_recipe.sh libri_gru 2 30 PK_GRU_BWD_GEN=2 PK_GRU_BWD_GEN=3
bash tools/gpu_ab_recipe.sh timit_mlp 2 400 PK_HEAD_DX_SHARE=1 PK_HEAD_DX_SHARE=0
bash tools/gpu_ab_recipe.sh timit_sincnet 1 100 PK_HEAD_DX_SHARE=1 PK_HEAD_DX_SHARE=0
