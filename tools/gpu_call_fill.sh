#!/bin/bash
python -m pytest tests/test_gpu_parity.py -q -x -k "self_filled or persistent_matches" 2>&1 | tail -4
mkdir -p gpurun_out/fill
for rcp in timit_lstm libri_gru; do
for sf in 0 1; do
  ms=$(PK_REC_SELF_FILL=$sf python bench.py --no-extras --steps 20 --recipe $rcp 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$rcp PK_REC_SELF_FILL=$sf ms_per_step $ms" | tee -a gpurun_out/fill/log.txt
done; done
