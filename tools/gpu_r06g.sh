#!/bin/bash
# round 6, call g: forward-only chunks without saved state (A/B on one box), then the WHOLE -m gpu suite on the tree
set -u
out=gpurun_out/r06g; mkdir -p "$out"
timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))" || { echo "BAD BOX"; exit 0; }
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > "$out/pytest_round6.txt" 2>&1; echo "round6 rc=$?"; tail -3 "$out/pytest_round6.txt"
for rep in 1 2; do
  for v in 1 0; do
    for r in timit_ligru timit_lstm libri_gru; do
      PK_EXPERIMENT=fwd_nosave=$v timeout 300 python3 bench.py --recipe $r --only-forward-mode > "$out/fwd_${r}_nosave${v}_$rep.json" 2> "$out/fwd_${r}_nosave${v}_$rep.err"
      echo "$r forward mode nosave=$v #$rep: $(python3 tools/jget.py "$out/fwd_${r}_nosave${v}_$rep.json" ms_per_step loss_final)"
    done
  done
done
timeout 2400 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu_all.txt" 2>&1; echo "suite rc=$?"; tail -4 "$out/pytest_gpu_all.txt"
