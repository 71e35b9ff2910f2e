#!/bin/bash
# round 5, call B: wave-role helpers (sweep, headline + LSTM / GRU recipe A/B), then the whole GPU suite on the pruned tree.
set -u
out=$PWD/gpurun_out/r05b
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
ROUNDS=2 timeout 300 python tools/helper_sweep.py "$out/helper_ligru.json" > "$out/helper_ligru.log" 2>&1; echo "sweep liGRU rc=$?"; grep -E "fwd|Error|error" "$out/helper_ligru.log" | head -20
KIND=LSTM ROUNDS=2 CONFIGS="P lead3 x4;P+out lead3/1 x4;bwd lead4 x4;all lead3/1/2 x4" timeout 300 python tools/helper_sweep.py "$out/helper_lstm.json" > "$out/helper_lstm.log" 2>&1; echo "sweep LSTM rc=$?"; grep -E "fwd|Error|error" "$out/helper_lstm.log" | head
KIND=GRU ROUNDS=2 CONFIGS="P lead3 x4;P+out lead3/1 x4" timeout 300 python tools/helper_sweep.py "$out/helper_gru.json" > "$out/helper_gru.log" 2>&1; echo "sweep GRU rc=$?"; grep -E "fwd|Error|error" "$out/helper_gru.log" | head
for i in 1 2; do for v in PK_REC_HELPER=0 PK_REC_HELPER=1 PK_REC_HELPER=3 PK_REC_HELPER=7; do
  ms=$(env $v timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "$v headline $ms" | tee -a "$out/ab.txt"
done; done
for r in timit_lstm libri_gru; do for v in PK_REC_HELPER=0 PK_REC_HELPER=3 PK_REC_HELPER=7; do
  ms=$(env $v timeout 200 python bench.py --recipe $r --no-extras --no-cpu-baseline --steps 30 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "$v $r $ms" | tee -a "$out/ab.txt"
done; done
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$? $(tail -1 "$out/pytest_gpu.log")"
grep -E "^(FAILED|ERROR)" "$out/pytest_gpu.log" | head -12 | cut -c1-250
