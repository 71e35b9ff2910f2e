#!/bin/bash
# Role-split recurrences (pk_rec_split.hip, PK_REC_GEN=5): parity subset, phase traces over the polling waves' delay,
# round-robin A/B of the headline step against the default generations.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_split_eval.sh r04b'
set -u
tag=${1:-r04b}
out=gpurun_out/$tag
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
# the driver's exact command, first process on the fresh box (pre-warm on by default)
true
true
PK_REC_GEN=5 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16_persistent_matches or dirty_buffer or full_geometry or full_size or bf16_mode_is_close" > "$out/pytest_split.log" 2>&1
echo "pytest split rc=$? $(tail -1 "$out/pytest_split.log")"
grep -E "FAILED|Error|error" "$out/pytest_split.log" | head -10
for d in -1 0 8 20; do
  PK_REC_GEN=5 DELAY=$d JSON_OUT="$out/trace_gen5_d$d.json" timeout 120 python tools/trace_rec2.py > "$out/trace_gen5_d$d.log" 2>&1
  echo "gen5 delay $d: $(grep -E 'cycles/step|retries|launch ms' "$out/trace_gen5_d$d.log" | tr '\n' ' ' | cut -c1-420)"
done
JSON_OUT="$out/trace_default.json" timeout 120 python tools/trace_rec2.py > "$out/trace_default.log" 2>&1
echo "default gens: $(grep -E 'cycles/step|launch ms' "$out/trace_default.log" | tr '\n' ' ' | cut -c1-300)"
PK_REC_GEN=5 EMPTY=1 JSON_OUT="$out/trace_gen5_empty.json" timeout 120 python tools/trace_rec2.py > "$out/trace_gen5_empty.log" 2>&1
echo "gen5 empty: $(grep -E 'cycles/step' "$out/trace_gen5_empty.log" | tr '\n' ' ' | cut -c1-300)"
for i in 1 2; do
  for v in "PK_REC_GEN=0" "PK_REC_GEN_FWD=5" "PK_REC_GEN_BWD=5" "PK_REC_GEN=5"; do
    ms=$(env $v timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 --prewarm-s 0.5 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step loss_final)
    echo "$v  $ms" | tee -a "$out/ab.txt"
  done
done
