#!/bin/bash
# Role-split recurrences, second look: is the I/O wave what the step waits for?  (slot 7 of the phase trace = its arrival at
# the step's barrier; NP = 0)  Plus: which host op launches which kernel in an eager timit_mlp / timit_sincnet step.
set -u
tag=${1:-r04e}
out=gpurun_out/$tag
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
PK_REC_GEN=5 PK_SPLIT_POLLERS=0 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "(bf16_persistent_matches and 550) or full_geometry or (dirty_buffer and 550)" > "$out/pytest_np0.log" 2>&1
echo "pytest NP=0 rc=$? $(tail -1 "$out/pytest_np0.log")"
for np in 0 3; do
  PK_REC_GEN=5 PK_SPLIT_POLLERS=$np JSON_OUT="$out/trace_np$np.json" timeout 120 python tools/trace_rec2.py > "$out/trace_np$np.log" 2>&1
  echo "split NP=$np: $(grep -vE 'amdgpu' "$out/trace_np$np.log" | tr '\n' ' ' | tr -s ' ' | cut -c1-1100)"
done
PK_REC_GEN=5 PK_SPLIT_POLLERS=0 EMPTY=1 JSON_OUT="$out/trace_np0_empty.json" timeout 120 python tools/trace_rec2.py > "$out/trace_np0_empty.log" 2>&1
echo "NP=0 empty: $(grep -vE 'amdgpu' "$out/trace_np0_empty.log" | tr '\n' ' ' | tr -s ' ' | cut -c1-1100)"
JSON_OUT="$out/trace_default.json" timeout 120 python tools/trace_rec2.py > "$out/trace_default.log" 2>&1
echo "default gens: $(grep -E 'cycles/step|launch ms' "$out/trace_default.log" | tr '\n' ' ' | cut -c1-300)"
for i in 1 2; do
  for v in "PK_REC_GEN=0" "PK_REC_GEN=5 PK_SPLIT_POLLERS=0" "PK_REC_GEN=5 PK_SPLIT_POLLERS=3"; do
    ms=$(env $v timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 --prewarm-s 0.5 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step loss_final)
    echo "$v  $ms" | tee -a "$out/ab.txt"
  done
done
for r in timit_mlp timit_sincnet; do
  timeout 200 python tools/step_ops_profile.py $r > "$out/ops_$r.txt" 2> "$out/ops_$r.err"
  echo "ops $r: $(grep -c ' us ' "$out/ops_$r.txt") kernels"
done
