#!/bin/bash
# Role-split recurrences, third look: issue priorities (compute waves 3, helpers 0; PK_REC_FLUSH_LATE=1 = no priorities) and
# the phase trace of a compute wave that does NOT share its SIMD with the I/O wave.  Plus the host op -> kernel maps.
set -u
tag=${1:-r04f}
out=gpurun_out/$tag
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
for np in 0 3; do for prio in 0 1; do for w in 0 1; do
  PK_REC_GEN=5 PK_SPLIT_POLLERS=$np PK_REC_FLUSH_LATE=$prio PK_TRACE_WAVE=$w JSON_OUT="$out/trace_np${np}_noprio${prio}_w$w.json" timeout 120 python tools/trace_rec2.py > "$out/trace_np${np}_noprio${prio}_w$w.log" 2>&1
  echo "NP=$np noprio=$prio wave=$w: $(grep -vE 'amdgpu' "$out/trace_np${np}_noprio${prio}_w$w.log" | tr '\n' ' ' | tr -s ' ' | cut -c1-1100)"
done; done; done
for i in 1 2; do
  for v in "PK_REC_GEN=0" "PK_REC_GEN=5 PK_SPLIT_POLLERS=0" "PK_REC_GEN=5 PK_SPLIT_POLLERS=3"; do
    ms=$(env $v timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 --prewarm-s 0.5 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step loss_final)
    echo "$v  $ms" | tee -a "$out/ab.txt"
  done
done
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -x -k "mlp or MLP or small_batch or sincnet" > "$out/pytest_mlp.log" 2>&1
echo "pytest mlp rc=$? $(tail -1 "$out/pytest_mlp.log")"
for r in timit_mlp timit_sincnet; do
  timeout 200 python tools/step_ops_profile.py $r > "$out/ops_$r.txt" 2> "$out/ops_$r.err"
  echo "ops $r: $(grep -c ' us ' "$out/ops_$r.txt") kernels; $(tail -2 "$out/ops_$r.err" | tr '\n' ' ' | cut -c1-200)"
done
python bench.py --recipe timit_mlp --steps 400 --warmup 5 --repeats 3 --no-extras --no-cpu-baseline > "$out/mlp.json" 2> "$out/mlp.err"
echo "timit_mlp: $(python3 tools/jget.py "$out/mlp.json" ms_per_step regions_ms_per_step)"
