#!/bin/bash
# Small-batch (launch-bound) recipes: kernel tests of the round-4 small-batch kernels, the MLP / SincNet parity tests, A/B of
# timit_mlp / timit_sincnet with and without the split reductions, the op -> kernel map.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu_mlp_eval.sh r04g'
set -u
tag=${1:-r04g}
out=gpurun_out/$tag
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "gemm_bf16 or small_batch or bn_act_bwd_small or output_layer or head_nll or linear_autograd" > "$out/pytest_kernels.log" 2>&1
echo "pytest kernels rc=$? $(tail -1 "$out/pytest_kernels.log")"; grep -E "^FAILED|^E  " "$out/pytest_kernels.log" | head -8
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -x -k "mlp or MLP or small_batch or sincnet or hip_graph" > "$out/pytest_mlp.log" 2>&1
echo "pytest mlp rc=$? $(tail -1 "$out/pytest_mlp.log")"; grep -E "^FAILED|^E  " "$out/pytest_mlp.log" | head -8
for i in 1 2; do
  for v in "PK_SMALL_SPLITK=0" "PK_SMALL_SPLITK=1"; do
    for r in timit_mlp timit_sincnet; do
      st=400; [ $r = timit_sincnet ] && st=100
      ms=$(env $v timeout 200 python bench.py --recipe $r --steps $st --warmup 5 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step regions_ms_per_step)
      echo "$v $r $ms" | tee -a "$out/ab.txt"
    done
  done
done
timeout 200 python tools/step_ops_profile.py timit_mlp > "$out/ops_timit_mlp.txt" 2> "$out/ops_timit_mlp.err"
echo "ops timit_mlp: $(grep -c ' us ' "$out/ops_timit_mlp.txt") kernels"
