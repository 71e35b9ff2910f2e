#!/bin/bash
# round 5, call E: one dX GEMM for the heads on a shared input (tests + headline A/B), host-op map of the MLP step
set -u
out=$PWD/gpurun_out/r05e
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
timeout 200 python tools/step_ops_profile.py timit_mlp > "$out/ops_timit_mlp.txt" 2> "$out/ops_timit_mlp.err"; echo "ops map rc=$? lines $(wc -l < $out/ops_timit_mlp.txt)"
timeout 900 python -m pytest tests/test_gpu_reference_pins.py tests/test_gpu_kernels.py -q -m gpu -x -k "two_heads or e2e or config_scale or ce_loss or recipe_scale or head or nll" > "$out/pytest_heads.log" 2>&1; echo "head tests rc=$? $(tail -1 $out/pytest_heads.log)"; grep -E "^FAILED|^E  " "$out/pytest_heads.log" | head -8 | cut -c1-300
for i in 1 2 3; do for v in "PK_EXPERIMENT=head_dx_cat=0" "PK_EXPERIMENT=head_dx_cat=1"; do
  ms=$(env $v timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "$v headline $ms" | tee -a "$out/ab.txt"
done; done
timeout 900 python -m pytest tests/test_gpu_dp_two_ranks.py tests/test_gpu_full_shape.py -q -m gpu -s -k "launch_bound or full_headline or forced" > "$out/pytest_dp.log" 2>&1; echo "dp / full shape rc=$? $(tail -1 $out/pytest_dp.log)"; grep -E "plain graph|full shape:|^FAILED|^E  " "$out/pytest_dp.log" | head -8 | cut -c1-400
