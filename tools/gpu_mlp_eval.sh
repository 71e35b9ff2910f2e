#!/bin/bash
# Small-batch (launch-bound) recipes: kernel tests of the round-4 small-batch kernels / fused step, the MLP / SincNet parity
# tests, A/B of timit_mlp / timit_sincnet, kernel trace of the replayed timit_mlp step.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu_mlp_eval.sh r04h'
set -u
tag=${1:-r04h}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x > "$out/pytest_kernels.log" 2>&1
echo "pytest kernels rc=$? $(tail -1 "$out/pytest_kernels.log")"; grep -E "^FAILED|^E  " "$out/pytest_kernels.log" | head -8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py tests/test_core_chunk.py tests/test_gpu_dp_two_ranks.py -q -m gpu -x -k "mlp or MLP or small_batch or sincnet or hip_graph or recipe_scale or chunk or reducer or two_ranks or e2e or fused" > "$out/pytest_mlp.log" 2>&1
echo "pytest mlp rc=$? $(tail -1 "$out/pytest_mlp.log")"; grep -E "^FAILED|^E  " "$out/pytest_mlp.log" | head -8
for i in 1 2; do
  for v in "PK_SMALL_SPLITK=0 PK_FUSED_STEP=0" "PK_SMALL_SPLITK=1 PK_FUSED_STEP=0" "PK_SMALL_SPLITK=1 PK_FUSED_STEP=1"; do
    for r in timit_mlp timit_sincnet; do
      st=400; [ $r = timit_sincnet ] && st=100
      ms=$(env $v timeout 200 python bench.py --recipe $r --steps $st --warmup 5 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step regions_ms_per_step loss_final)
      echo "$v $r $ms" | tee -a "$out/ab.txt"
    done
  done
done
R=$PWD
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$out/prof" -- python "$R/bench.py" --recipe timit_mlp --steps 200 --warmup 5 --no-cpu-baseline --no-extras > "$out/prof_mlp.log" 2>&1 )
db=$(find "$out/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" "$out/r04_timit_mlp_kernel_stats.csv" > /dev/null 2> "$out/kstats.err"; head -30 "$out/r04_timit_mlp_kernel_stats.csv" | cut -c1-150; rm -rf "$out/prof"; fi
python bench.py --steps 40 --no-extras --no-cpu-baseline > "$out/headline.json" 2> "$out/headline.err"; echo "headline: $(python3 tools/jget.py "$out/headline.json" ms_per_step loss_final)"
