#!/bin/bash
# targeted check of the arg-max cost path: head tests, end-to-end fixtures, graph replay; then a round-robin A/B
set -u
tag=${1:-r04r}
out=gpurun_out/$tag
mkdir -p "$out"
if ! timeout 90 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py tests/test_core_chunk.py -q -m gpu --maxfail=8 \
    -k "head or cost or golden or graph or softmax or output_layer or recipe or e2e or trajectory or run_nn" > "$out/pytest_sel.log" 2>&1
echo "pytest rc=$? $(tail -1 "$out/pytest_sel.log")"
grep -E "^(FAILED|ERROR)" "$out/pytest_sel.log" | head -10
for i in 1 2 3; do for v in 1 0; do
  ms=$(PK_HEAD_ARGMAX=$v timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step step_ms.median)
  echo "PK_HEAD_ARGMAX=$v $ms" | tee -a "$out/ab.txt"
done; done
for v in 1 0; do
  ms=$(PK_HEAD_ARGMAX=$v timeout 200 python bench.py --recipe timit_mlp --steps 400 --warmup 5 --repeats 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step)
  echo "timit_mlp PK_HEAD_ARGMAX=$v $ms" | tee -a "$out/ab.txt"
done
