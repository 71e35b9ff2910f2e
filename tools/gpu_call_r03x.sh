#!/bin/bash
out=$PWD/gpurun_out/r03x
mkdir -p $out
# a box whose GPU faults on its first touch (seen twice in round 3: 'Memory access fault' inside the first .cuda()) would
# burn minutes on core dumps: stop at once
if ! timeout 60 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
  echo "BAD BOX: first GPU touch failed"; exit 0
fi
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu \
  -k "conv or cnn or CNN or sinc or Sinc" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "^FAILED|^ERROR|Error" $out/pytest.log | head -30
bash tools/gpu_ab_recipe.sh timit_sincnet 1 100 PK_CONV_BF16=0 PK_CONV_BF16=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 100 rocprofv3 --kernel-trace -d $out/kt -- python $R/bench.py --recipe timit_sincnet --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) $out/sinc_stats.csv; rm -rf $out/kt
head -9 $out/sinc_stats.csv | cut -c1-150
