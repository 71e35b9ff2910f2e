#!/bin/bash
out=$PWD/gpurun_out/r03x2
mkdir -p $out
if ! timeout 60 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
  echo "BAD BOX: first GPU touch failed"; exit 0
fi
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "conv_pool" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "^FAILED|^ERROR|Error" $out/pytest.log | head
bash tools/gpu_ab_recipe.sh timit_sincnet 1 100 PK_CONV_BF16=0 PK_CONV_BF16=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PK_CONV_BF16=1 timeout 100 rocprofv3 --kernel-trace -d $out/kt -- python $R/bench.py --recipe timit_sincnet --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) $out/sinc_stats.csv; rm -rf $out/kt
head -7 $out/sinc_stats.csv | cut -c1-150
