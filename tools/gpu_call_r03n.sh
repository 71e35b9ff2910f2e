#!/bin/bash
out=$PWD/gpurun_out/r03n
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -q -m gpu -x \
  -k "gemm or mlp or MLP or linear or head or nll or logsoftmax or sincnet or e2e or hip_graph or fused" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
grep -E "FAILED|Error" $out/pytest.log | head
bash tools/gpu_ab_recipe.sh timit_mlp 2 400 "PK_GEMM_SKINNY=0 PK_MLP_FUSED=0" PK_GEMM_SKINNY=1
bash tools/gpu_ab_recipe.sh timit_sincnet 2 100 "PK_GEMM_SKINNY=0 PK_MLP_FUSED=0" PK_GEMM_SKINNY=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 150 rocprofv3 --kernel-trace -d $out/kt -- python $R/bench.py --recipe timit_mlp --steps 60 --warmup 5 --no-cpu-baseline --no-extras > $out/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) $out/mlp_stats.csv; rm -rf $out/kt
head -14 $out/mlp_stats.csv | cut -c1-150
