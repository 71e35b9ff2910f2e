#!/bin/bash
set -u
out=$PWD/gpurun_out/r02e
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $out/pmc_sq -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py > $out/pmc_sq.log 2>&1
echo "pmc rc=$?"
db=$(find $out/pmc_sq -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $db $out/pmc_sq.csv; grep -i gemm $out/pmc_sq.csv | head -80
REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d $out/pmc_tcc -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py > $out/pmc_tcc.log 2>&1
db=$(find $out/pmc_tcc -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $db $out/pmc_tcc.csv; grep -i gemm $out/pmc_tcc.csv | head -40
