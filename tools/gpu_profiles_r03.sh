#!/bin/bash
# Round-3 evidence pass, second half (after tools/gpu_round.sh): kernel-trace summaries of the other recipes and of the
# fp32 mode, PMC passes of the headline (FETCH_SIZE and WRITE_SIZE in SEPARATE passes; matrix-pipe busy / wait shares).
set -u
out=$PWD/gpurun_out/r03prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
S="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -- $S > $out/pmc_fetch.log 2>&1
python $R/tools/rocpd_pmc.py $(find $out/pmc_fetch -name "*.db" | head -1) $out/r03_pmc_fetch_size.csv
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -- $S > $out/pmc_write.log 2>&1
python $R/tools/rocpd_pmc.py $(find $out/pmc_write -name "*.db" | head -1) $out/r03_pmc_write_size.csv
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $out/pmc_sq -- $S > $out/pmc_sq.log 2>&1
python $R/tools/rocpd_pmc.py $(find $out/pmc_sq -name "*.db" | head -1) $out/r03_pmc_sq.csv
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_sq
for r in timit_lstm libri_gru timit_mlp timit_sincnet; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $out/kt_$r -- $B --recipe $r > $out/kt_$r.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_$r -name "*.db" | head -1) $out/r03_${r}_kernel_stats.csv
  rm -rf $out/kt_$r
done
timeout 120 rocprofv3 --kernel-trace --stats -d $out/kt_fp32 -- python $R/bench.py --prec fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/kt_fp32.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt_fp32 -name "*.db" | head -1) $out/r03_bench_fp32_kernel_stats.csv
rm -rf $out/kt_fp32
grep -E "rec[23]_" $out/r03_pmc_fetch_size.csv | head -4; grep -E "rec[23]_" $out/r03_pmc_write_size.csv | head -4
ls $out
