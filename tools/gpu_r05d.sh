#!/bin/bash
# round 5, call D: gated dY (tests + headline A/B), DP-in-graph test, fp32 step-wise GRU, LSTM default helper, GEMM reference
set -u
out=$PWD/gpurun_out/r05d
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -x -k "gated_dy or conv_layer_tail" > "$out/pytest_gate.log" 2>&1; echo "gate tests rc=$? $(tail -1 $out/pytest_gate.log)"; grep -E "^FAILED|^E  " "$out/pytest_gate.log" | head -8 | cut -c1-300
for i in 1 2 3; do for v in "PK_EXPERIMENT=dy_gate=0" "PK_EXPERIMENT=dy_gate=1" "PK_EXPERIMENT=dy_gate=1,dy_gate_shift=7" "PK_EXPERIMENT=dy_gate=1,dy_gate_shift=5"; do
  ms=$(env $v timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "$v headline $ms" | tee -a "$out/ab.txt"
done; done
timeout 900 python -m pytest tests/test_gpu_dp_two_ranks.py -q -m gpu -s -k "launch_bound" > "$out/pytest_dpgraph.log" 2>&1; echo "dp graph rc=$? $(tail -1 $out/pytest_dpgraph.log)"; grep -E "plain graph|^FAILED|^E  " "$out/pytest_dpgraph.log" | head -8 | cut -c1-400
for r in timit_lstm libri_gru; do
  ms=$(timeout 300 python bench.py --recipe $r --no-extras --no-cpu-baseline --steps 30 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "default $r $ms" | tee -a "$out/ab.txt"
done
ms=$(timeout 600 python bench.py --recipe libri_gru --prec fp32 --no-extras --no-cpu-baseline --steps 2 --warmup 1 --prewarm-s 0 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step); echo "libri_gru fp32 (step-wise, split products) $ms" | tee -a "$out/ab.txt"
ms=$(timeout 600 python bench.py --prec fp32 --no-extras --no-cpu-baseline --steps 3 --warmup 1 --prewarm-s 0 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step); echo "timit_ligru fp32 $ms" | tee -a "$out/ab.txt"
timeout 300 python tools/bench_gemm_lib.py "$out/r05_gemm_library_reference.json" > "$out/gemm_lib.log" 2>&1; echo "gemm lib rc=$?"; cat "$out/gemm_lib.log" | cut -c1-250
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$out/prof" -- python /root/repo/bench.py --steps 8 --warmup 2 --prewarm-s 0 --no-cpu-baseline --no-extras > "$out/prof_bench.log" 2>&1 )
db=$(find "$out/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then
    python tools/rocpd_stats.py "$db" "$out/kernel_stats.csv" > /dev/null 2> "$out/kernel_stats.err" || true
    python tools/rocpd_dump.py "$db" "$out/tail.csv" 1200 > /dev/null 2>&1 || true
    python tools/timeline_step.py "$out/tail.csv" 3 > "$out/timeline_step.txt" 2>/dev/null || true
    head -8 "$out/kernel_stats.csv" | cut -c1-140
    rm -rf "$out/prof"
fi
