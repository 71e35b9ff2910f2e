#!/bin/bash
# round 5, call C: helper touch delay sweep; fused conv-layer tail (kernel test, SincNet / CNN fixtures, A/B with bf16 convs);
# the launch-bound steps with their collectives inside the HIP graph.
set -u
out=$PWD/gpurun_out/r05c
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
ROUNDS=2 timeout 300 python tools/helper_sweep.py "$out/helper_ligru.json" > "$out/helper_ligru.log" 2>&1; echo "sweep liGRU rc=$?"; grep -E "fwd|Error|error" "$out/helper_ligru.log" | head -20
KIND=LSTM ROUNDS=2 CONFIGS="P+out lead3/1 x4 d8;P+out lead4/2 x4 d12;bwd lead4 x4 d0;bwd lead4 x4 d8;bwd lead6 x6 d8;all lead3/1/4 x4 d8" timeout 300 python tools/helper_sweep.py "$out/helper_lstm.json" > "$out/helper_lstm.log" 2>&1; echo "sweep LSTM rc=$?"; grep -E "fwd|Error|error" "$out/helper_lstm.log" | head
KIND=GRU ROUNDS=2 CONFIGS="P lead3 x4 d8;bwd lead4 x4 d8;all lead3/1/4 x4 d8" timeout 300 python tools/helper_sweep.py "$out/helper_gru.json" > "$out/helper_gru.log" 2>&1; echo "sweep GRU rc=$?"; grep -E "fwd|Error|error" "$out/helper_gru.log" | head
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_reference_pins.py tests/test_gpu_parity.py -q -m gpu -k "conv_layer_tail or sincnet or cnn or conv" > "$out/pytest_ln.log" 2>&1; echo "fused tail tests rc=$? $(tail -1 $out/pytest_ln.log)"; grep -E "^FAILED|^E  " "$out/pytest_ln.log" | head -8 | cut -c1-300
PK_CONV_BF16=1 timeout 600 python -m pytest tests/test_gpu_reference_pins.py tests/test_gpu_parity.py -q -m gpu -k "sincnet or cnn" > "$out/pytest_ln_bf.log" 2>&1; echo "same with PK_CONV_BF16=1 rc=$? $(tail -1 $out/pytest_ln_bf.log)"; grep -E "^FAILED|^E  " "$out/pytest_ln_bf.log" | head -8 | cut -c1-300
for i in 1 2; do for v in PK_CONV_BF16=0 PK_CONV_BF16=1; do
  ms=$(env $v timeout 200 python bench.py --recipe timit_sincnet --steps 100 --warmup 5 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step)
  echo "$v timit_sincnet $ms" | tee -a "$out/ab.txt"
done; done
timeout 900 python -m pytest tests/test_gpu_dp_two_ranks.py -q -m gpu -s -k "launch_bound" > "$out/pytest_dpgraph.log" 2>&1; echo "dp graph rc=$? $(tail -1 $out/pytest_dpgraph.log)"; grep -E "plain graph|^FAILED|^E  " "$out/pytest_dpgraph.log" | head -8 | cut -c1-400
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$out/prof" -- python /root/repo/bench.py --recipe timit_sincnet --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$out/prof.log" 2>&1 )
db=$(find "$out/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" "$out/r05_timit_sincnet_kernel_stats.csv" > /dev/null 2> "$out/kstats.err"; head -30 "$out/r05_timit_sincnet_kernel_stats.csv" | cut -c1-150; rm -rf "$out/prof"; fi
