#!/bin/bash
# round 5, call A: L2 run-ahead helpers (sweep + headline A/B + parity with helpers on), bf16 convolutions graded on whole
# tensors, GEMM library reference with the product's call, the full-shape parity test.
set -u
out=$PWD/gpurun_out/r05a
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
ROUNDS=2 timeout 300 python tools/helper_sweep.py "$out/helper_ligru.json" > "$out/helper_ligru.log" 2>&1; echo "sweep liGRU rc=$?"; grep -E "fwd|Error|error" "$out/helper_ligru.log" | head -20
KIND=LSTM ROUNDS=2 CONFIGS="P lead3 x4;P+out lead3/1 x4;all lead3/1/2 x4" timeout 300 python tools/helper_sweep.py "$out/helper_lstm.json" > "$out/helper_lstm.log" 2>&1; echo "sweep LSTM rc=$?"; grep -E "fwd|Error|error" "$out/helper_lstm.log" | head
KIND=GRU ROUNDS=2 CONFIGS="P lead3 x4;P+out lead3/1 x4" timeout 300 python tools/helper_sweep.py "$out/helper_gru.json" > "$out/helper_gru.log" 2>&1; echo "sweep GRU rc=$?"; grep -E "fwd|Error|error" "$out/helper_gru.log" | head
for i in 1 2; do for v in PK_REC_HELPER=0 PK_REC_HELPER=1 PK_REC_HELPER=3; do
  ms=$(env $v timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python tools/jget.py /dev/stdin ms_per_step)
  echo "$v headline $ms" | tee -a "$out/ab.txt"
done; done
PK_REC_HELPER=7 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "persistent or full_geometry or dirty" > "$out/pytest_helper.log" 2>&1; echo "parity with helpers rc=$? $(tail -1 $out/pytest_helper.log)"
PK_CONV_BF16=1 timeout 900 python -m pytest tests/test_gpu_reference_pins.py tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -s -k "sincnet or cnn or conv" > "$out/pytest_conv1.log" 2>&1; echo "conv bf16=1 rc=$? $(tail -1 $out/pytest_conv1.log)"; grep -E "^FAILED|^E  " "$out/pytest_conv1.log" | head -8 | cut -c1-300
timeout 300 python tools/bench_gemm_lib.py "$out/r05_gemm_library_reference.json" > "$out/gemm_lib.log" 2>&1; echo "gemm lib rc=$?"; cat "$out/gemm_lib.log" | cut -c1-250
PK_FULL_SHAPE_JSON=$out/r05_full_shape_parity.json timeout 900 python -m pytest tests/test_gpu_full_shape.py -q -m gpu -s > "$out/pytest_full.log" 2>&1; echo "full shape rc=$? $(tail -1 $out/pytest_full.log)"; grep "full shape:" "$out/pytest_full.log" | cut -c1-400
