#!/bin/bash
out=gpurun_out/r03e
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -x -k "bf16_persistent or full_geometry or dirty or share_their_input" > $out/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $out/pytest.log)"
bash tools/gpu_ab3.sh 3 "PK_REC_GEN_BWD=4" "PK_REC_GEN_BWD=2" "PK_REC_GEN_BWD=4 PK_HEAD_DX_SHARE=0"
python bench.py --mask-rng reference --steps 20 --no-extras --no-cpu-baseline > $out/bench_maskref.json 2> $out/bench_maskref.err
echo "mask-rng reference: $(python -c "import json;d=json.loads(open('$out/bench_maskref.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
PK_REC_GEN_BWD=4 JSON_OUT=$out/trace_bwd4.json timeout 120 python tools/trace_rec2.py > $out/trace_bwd4.log 2>&1; grep -A8 "^bwd" $out/trace_bwd4.log
