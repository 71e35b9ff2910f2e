#!/bin/bash
out=gpurun_out/r03d
mkdir -p $out
PK_REC_GEN_FWD=3 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16_persistent or full_geometry" > $out/pytest_fwd3.log 2>&1
echo "pytest fwd3 rc=$? $(tail -1 $out/pytest_fwd3.log)"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16_persistent or full_geometry or dirty" > $out/pytest_default.log 2>&1
echo "pytest default rc=$? $(tail -1 $out/pytest_default.log)"
bash tools/gpu_ab3.sh 2 "PK_REC_GEN_FWD=2 PK_REC_GEN_BWD=4" "PK_REC_GEN_FWD=3 PK_REC_GEN_BWD=4" "PK_REC_GEN_FWD=4 PK_REC_GEN_BWD=4" "PK_REC_GEN_FWD=2 PK_REC_GEN_BWD=3" "PK_REC_GEN_FWD=2 PK_REC_GEN_BWD=2"
python bench.py --mask-rng reference --steps 20 --no-extras --no-cpu-baseline > $out/bench_maskref.json 2> $out/bench_maskref.err
echo "mask-rng reference: $(python -c "import json;d=json.loads(open('$out/bench_maskref.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
