#!/bin/bash
mkdir -p gpurun_out/bnb
{
PK_BNB_CW=16 python tools/bench_bnb.py
PK_BNB_CW=48 python tools/bench_bnb.py
PK_BNB_CW=48 PK_BNB_RBR=128 PK_BNB_RBA=128 python tools/bench_bnb.py
PK_BNB_CW=48 PK_BNB_RBR=384 PK_BNB_RBA=512 python tools/bench_bnb.py
PK_BNB_CW=16 G=4 python tools/bench_bnb.py
PK_BNB_CW=48 G=4 python tools/bench_bnb.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bnb/out3.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "bf16" 2>&1 | tail -3
