#!/bin/bash
# round 3, call c: third generation with coalesced HBM accesses (transposer patches) vs direct vs second generation
out=gpurun_out/r03c
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -x -k "persistent or full_geometry or dirty or oracle_parity or permlane or golden_module" > $out/pytest_gen3.log 2>&1
echo "pytest gen3 rc=$? $(tail -1 $out/pytest_gen3.log)"
PK_REC_FLUSH_LATE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16_persistent or full_geometry or dirty" > $out/pytest_gen3_late.log 2>&1
echo "pytest gen3 late rc=$? $(tail -1 $out/pytest_gen3_late.log)"
PK_REC_GEN=4 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16_persistent or full_geometry" > $out/pytest_gen4.log 2>&1
echo "pytest gen4 rc=$? $(tail -1 $out/pytest_gen4.log)"
for v in "PK_REC_GEN=2" "PK_REC_GEN=3" "PK_REC_GEN=3 PK_REC_FLUSH_LATE=1" "PK_REC_GEN=4"; do
  tag=$(echo $v | tr ' =' '__')
  env $v JSON_OUT=$out/trace_$tag.json timeout 120 python tools/trace_rec2.py > $out/trace_$tag.log 2>&1
  echo "== $v"; grep -E "cycles/step|mean  " $out/trace_$tag.log | head -16
done
bash tools/gpu_ab3.sh 2 PK_REC_GEN=2 PK_REC_GEN=3 "PK_REC_GEN=3 PK_REC_FLUSH_LATE=1" PK_REC_GEN=4
