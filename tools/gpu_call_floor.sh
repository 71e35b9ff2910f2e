#!/bin/bash
out=gpurun_out/floor2
mkdir -p $out
JSON_OUT=$out/trace_full.json timeout 120 python tools/trace_rec2.py > $out/trace_full.log 2>&1; grep -E "cycles/step|launch ms" $out/trace_full.log | head
EMPTY=1 JSON_OUT=$out/trace_empty.json timeout 120 python tools/trace_rec2.py > $out/trace_empty.log 2>&1; grep -E "cycles/step|launch ms" $out/trace_empty.log | head
