#!/bin/bash
set -u
out=gpurun_out/r02f
mkdir -p $out
timeout 300 python tools/dp_diag.py > $out/dp_diag.log 2>&1; tail -8 $out/dp_diag.log
