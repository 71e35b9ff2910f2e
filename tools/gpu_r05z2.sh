#!/bin/bash
# round 5, last call: the chunk loop with the step fence through its own GPU tests, smoke
set -u
out=$PWD/gpurun_out/r05z2; mkdir -p "$out"
timeout 70 python -m pytest tests/test_core_chunk.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2 | tee "$out/pytest_chunk.txt"
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee "$out/smoke.txt"
