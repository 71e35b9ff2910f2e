#!/bin/bash
# The driver's exact command in fresh processes, with the per-step trace:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_driver_cmd.sh r04a'
# Legs: the driver's flags without / with the untimed pre-warm, the builder's 100/3, the launch-bound recipes; a
# background sampler records sclk / power while the first leg runs.
set -u
tag=${1:-r04a}
out=gpurun_out/$tag
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
(nproc; lscpu | grep -E "Model name|MHz" ; rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -vE "^=|^$" ) > "$out/box.txt" 2>&1
( for i in $(seq 1 150); do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr '\n' ' ')"; sleep 0.2; done ) > "$out/smi_trace.txt" 2>&1 &
SMI=$!
D="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --step-trace"
for rep in 1 2; do
  $D --prewarm-s 0 > "$out/drv_pre0_$rep.json" 2> "$out/drv_pre0_$rep.err"; echo "drv pre0 #$rep: $(python3 tools/jget.py "$out/drv_pre0_$rep.json" ms_per_step step_ms.first step_ms.median step_ms.host_enqueue_median)"
done
kill $SMI 2>/dev/null
for rep in 1 2; do
  $D > "$out/drv_pre_$rep.json" 2> "$out/drv_pre_$rep.err"; echo "drv prewarm #$rep: $(python3 tools/jget.py "$out/drv_pre_$rep.json" ms_per_step step_ms.first step_ms.median step_ms.host_enqueue_median config.prewarm_steps)"
done
python3 bench.py --gpus 1 --steps 100 --warmup 3 --no-extras --no-cpu-baseline --prewarm-s 0 > "$out/b100_pre0.json" 2> "$out/b100_pre0.err"
echo "100/3 pre0: $(python3 tools/jget.py "$out/b100_pre0.json" ms_per_step step_ms.first step_ms.median step_ms.host_enqueue_median)"
for r in timit_mlp timit_sincnet; do
  st=400; [ $r = timit_sincnet ] && st=100
  for pw in 0 -1; do
    python3 bench.py --recipe $r --steps $st --warmup 5 --repeats 3 --no-extras --no-cpu-baseline --prewarm-s $pw > "$out/${r}_pw$pw.json" 2> "$out/${r}_pw$pw.err"
    echo "$r prewarm=$pw: $(python3 tools/jget.py "$out/${r}_pw$pw.json" ms_per_step regions_ms_per_step step_ms.median step_ms.host_enqueue_median)"
  done
done
