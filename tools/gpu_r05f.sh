#!/bin/bash
# round 5, call F: the sinc bank as one launch each way (kernel test, SincNet fixtures in both precisions), SincNet / MLP steps
set -u
out=$PWD/gpurun_out/r05f
mkdir -p "$out"
if ! timeout 120 python -c "import torch; x = torch.zeros(1 << 20).cuda() + 1; torch.cuda.synchronize(); print('gpu ok', float(x.sum()))"; then
    echo "BAD BOX: first GPU touch failed"; exit 0
fi
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_reference_pins.py tests/test_gpu_parity.py -q -m gpu -k "sinc or cnn or conv" > "$out/pytest_sinc.log" 2>&1; echo "sinc tests rc=$? $(tail -1 $out/pytest_sinc.log)"; grep -E "^FAILED|^E  " "$out/pytest_sinc.log" | head -8 | cut -c1-300
for i in 1 2; do
  ms=$(timeout 200 python bench.py --recipe timit_sincnet --steps 100 --warmup 5 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step)
  echo "timit_sincnet $ms" | tee -a "$out/ab.txt"
  ms=$(timeout 200 python bench.py --recipe timit_mlp --steps 400 --warmup 5 --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 tools/jget.py /dev/stdin ms_per_step)
  echo "timit_mlp $ms" | tee -a "$out/ab.txt"
done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$out/prof" -- python /root/repo/bench.py --recipe timit_sincnet --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$out/prof.log" 2>&1 )
db=$(find "$out/prof" -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" "$out/r05_timit_sincnet_kernel_stats.csv" > /dev/null 2> "$out/kstats.err"; python - "$out/r05_timit_sincnet_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = max(int(r["Calls"]) for r in rows if "fused_step_kernel<1>" in r["Name"] or "fused_step_kernel" in r["Name"]) // 3 or 1
stock = sum(int(r["Calls"]) for r in rows if r["Name"].startswith("at::") or "rocclr" in r["Name"] or r["Name"].startswith("void at::") or "Cijk" in r["Name"])
total = sum(int(r["Calls"]) for r in rows)
print("launches per step ~%.0f, stock ~%.0f (steps ~%d)" % (total / steps, stock / steps, steps))
PY
rm -rf "$out/prof"; fi
