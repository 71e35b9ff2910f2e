#!/usr/bin/env python
"""Which host-side op launches which kernel in one EAGER training step of a recipe (torch.profiler, GPU box):
    python tools/step_ops_profile.py timit_mlp [fp32] > gpurun_out/ops_timit_mlp.txt
Lists every device kernel of one step in launch order with the torch op and the python frames that issued it - the map from
the stock `at::native` / `rocclr` launches of a kernel trace back to the lines of functional.py / nn.py that cause them."""
import collections
import importlib
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")

recipe = sys.argv[1] if len(sys.argv) > 1 else "timit_mlp"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
sys.argv = ["bench.py", "--recipe", recipe, "--graph", "off", "--prec", prec]
args = bench.parse()
tr = bench.Trainer(args, 0, 1)
for i in range(4):
    tr.step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
    tr.step(0)
    torch.cuda.synchronize()
evs = prof.events()
# device kernels, each with the CPU op that launched it (correlation through the profiler's linked events)
rows = []
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        continue
    for k in getattr(e, "kernels", []) or []:
        stack = [s for s in (e.stack or []) if "pytorch-kaldi_amd" in s or "bench.py" in s][:3]
        rows.append((e.time_range.start, e.name, k.name, k.duration, " <- ".join(s.split("/")[-1] for s in stack)))
rows.sort()
seen = set()
n = 0
count = collections.Counter()
for t, op, kern, dur, stack in rows:
    key = (t, kern)
    if key in seen:
        continue
    seen.add(key)
    n += 1
    stock = kern.startswith("void at::") or "rocclr" in kern or kern.startswith("at::")
    short = kern.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    count[(op, short, stack)] += 1 if stock else 0
    print("%3d %-34s %-70s %6.1f us  %s" % (n, op[:34], short, dur, stack))
print("\n---- stock torch launches by (op, kernel, frames)")
for (op, kern, stack), c in count.most_common():
    if c:
        print("%3d x %-30s %-60s %s" % (c, op[:30], kern[:60], stack))
