#!/usr/bin/env python
"""Where do the sporadic multi-second steps of a long LSTM / GRU run come from?  RECIPES (comma list, one process, built
and released one after the other like bench.py's other_configs), STEPS per recipe; SYNC=1 times every step with a device
sync (allocator numbers beside it), SYNC=0 lets the host run ahead like bench.py does and times the steps with one HIP
event each.  PK_REC_HELPER / PK_EXPERIMENT select the arm."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N = int(os.environ.get("STEPS", "150"))
SYNC = os.environ.get("SYNC", "1") == "1"
print("arm helper=%s exp=%s sync=%s" % (os.environ.get("PK_REC_HELPER"), os.environ.get("PK_EXPERIMENT"), SYNC))
for recipe in os.environ.get("RECIPES", "timit_lstm").split(","):
    sys.argv = [sys.argv[0], "--recipe", recipe, "--no-extras", "--no-cpu-baseline"]
    args = bench.parse()
    tr = bench.Trainer(args, 0, 1)
    for i in range(3):
        tr.step(i)
    torch.cuda.synchronize()
    ms, mem, host = [], [], []
    if SYNC:
        for i in range(N):
            t0 = time.perf_counter()
            tr.step(i)
            torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
            if i % 20 == 0 or ms[-1] > 100:
                mem.append((i, round(ms[-1], 1), round(torch.cuda.memory_allocated() / 2**30, 2), round(torch.cuda.memory_reserved() / 2**30, 2)))
    else:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
        evs[0].record()
        for i in range(N):
            t0 = time.perf_counter()
            tr.step(i)
            evs[i + 1].record()
            host.append(1e3 * (time.perf_counter() - t0))
        torch.cuda.synchronize()
        ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
    s = sorted(ms)
    slow = [(i, round(v, 1)) + ((round(host[i], 1),) if host else ()) for i, v in enumerate(ms) if v > 2 * s[len(s) // 2]]
    print("%s: median %.2f max %.1f mean %.2f slow(>2x median: step, ms[, host enqueue ms]) %d %s" % (
        recipe, s[len(s) // 2], s[-1], sum(ms) / len(ms), len(slow), slow[:10]), flush=True)
    if mem:
        print("  (step, ms, allocated GB, reserved GB):", mem)
    st = torch.cuda.memory_stats()
    print("  alloc_retries", st.get("num_alloc_retries"), "device mallocs", st.get("num_device_alloc"), "device frees", st.get("num_device_free"), flush=True)
    bench.release(tr)
    del tr
