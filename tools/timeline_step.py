#!/usr/bin/env python
"""One training step of a rocprofv3 kernel trace, in time order (start offset, duration, gap to the end of everything before
it - negative = it overlaps an earlier kernel - and the kernel's name), from the CSV tools/rocpd_dump.py wrote:
    python tools/timeline_step.py gpurun_out/<tag>/tail.csv [steps_back=3] > profiles/<round>_timeline_step.txt
A step = the dispatches between the last optimizer launch of one step and the last optimizer launch of the next."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["start"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:70]


opt = [i for i, r in enumerate(rows) if "fused_step" in r["name"] or "rmsprop_kernel" in r["name"]]
ends = [i for j, i in enumerate(opt) if j + 1 == len(opt) or opt[j + 1] != i + 1]  # last launch of each run of optimizer kernels
a, b = ends[-back - 1], ends[-back]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["start"]), int(step[-1]["end"])
print("# step of %d dispatches, %.3f ms from the first start to the last end (under the tracer)" % (len(step), (t1 - t0) / 1e6))
c, d = collections.Counter(), collections.Counter()
for r in step:
    c[short(r["name"])] += 1
    d[short(r["name"])] += int(r["end"]) - int(r["start"])
print("# by kernel: launches, summed duration")
for n, k in sorted(c.items(), key=lambda kv: -d[kv[0]]):
    print("#  %3d %9.1f us  %s" % (k, d[n] / 1e3, n))
print("#\n# start_us   dur_us   gap_us  kernel")
prev = t0
for r in step:
    s, e = int(r["start"]), int(r["end"])
    print("%9.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, short(r["name"])))
    prev = max(prev, e)
