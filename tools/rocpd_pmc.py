#!/usr/bin/env python
"""Per-kernel mean of one PMC counter from a rocprofv3 rocpd database (counters_collection view).
python tools/rocpd_pmc.py db [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 100 else name[:97] + "..."


db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
acc = {}
for k, c, v, d in rows:
    a = acc.setdefault((k, c), [0, 0.0, 0])
    a[0] += 1
    a[1] += v
    a[2] += d
out = [("Kernel", "Counter", "Dispatches", "MeanValue", "SumValue", "MeanDurationNs")]
for (k, c), a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    out.append((short(k), c, a[0], round(a[1] / a[0], 3), round(a[1], 3), round(a[2] / a[0], 1)))
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerows(out)
