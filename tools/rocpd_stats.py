#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel-trace) as a per-kernel stats table:
calls, total / average / min / max duration.  `python tools/rocpd_stats.py db [out.csv]`"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end from kernels").fetchall()
    acc = {}
    for name, s, e in rows:
        d = acc.setdefault(name, [0, 0, 1 << 62, 0])
        dt = e - s
        d[0] += 1
        d[1] += dt
        d[2] = min(d[2], dt)
        d[3] = max(d[3], dt)
    total = sum(v[1] for v in acc.values()) or 1
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
    for name, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        out.append((short(name), v[0], v[1], round(v[1] / v[0], 1), round(100.0 * v[1] / total, 2), v[2], v[3]))
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == "__main__":
    main()
