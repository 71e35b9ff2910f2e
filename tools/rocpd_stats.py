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


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] == "--gaps"):
    main()


def gaps(db_path, top=25, tail_frac=1.0):
    """Idle time on the device between consecutive kernels, grouped by (previous, next) kernel names.
    tail_frac < 1 restricts the analysis to the last part of the trace (steady state, no warm-up)."""
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    if tail_frac < 1.0:
        t0 = rows[0][1] + (rows[-1][2] - rows[0][1]) * (1.0 - tail_frac)
        rows = [r for r in rows if r[1] >= t0]
    acc = {}
    busy = 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
        busy += e0 - s0
        g = s1 - e0
        if g <= 0:
            continue
        k = (short(n0)[:50], short(n1)[:50])
        d = acc.setdefault(k, [0, 0])
        d[0] += g
        d[1] += 1
    span = rows[-1][2] - rows[0][1]
    print("span %.3f ms, busy %.3f ms, idle %.3f ms" % (span / 1e6, busy / 1e6, (span - busy) / 1e6))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
        print("%9.3f ms %6d x  %s  ->  %s" % (v[0] / 1e6, v[1], k[0], k[1]))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "--gaps":
    gaps(sys.argv[1], tail_frac=float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
