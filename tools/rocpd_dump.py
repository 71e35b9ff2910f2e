#!/usr/bin/env python
"""Dump the tail of a rocprofv3 rocpd kernel trace as CSV (one row per dispatch, every column of the `kernels` view
that is a number or a short string) for timeline analysis off the box:  python tools/rocpd_dump.py db out.csv [last_n]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
rows = db.execute("select * from kernels order by start desc limit %d" % n).fetchall()[::-1]
keep = [i for i, c in enumerate(cols) if c not in ("extdata", "args", "guid")]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow([cols[i] for i in keep])
    for r in rows:
        w.writerow([(str(r[i])[:90] if isinstance(r[i], str) else r[i]) for i in keep])
print("columns:", cols)
