#!/usr/bin/env python
"""Time-ordered kernel list of the LAST step in a rocprofv3 rocpd database: start offset, duration, gap to the
previous kernel's end, name.  `python tools/rocpd_timeline.py db n_kernels_back`  (steady state: the trace's tail)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows = db.execute("select name, start, end from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
prev_end = t0
for name, s, e in rows:
    print("%9.1f us  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(name)))
    prev_end = max(prev_end, e)
