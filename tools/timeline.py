#!/usr/bin/env python3
"""Timeline of the frame kernels of a rocprofv3 --kernel-trace database: which dispatches ran side by side.
usage: tools/timeline.py <results.db> [first] [count]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, queue_id, start, end, grid_x from kernels order by start"))
keep = [r for r in rows if any(k in r[0] for k in ("k_primary", "k_tail", "fillBuffer"))]
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(keep) - 60)
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = keep[first][2]
for name, q, s, e, g in keep[first:first + count]:
    short = "primary" if "k_primary" in name else ("tail" if "k_tail" in name else "fill")
    print(f"q{q:<3} {short:8s} start {(s - t0) / 1e3:9.1f} us  end {(e - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  grid {g}")
