#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel calls / total / average, like --stats.
usage: tools/rocprof_top.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                  "group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,percent"]
for n, c, s, a, mn, mx in rows:
    lines.append(f"\"{n}\",{c},{s / 1e6:.3f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * s / tot:.2f}")
text = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
print(text)
