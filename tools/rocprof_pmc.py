#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (rocpd sqlite): per kernel and counter: dispatches, sum, mean.
usage: tools/rocprof_pmc.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                  "group by kernel_name, counter_name order by sum(value) desc").fetchall()
lines = ["kernel,counter,dispatches,sum,mean"]
for k, c, n, s, a in rows:
    lines.append(f"\"{k}\",{c},{n},{s:.1f},{a:.2f}")
text = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
print(text)
