#!/usr/bin/env python3
"""Timeline of the Cholesky panel chain from a rocprofv3 (rocpd sqlite) kernel trace: for the LAST factorisation
in the trace print start offset / duration / gap of every kernel between two block-column steps.
usage: tools/rocprof_timeline.py <results.db> [first_potrf_index] [count]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
start = "start" if "start" in cols else "start_timestamp"
end = "end" if "end" in cols else "end_timestamp"
extra = [c for c in ("stream_id", "queue_id", "grid_x", "grid_size_x", "workgroup_x") if c in cols]
q = f"select name, {start}, {end}" + "".join(", " + c for c in extra) + f" from kernels order by {start}"
rows = db.execute(q).fetchall()
pot = [i for i, r in enumerate(rows) if "k_potrf128" in r[0] or "k_panel128" in r[0]]
# last factorisation = last run of potrf launches; a factorisation has nt of them, find nt from gaps > 1 ms
runs, cur = [], [pot[0]]
for a, b in zip(pot, pot[1:]):
    if rows[b][1] - rows[a][1] > 1.5e6:
        runs.append(cur); cur = []
    cur.append(b)
runs.append(cur)
run = runs[-1]
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(run) // 2
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lo, hi = run[i0], run[min(i0 + cnt, len(run) - 1)]
t0 = rows[lo][1]
print(f"factorisation with {len(run)} block columns, total {(rows[run[-1]][2] - rows[run[0]][1]) / 1e6:.3f} ms; steps {i0}..{i0 + cnt}")
print("start_us  dur_us  name  " + " ".join(extra))
for r in rows[lo:hi + 1]:
    nm = r[0].split("(")[0][-40:]
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f}  {nm}  " + " ".join(str(x) for x in r[3:]))
starts = [rows[i][1] for i in run]
import statistics
d = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
print("potrf start-to-start us: median %.1f mean %.1f min %.1f max %.1f" % (statistics.median(d), sum(d) / len(d), min(d), max(d)))
