cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace -d /tmp/prof -o tl -- python $GRAFT_REPO_ROOT/bench.py --workload ${WL:-sphere2500} --steps 2 --warmup 1 --cpu-baseline off --skip-dense-roofline > /tmp/b.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python - <<PY
import sqlite3
db=sqlite3.connect("$DB")
cols=[r[1] for r in db.execute("pragma table_info(kernels)")]
rows=db.execute("select name,start,end,stream_id,grid_x from kernels order by start").fetchall()
pan=[i for i,r in enumerate(rows) if "k_panel128" in r[0]]
# last factorisation: find last gap > 1ms between panels
runs=[[pan[0]]]
for a,b in zip(pan,pan[1:]):
    if rows[b][1]-rows[a][1] > 1.5e6: runs.append([])
    runs[-1].append(b)
run=runs[-1]
lo,hi=run[0],run[-1]
t0=rows[lo][1]
print("factorisation", len(run), "panels, total ms", (rows[hi][2]-t0)/1e6)
import collections
for r in rows[lo:lo+70]:
    print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:7.1f} s{r[3]} {r[0].split('(')[0][-28:]} g{r[4]}")
PY
