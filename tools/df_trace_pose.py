#!/usr/bin/env python3
"""tools/df_trace_pose.py [w20000|sphere2500] -- timeline of one dataflow factorisation of a pose graph (several chains): per chain
workgroup when its tiles came in / went out, how long it waited in front of each, and what the bulk tasks did meanwhile.  GTG_DF_TRACE=1."""
import json
import os
import sys

import numpy as np

os.environ["GTG_DF_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gtsam_amd import lib as L  # noqa: E402
from tests import problems as PB  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "w20000"
    if name == "w20000":
        p, v0 = PB.pose2_graph(dict(np.load(os.path.join(ROOT, "tests", "golden", "pose2_w20000.npz"))))
    else:
        p, v0 = PB.sphere2500(dict(np.load(os.path.join(ROOT, "tests", "golden", "sphere2500.npz"))))
    dev = L.DeviceGraph(p)
    dev.set_values(v0); dev.linearize()
    for _ in range(3):
        dev.try_lambda(1e-4, False)
    tasks, chain = dev.df_trace()
    pl = dev.df_plan(); T = pl["tasks"]; nt = pl["nt"]
    ch = pl
    t0 = min(tasks[:, 0].min(), chain[chain[:, 0] > 0, 0].min())
    us = lambda x: (x - t0) / 100.0
    start, acc, done = us(tasks[:, 0]), us(tasks[:, 1]), us(tasks[:, 2])
    cin, cout = us(chain[:, 0]), us(chain[:, 1])
    total = max(done.max(), cout.max())
    I, J, kc, r, R = T[:, 0], T[:, 1], T[:, 3], T[:, 4], T[:, 5]
    off, tiles = ch["chain_off"], ch["chain_tiles"]
    out = {"workload": name, "nt": int(nt), "n_tasks": int(len(T)), "total_us": float(total), "n_chain_workgroups": int(len(off) - 1),
           "ksteps": int(kc.sum()), "pieces_max_per_tile": int(R.max()), "task_resident_us_total": float((done - start).sum()),
           "workgroups_seen": int(len(set((tasks[:, 3] & ((1 << 40) - 1)).tolist())))}
    rows = []
    for w in range(len(off) - 1):
        mine = tiles[off[w]:off[w + 1]]
        if len(mine) == 0:
            continue
        busy = float((cout[mine] - cin[mine]).sum())          # PD seen -> factored (includes waiting for the slices of (J, J-1))
        first, last = float(cin[mine].min()), float(cout[mine].max())
        gaps = float(sum(max(0.0, cin[mine[k + 1]] - cout[mine[k]]) for k in range(len(mine) - 1)))   # waiting for the next tile's PD
        rows.append(dict(wg=w, tiles=int(len(mine)), first_in=round(first, 1), last_out=round(last, 1), in_to_out_sum=round(busy, 1), waiting_for_pd_sum=round(gaps, 1)))
    out["chain_workgroups"] = rows
    # the last diagonal tiles: who finished when
    order = np.argsort(cout)
    out["last_tiles_out"] = [(int(j), round(float(cin[j]), 1), round(float(cout[j]), 1)) for j in order[-8:]]
    pdm = np.where((I == J) & (r == R - 1))[0]
    late = sorted([(float(done[t] - start[t]), int(J[t]), int(R[t]), int(kc[t])) for t in pdm])[-6:]
    out["slowest_pd_final_pieces(us, J, pieces, steps)"] = late
    fin = np.where((I != J) & (r == R - 1))[0]
    out["final_piece_finalize_us_p50_p90"] = [float(x) for x in np.percentile(done[fin] - acc[fin], [50, 90])]
    out["early_piece_us_per_step_p50"] = float(np.median((acc - start)[(r < R - 1) & (kc > 0)] / kc[(r < R - 1) & (kc > 0)])) if ((r < R - 1) & (kc > 0)).any() else None
    print(json.dumps(out))
    dev.close()


if __name__ == "__main__":
    main()
