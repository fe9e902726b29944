#!/usr/bin/env python3
"""tools/df_hop_timeline.py [workload] -- the serial chain of the dataflow Cholesky, tile to tile (GTG_DF_TRACE=1): for every diagonal tile J
whose left neighbour (J, J-1) is stored, the events between the start of potrf(J-1) and the start of potrf(J), medians in microseconds
after potrf(J-1)'s first stamp:
  panel q of (J-1, J-1) released (the stamp after the flag store of chol_device.h::potrf_body),
  the substitution task of tile (J, J-1) sees panel q (chol_dataflow.hip::substitute, tr[4 + q]), its contraction done, the task done,
  the chain workgroup of tile J: PD(J) seen (trace[2 J]), the slices 2 and 3 of tile (J, J-1) seen / in LDS (stamps 60 - 63), potrf_body(J)
  starts (stamp 0), its image complete (stamp 1)."""
import json
import os
import sys

import numpy as np

os.environ["GTG_DF_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gtsam_amd import lib as L  # noqa: E402
from gtsam_amd.params import LevenbergMarquardtParams  # noqa: E402


def main():
    w = sys.argv[1] if len(sys.argv) > 1 else "ladybug1723"
    (p, v0), _ = bench.build_workload(w)
    prm = LevenbergMarquardtParams.CeresDefaults() if p.n_sfm else LevenbergMarquardtParams()
    dev = L.DeviceGraph(p)
    dev.set_values(v0); dev.linearize()
    for _ in range(3):
        dev.try_lambda(prm.lambdaInitial, prm.diagonalDamping, prm.minDiagonal, prm.maxDiagonal)
    tasks, chain = dev.df_trace()
    st = dev.potrf_stamps.astype(np.float64) / 100.0
    tk = tasks.astype(np.float64) / 100.0
    ch = chain.astype(np.float64) / 100.0
    pl = dev.df_plan()
    T = pl["tasks"]
    I, J, piece, pieces = T[:, 0], T[:, 1], T[:, 4], T[:, 5]
    sub = {int(J[i]): i for i in range(len(T)) if I[i] == J[i] + 1 and piece[i] == pieces[i] - 1}   # the task that substitutes tile (J+1, J)
    rows = []
    for j in range(1, pl["nt"]):
        if j - 1 not in sub or st[j, 14] <= 0 or st[j - 1, 14] <= 0:
            continue
        s = sub[j - 1]
        t0 = st[j - 1, 0]
        rel = [st[j - 1, 4] - t0, st[j - 1, 7] - t0, st[j - 1, 10] - t0, st[j - 1, 11] - t0, st[j - 1, 14] - t0,      # panels 0-2 (late flags), panel 3, last column stored
               tk[s, 0] - t0, tk[s, 1] - t0, tk[s, 4] - t0, tk[s, 5] - t0, tk[s, 6] - t0, tk[s, 7] - t0, tk[s, 2] - t0,   # sub task: taken, contraction done, sees panel 0..3, done
               ch[j, 0] - t0, st[j, 60] - t0, st[j, 61] - t0, st[j, 62] - t0, st[j, 63] - t0, st[j, 0] - t0, st[j, 1] - t0, st[j, 0] - st[j - 1, 0]]
        rows.append(rel)
    rows = np.array(rows)
    names = ["panel 0 released", "panel 1 released", "panel 2 released", "panel 3 released", "potrf(J-1) last column stored",
             "sub task taken", "sub contraction done", "sub sees panel 0", "sub sees panel 1", "sub sees panel 2", "sub sees panel 3", "sub task done (tile (J,J-1) final)",
             "chain(J): PD(J) seen", "chain(J): slice 2 seen", "chain(J): slice 2 in LDS", "chain(J): slice 3 seen", "chain(J): slice 3 in LDS", "potrf_body(J) starts", "potrf_body(J): barrier after image", "period"]
    out = {"workload": w, "tiles": len(rows), "median_us_after_potrf_start": {n: round(float(np.median(rows[:, k])), 2) for k, n in enumerate(names)},
           "p10": {n: round(float(np.percentile(rows[:, k], 10)), 2) for k, n in enumerate(names)},
           "p90": {n: round(float(np.percentile(rows[:, k], 90)), 2) for k, n in enumerate(names)},
           "sum_periods_us": round(float(rows[:, -1].sum()), 1)}
    print(json.dumps(out, indent=1))
    dev.close()


if __name__ == "__main__":
    main()
