#!/usr/bin/env python3
"""tools/df_potrf_stamps.py [workload] -- where a diagonal tile's time goes inside k_df_chain: the 15 stamps of chol_device.h::potrf_body
(GTG_DF_TRACE=1), median over the tiles of one factorisation, in microseconds.
stamps: 0 start, 1 tile image in LDS, then per panel jb: 2+3jb = 3+3jb after the stage (pivots + followers + deferred work) and its barrier,
4+3jb after the next panel's updates (P3), the drain and the barrier; 14 = last column stored."""
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
    ok = st[:, 14] > 0
    st = st[ok]
    d = np.diff(st[:, :15], axis=1)
    med = np.median(d, axis=0)
    names = ["load tile image"] + sum([[f"panel {jb}: stage (pivots, followers, deferred) + barrier", f"panel {jb}: (flag)", f"panel {jb}: next panel's updates + drain + barrier"] for jb in range(4)], []) + ["store last column"]
    out = {"workload": w, "tiles": int(ok.sum()), "potrf_body_us_median": float(np.median(st[:, 14] - st[:, 0])),
           "in_to_out_us_median": float(np.median((chain[ok, 1] - chain[ok, 0]) / 100.0)),
           "stages_us_median": {n: round(float(x), 2) for n, x in zip(names, med)}}
    # per panel and wavefront: when the wavefront was done with its part of the stage, relative to the stage's start (us, median)
    starts = np.stack([st[:, 1], st[:, 4], st[:, 7], st[:, 10]], 1)
    wv = st[:, 16:48].reshape(-1, 4, 8)
    out["wavefront_done_after_stage_start_us_median"] = {f"panel {jb}": [round(float(np.median(wv[:, jb, w] - starts[:, jb])), 2) for w in range(8)] for jb in range(4)}
    fw = st[:, 48:60].reshape(-1, 3, 4) - st[:, 7][:, None, None]
    out["panel 2 followers (wave 1, 5, 7): a[] loaded, pivots replayed, rs seen, done (us after stage start, median)"] = [[round(float(np.median(fw[:, k, j])), 2) for j in range(4)] for k in range(3)]
    print(json.dumps(out, indent=1))
    dev.close()


if __name__ == "__main__":
    main()
