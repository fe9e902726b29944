#!/usr/bin/env python3
"""tools/df_check.py -- GPU check of the dataflow Cholesky (csrc/chol_dataflow.hip): dense matrices of a few sizes against
numpy, then the Ladybug-1723 shape: one damped solve with both schedules (same delta), phase times of repeated tries."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_amd import lib as L  # noqa: E402
from gtsam_amd.problem import Problem  # noqa: E402


def dense(n, seed):
    rng = np.random.default_rng(seed)
    M = rng.normal(size=(n, n)) * 0.1
    A = M @ M.T + np.eye(n) * (1.0 + rng.uniform(0, 1, n))
    g = rng.normal(size=n)
    dev = L.DeviceGraph(Problem(var_type=np.array([0], np.int32)))
    t0 = time.perf_counter()
    rc, Lf, x = dev.dense_cholesky(A, g)
    dt = time.perf_counter() - t0
    Ld = np.linalg.cholesky(A)
    out = {"n": n, "rc": rc, "L_err": float(np.abs(np.tril(Lf) - Ld).max() / np.abs(Ld).max()),
           "x_err": float(np.abs(x - np.linalg.solve(A, g)).max() / np.abs(x).max()), "sec": dt}
    dev.close()
    return out


def main():
    for n in (() if "--df-only" in sys.argv else (100, 128, 200, 300, 640, 1500, 2600)):
        print(json.dumps(dense(n, n)), flush=True)
    if "--dense-only" in sys.argv:
        return
    from gtsam_amd import datasets as D
    from gtsam_amd.problem import bal_problem
    p, v0 = bal_problem(*D.ladybug_1723())
    res = {}
    scheds = ("streams", "df") if "--df-only" not in sys.argv else ("df",)
    for sched in scheds:
        os.environ["GTG_CHOL"] = sched
        dev = L.DeviceGraph(p)
        dev.set_values(v0)
        dev.linearize()
        try:
            rc, out = dev.try_lambda(1e-4, True)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"first try": str(e)[:60], "ctrl": dev.df_ctrl().tolist()}), flush=True)
            raise
        d = dev.delta()
        dev.enable_timing(True); dev.reset_timing()
        t0 = time.perf_counter()
        for it in range(10):
            try:
                dev.try_lambda(1e-4, True)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"try": it, "error": str(e)[:80], "ctrl": dev.df_ctrl().tolist()}), flush=True)
                break
        wall = (time.perf_counter() - t0) / 10
        ph = dev.phase_ms()
        res[sched] = d
        print(json.dumps({"schedule": sched, "rc": rc, "out": [float(x) for x in out], "try_ms_wall": 1e3 * wall,
                          "cholesky_ms": ph["cholesky"][0] / max(ph["cholesky"][1], 1), "flops": dev.cholesky_flops(),
                          "solve_ms": ph["solve"][0] / max(ph["solve"][1], 1)}), flush=True)
        dev.close()
    if "streams" in res:
        print(json.dumps({"delta_diff_rel": float(np.abs(res["df"] - res["streams"]).max() / np.abs(res["streams"]).max())}))


if __name__ == "__main__":
    main()
