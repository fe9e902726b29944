#!/usr/bin/env python
"""Times one damped solve of a bench workload with the Cholesky path and with block-Jacobi PCG on the implicit Schur
complement (gtg_try_lambda_pcg), for a few CG tolerances.  Prints one JSON line per case.

    python tools/pcg_probe.py [--workload ladybug1723] [--lambda 1e-4]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ladybug1723")
    ap.add_argument("--lam", type=float, default=1e-4)
    args = ap.parse_args()
    import torch
    from bench import build_workload
    from gtsam_amd import lib as L
    assert torch.cuda.is_available()
    (problem, values0), desc = build_workload(args.workload)
    dev = L.DeviceGraph(problem)
    dev.set_values(values0)
    dev.linearize()
    diag = bool(problem.n_sfm)      # Ceres-style diagonal damping for bundle adjustment, Levenberg for pose graphs

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) / reps * 1e3

    (rc, out), ms = timed(lambda: dev.try_lambda(args.lam, diag))
    d_direct = dev.delta()
    print(json.dumps({"workload": args.workload, "solver": "cholesky", "status": rc, "ms": round(ms, 3), "lin_decrease": out[0] - out[1]}))
    for er, ea in ((1e-1, 1e-3), (1e-2, 1e-3), (1e-3, 1e-3), (1e-6, 1e-12), (1e-10, 1e-20)):
        (rc, out, its), ms = timed(lambda: dev.try_lambda_pcg(args.lam, diag, max_iterations=2000, epsilon_rel=er, epsilon_abs=ea), reps=2)
        d = dev.delta()
        print(json.dumps({"workload": args.workload, "solver": "pcg", "epsilon_rel": er, "status": rc, "iterations": its, "ms": round(ms, 3),
                          "ms_per_iteration": round(ms / max(its, 1), 4), "lin_decrease": out[0] - out[1],
                          "delta_vs_cholesky": float(np.abs(d - d_direct).max() / np.abs(d_direct).max())}))
    dev.close()


if __name__ == "__main__":
    main()
