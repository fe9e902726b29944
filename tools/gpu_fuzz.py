#!/usr/bin/env python3
"""Fuzz parity of the HIP path against the oracle on random graphs (run on an MI355X; not part of the test suite).

    python tools/gpu_fuzz.py [--seeds 16]

The CPU counterpart (tests/test_oracle_golden.py::test_oracle_vs_live_reference_on_random_graphs) pins the oracle against
the live reference on the same generators; this script closes the triangle on the device: for every seed, random Pose3 /
projection (+- body_P_sensor) / BAL / Pose2 graphs with every noise kind and a different m-estimator per seed --
error, whitened Jacobians, Hessian diagonal, one damped solve per damping mode (Cholesky) and, where the graph has landmarks
or poses only, the LM trajectory.  Prints one JSON line per graph and a summary; exit code 1 on any violation.
Tolerances are those of tests/test_gpu_parity.py.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(float(np.abs(b).max()), 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=16)
    args = ap.parse_args()
    from gtsam_amd import datasets as D
    from gtsam_amd.lib import DeviceGraph
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt
    from gtsam_amd.params import LevenbergMarquardtParams as LMP
    from oracle import gtsam_oracle as O
    from tests import problems as PB
    from tests.test_oracle_golden import _bal_with_gauge_priors, _random_pose2_graph
    estimators = [(0, 0.0), (1, 1.3998), (2, 1.345), (3, 3.0), (4, 4.6851), (5, 2.9846), (6, 5.0), (0, 0.0)]
    bad = 0
    for seed in range(args.seeds):
        rk = estimators[seed % 8]
        graphs = {"pose3": D.random_pose_graph(8 + seed, 3 + seed % 3, seed=100 + seed, rot_scale=0.6 + 0.2 * (seed % 4)),
                  "projection": D.random_projection_graph(n_poses=4 + seed % 3, n_points=25, seed=200 + seed, with_sensor=bool(seed % 2)),
                  "bal": _bal_with_gauge_priors(300 + seed), "pose2": _random_pose2_graph(400 + seed)}
        for name, (p, v0) in graphs.items():
            if rk[0]:
                p, v0 = PB.robustify((p, v0), rk[0], rk[1])
            g = DeviceGraph(p); g.set_values(v0)
            rec = {"seed": seed, "graph": name, "estimator": rk[0]}
            e = g.error(); rec["error"] = abs(e - O.error(p, v0)) / abs(e)
            g.linearize()
            rec["jacobians"] = max(rel(g.jacobians(ft), O.jacobians_flat(p, v0, ft)) for ft in range(4)
                                   if {0: p.n_sfm, 1: p.n_proj, 2: p.n_between, 3: p.n_prior}[ft])
            rec["hessian_diagonal"] = rel(g.hessian_diagonal(), O.hessian_diagonal(p, v0))
            worst = 0.0
            for lam, dd in ((1e-2, False), (1e-3, True)):
                rc, out = g.try_lambda(lam, dd)
                st, d, _, _, lin = O.solve_damped(p, v0, lam, dd)
                if rc != st:
                    worst = float("inf")
                elif st == 0:
                    worst = max(worst, rel(g.delta(), d))
            rec["delta"] = worst
            g.close()
            params = LMP(); params.setMaxIterations(6)
            dev = DeviceLevenbergMarquardt(p, v0, params); dev.optimize()
            ref = O.lm_optimize(p, v0, params)
            tr = np.array(dev.trace)[:, :3]
            rec["trajectory"] = (rel(tr[:, 1], ref["trace"][:, 1]) if tr.shape == ref["trace"][:, :3].shape and np.array_equal(tr[:, 0], ref["trace"][:, 0])
                                 else float("inf"))
            ok = rec["error"] <= 1e-9 and rec["jacobians"] <= 1e-11 and rec["hessian_diagonal"] <= 1e-10 and rec["delta"] <= 1e-6 and rec["trajectory"] <= 1e-6
            rec["ok"] = bool(ok); bad += not ok
            print(json.dumps(rec), flush=True)
    print(json.dumps({"graphs": 4 * args.seeds, "violations": bad}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
