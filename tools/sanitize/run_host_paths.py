#!/usr/bin/env python3
"""Drives the HOST code of the library (upload + symbolic analysis, one linearize / try_lambda / PCG issue sequence, plan
getters; single handle and two shards) under tools/hipstub with a sanitizer-instrumented build: tools/sanitize/host_sanitizers.sh.
Development tool; nothing numeric is computed (kernels do not run under the stub)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gtsam_amd import lib as L
if os.environ.get("SAN_LIB"):
    L.LIB_PATH = os.environ["SAN_LIB"]        # the sanitizer-instrumented build (host_sanitizers.sh); default: the product library
from tools import host_profile as HP
import numpy as np
def edge_cases():
    """Degenerate graphs through the whole host path: a single pose, a variable without factors, no factors at all, landmarks
    only, a landmark with one observation, camera counts around the reordering threshold and the 128-column tile boundary."""
    from gtsam_amd.problem import Problem, NOISE_UNIT, VAR_POSE3, VAR_POINT3, VAR_POSE2, bal_problem
    from gtsam_amd import datasets as D
    ident = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])
    def run(name, p, v0):
        g = L.DeviceGraph(p); g.set_values(np.ascontiguousarray(v0, float)); e = g.error(); g.linearize(); rc, out = g.try_lambda(1e-3, True)
        pl = g.cholesky_plan(); g.close(); print("ok", name, "nt", pl["nt"], flush=True)
    # 1: a single pose with a prior
    p = Problem(var_type=np.array([VAR_POSE3], np.int32)); p.add_prior(0, ident, p.add_noise(NOISE_UNIT, 6)); run("one pose + prior", p, ident)
    # 2: a variable without any factor
    p = Problem(var_type=np.array([VAR_POSE3, VAR_POSE3], np.int32)); p.add_prior(0, ident, p.add_noise(NOISE_UNIT, 6)); run("pose without factors", p, np.concatenate([ident, ident]))
    # 3: no factors at all
    p = Problem(var_type=np.array([VAR_POSE2], np.int32)); run("no factors", p, np.zeros(3))
    # 4: only landmarks (no reduced variables)
    p = Problem(var_type=np.array([VAR_POINT3, VAR_POINT3], np.int32)); p.add_prior(0, np.zeros(3), p.add_noise(NOISE_UNIT, 3)); run("landmarks only", p, np.zeros(6))
    # 5: BAL where a landmark has a single observation and one camera sees nothing
    cams, pts, oc, op, oz = D.synthetic_bal(8, 50, seed=2)[:5]
    keep = np.ones(len(oc), bool); keep[np.where(op == op[0])[0][1:]] = False
    p, v0 = bal_problem(cams, pts, oc[keep], op[keep], np.asarray(oz).reshape(-1, 2)[keep]); run("single-observation landmark", p, v0)
    # 6: exactly 16 / 17 cameras (the reordering threshold), 128-boundary sizes
    for nc in (14, 15, 16, 17, 29):
        p, v0 = bal_problem(*D.synthetic_bal(nc, 200, seed=nc)); run(f"bal {nc} cameras", p, v0)


for w in sys.argv[1:]:
    if w == "edge":
        edge_cases(); continue
    nd = None
    problem, v0 = HP.problem_for(w)
    for shards in (1, 2):
        for shard in range(shards):
            def lockstep(ptr, n, stream, shards=shards):
                buf = np.frombuffer((ctypes.c_double * n).from_address(ptr), dtype=np.float64); buf *= shards
            g = L.DeviceGraph(problem, shard=shard, n_shards=shards, allreduce=lockstep if shards > 1 else None)
            g.set_values(v0); g.linearize(); rc, out = g.try_lambda(1e-3, True)
            pl = g.cholesky_plan()
            if shards == 1:
                rc2, out2, its = g.try_lambda_pcg(1e-3, True, max_iterations=5)
            g.close()
    print("ok", w, flush=True)
