#!/usr/bin/env python3
"""tests/golden/make_golden.py -- generates the committed golden fixtures from the REAL reference.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Every number written here is computed by borglab/gtsam's own code (oracle/_ref, see oracle/Makefile and
oracle/ref_harness.cpp); inputs are the reference's shipped datasets parsed by the reference's own
loaders, or seeded synthetic graphs from gtsam_amd/datasets.py.  The .npz files travel to the GPU box
(where /root/reference does not exist) and pin both the numpy restatement (oracle/gtsam_oracle.py) and
the HIP path.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gtsam_amd import datasets as D  # noqa: E402
from gtsam_amd.params import LevenbergMarquardtParams as LMP  # noqa: E402
from gtsam_amd.problem import (NOISE_DIAGONAL, NOISE_ISOTROPIC, NOISE_UNIT, bal_problem,  # noqa: E402
                               pose_graph_problem)
from oracle import ref  # noqa: E402

DATA = "/root/reference/examples/Data/"
OUT = os.path.dirname(os.path.abspath(__file__))


def chain_init(n, v1, v2, z):
    """Initial values for a TORO file without VERTEX lines: chain the odometry edges i -> i+1 from
    identity (what matlab/+gtsam/load3D.m successive=true does)."""
    poses = np.zeros((n, 12)); poses[0, [0, 4, 8]] = 1.0
    have = np.zeros(n, bool); have[0] = True
    for a, b, zz in zip(v1, v2, z):
        if b == a + 1 and have[a] and not have[b]:
            Ra, ta = poses[a, :9].reshape(3, 3), poses[a, 9:]
            Rz, tz = zz[:9].reshape(3, 3), zz[9:]
            poses[b, :9] = (Ra @ Rz).reshape(-1); poses[b, 9:] = ta + Ra @ tz
            have[b] = True
    assert have.all()
    return poses


def probes(g, p, v, lams=((1e-3, False), (1e-4, True)), ordering_kind=1):
    out = {"error": g.error(v), "hessian_diagonal": g.hessian_diagonal(v)}
    for ft, n in ((0, p.n_sfm), (1, p.n_proj), (2, p.n_between), (3, p.n_prior)):
        if n:
            out[f"jac{ft}"] = g.jacobians(v, ft)
    for i, (lam, dd) in enumerate(lams):
        rc, delta, le = g.solve(v, lam, dd, ordering_kind=ordering_kind)
        out[f"solve{i}_lambda"] = lam; out[f"solve{i}_diag"] = dd; out[f"solve{i}_status"] = rc
        out[f"solve{i}_delta"] = delta; out[f"solve{i}_linerr"] = le
        if rc == 0:
            out[f"solve{i}_retract"] = g.retract(v, delta)
            out[f"solve{i}_trial_error"] = g.error(out[f"solve{i}_retract"])
    return out


def main():
    # ---- dubrovnik-3-7-pre: the reference's shipped BAL file (configs[1]) ---------------------------------
    cams, pts, oc, op, oz = ref.load_bal(DATA + "dubrovnik-3-7-pre.txt")
    p, v0 = bal_problem(cams, pts, oc, op, oz)                      # timeSFMBAL protocol: Unit(2), no priors
    g = ref.RefGraph(p)
    out = dict(cams=cams, pts=pts, obs_cam=oc, obs_pt=op, obs_z=oz, values0=v0)
    out.update({"timesfm_" + k: v for k, v in probes(g, p, v0).items()})
    r = g.lm(v0, LMP.CeresDefaults(), ordering_kind=1)               # timing/timeSFMBAL.h:64-95
    out["timesfm_trace"] = r["trace"][:, :3]; out["timesfm_values"] = r["values"]; out["timesfm_iterations"] = r["iterations"]
    r = g.lm(v0, LMP(), ordering_kind=0)                             # tests/testGeneralSFMFactorB.cpp:44-63
    out["default_trace"] = r["trace"][:, :3]; out["default_values"] = r["values"]; out["default_iterations"] = r["iterations"]
    # SFMExample_bal protocol: Isotropic(2,1.0)->Unit, priors sigma 0.1 on C0 and P0, legacy LM, COLAMD
    p2, _ = bal_problem(cams, pts, oc, op, oz)
    n9 = p2.add_noise(NOISE_ISOTROPIC, 9, [0.1]); n3 = p2.add_noise(NOISE_ISOTROPIC, 3, [0.1])
    p2.add_prior(0, cams[0], n9); p2.add_prior(cams.shape[0], pts[0], n3)
    g2 = ref.RefGraph(p2)
    out.update({"sfmex_" + k: v for k, v in probes(g2, p2, v0).items()})
    r = g2.lm(v0, LMP(), ordering_kind=0)                            # examples/SFMExample_bal.cpp:39-89
    out["sfmex_trace"] = r["trace"][:, :3]; out["sfmex_values"] = r["values"]; out["sfmex_iterations"] = r["iterations"]
    np.savez_compressed(os.path.join(OUT, "dubrovnik_3_7.npz"), **out)
    print("dubrovnik_3_7: init", out["timesfm_error"], "timesfm final", out["timesfm_trace"][-1],
          "default final", out["default_trace"][-1], "sfmex final", out["sfmex_trace"][-1])

    # ---- sphere2500 (configs[3]): TORO EDGE3 file, chained init, prior, legacy LM ---------------------------
    d = ref.load_g2o3d(DATA + "sphere2500.txt")
    n = int(max(d["v1"].max(), d["v2"].max())) + 1
    v0 = chain_init(n, d["v1"], d["v2"], d["z"])
    p = pose_graph_problem(n, d["v1"], d["v2"], d["z"], d["noise_kind"], d["noise"])
    npri = p.add_noise(NOISE_DIAGONAL, 6, np.sqrt([1e-6] * 3 + [1e-4] * 3))   # Pose3SLAMExample_g2o.cpp:41-43
    p.add_prior(0, np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0.0]), npri)
    g = ref.RefGraph(p)
    r = g.lm(v0.reshape(-1), LMP(), ordering_kind=0)
    rc, delta, le = g.solve(v0.reshape(-1), 1e-5, False, ordering_kind=0)
    np.savez_compressed(os.path.join(OUT, "sphere2500.npz"), v1=d["v1"].astype(np.int32), v2=d["v2"].astype(np.int32),
                        z=d["z"], noise_kind=d["noise_kind"], noise=d["noise"], values0=v0.reshape(-1),
                        error0=g.error(v0.reshape(-1)), trace=r["trace"][:, :3], iterations=r["iterations"],
                        final_values=r["values"], solve_delta=delta, solve_linerr=le, solve_status=rc,
                        ref_seconds=r["seconds"])
    print("sphere2500: init", g.error(v0.reshape(-1)), "final", r["trace"][-1], "outer", r["iterations"], "sec", r["seconds"])

    # ---- seeded synthetic graphs ---------------------------------------------------------------------------
    for name, (p, v0), ok in (("posegraph_small", D.random_pose_graph(14, 6, seed=3), 0),
                              ("posegraph_bigrot", D.random_pose_graph(10, 4, seed=5, rot_scale=1.8, init_noise=0.4), 0),
                              ("projection_small", D.random_projection_graph(seed=2), 1)):
        g = ref.RefGraph(p)
        out = {"values0": v0}
        out.update(probes(g, p, v0, ordering_kind=ok))
        r = g.lm(v0, LMP(), ordering_kind=ok)
        out["trace"] = r["trace"][:, :3]; out["final_values"] = r["values"]; out["iterations"] = r["iterations"]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "init", out["error"], "final", r["trace"][-1])

    c = D.synthetic_bal(12, 300, seed=1, n_loops=1)
    for name, noise in (("bal_small_unit", (NOISE_UNIT, ())), ("bal_small_iso", (NOISE_ISOTROPIC, [0.7]))):
        p, v0 = bal_problem(*c, noise)
        g = ref.RefGraph(p)
        out = {"values0": v0}
        out.update(probes(g, p, v0))
        r = g.lm(v0, LMP.CeresDefaults(), ordering_kind=1)
        out["trace"] = r["trace"][:, :3]; out["final_values"] = r["values"]; out["iterations"] = r["iterations"]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "init", out["error"], "final", r["trace"][-1])


def cal3ds2():
    """projection_ds2: GenericProjectionFactor<Pose3, Point3, Cal3DS2> through the real reference (same probes + LM trace)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import problems as PB
    name = "projection_ds2"
    p, v0 = PB.SYNTH[name]()
    g = ref.RefGraph(p)
    out = {"values0": v0}
    out.update(probes(g, p, v0, ordering_kind=1))
    r = g.lm(v0, LMP(), ordering_kind=1)
    out["trace"] = r["trace"][:, :3]; out["final_values"] = r["values"]; out["iterations"] = r["iterations"]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "init", out["error"], "final", r["trace"][-1])


def smart():
    """smart_orbit*.npz: SmartProjectionFactor graphs through the real reference -- error, Hessian diagonal (of the Schur-complemented
    Hessian factors), two damped solves, LM traces with the Ceres and the legacy parameters."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import problems as PB
    for name, mk in PB.SMART.items():
        p, v0 = mk()
        g = ref.RefGraph(p)
        out = {"values0": v0, "error": g.error(v0), "hessian_diagonal": g.hessian_diagonal(v0)}
        for i, (lam, dd) in enumerate(((1e-3, False), (1e-4, True))):
            rc, delta, le = g.solve(v0, lam, dd, ordering_kind=0)
            out[f"solve{i}_lambda"] = lam; out[f"solve{i}_diag"] = dd; out[f"solve{i}_status"] = rc
            out[f"solve{i}_delta"] = delta; out[f"solve{i}_linerr"] = le
            out[f"solve{i}_retract"] = g.retract(v0, delta); out[f"solve{i}_trial_error"] = g.error(out[f"solve{i}_retract"])
        for tag, prm in (("ceres", LMP.CeresDefaults()), ("legacy", LMP())):
            r = g.lm(v0, prm, ordering_kind=0)
            out[tag + "_trace"] = r["trace"][:, :3]; out[tag + "_values"] = r["values"]; out[tag + "_iterations"] = r["iterations"]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "init", out["error"], "ceres", out["ceres_trace"][-1], "legacy", out["legacy_trace"][-1])


def robust():
    """m-estimator fixtures: graphs of tests/problems.py ROBUST_SYNTH through the real noiseModel::Robust."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import problems as PB
    for name in PB.ROBUST_SYNTH:
        p, v0 = PB.SYNTH[name]()
        ok = PB.SYNTH_ORDERING[name]
        g = ref.RefGraph(p)
        out = {"values0": v0}
        out.update(probes(g, p, v0, ordering_kind=ok))
        prm = LMP()
        r = g.lm(v0, prm, ordering_kind=ok)
        out["trace"] = r["trace"][:, :3]; out["final_values"] = r["values"]; out["iterations"] = r["iterations"]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "init", out["error"], "final", r["trace"][-1], "outer", r["iterations"])


def pose2():
    """Pose2 pose graphs (BASELINE configs[0]: Pose2SLAMExample_g2o protocol, LM instead of GN): the reference's
    shipped w100.graph and noisyToyGraph.txt in full, and w20000.txt (the substitute for the absent w10000, SURVEY.md
    section 8(d) config 1; golden 32 626 834.02 -> 13 520 404.37, gives up at lambda max after 24 inner iterations)."""
    from gtsam_amd.problem import pose2_graph_problem
    for name, fn, full in (("pose2_w100", "w100.graph", True), ("pose2_toy", "noisyToyGraph.txt", True),
                           ("pose2_w20000", "w20000.txt", False)):
        d = ref.load_2d(DATA + fn)
        n = int(max(d["v1"].max(), d["v2"].max())) + 1
        p = pose2_graph_problem(n, d["v1"], d["v2"], d["z"], d["noise_kind"], d["noise"])
        npri = p.add_noise(NOISE_DIAGONAL, 3, np.sqrt([1e-6, 1e-6, 1e-8]))          # Pose2SLAMExample_g2o.cpp:65-67
        p.add_prior(0, np.zeros(3), npri)
        v0 = np.zeros((n, 3)); v0[d["vertex_keys"]] = d["vertex_poses"]; v0 = v0.reshape(-1)
        g = ref.RefGraph(p)
        out = dict(v1=d["v1"].astype(np.int32), v2=d["v2"].astype(np.int32), z=d["z"], noise_kind=d["noise_kind"],
                   noise=d["noise"], values0=v0)
        if full:
            out.update(probes(g, p, v0, ordering_kind=0))
        else:
            out["error"] = g.error(v0)
            out["jac2_head"] = g.jacobians(v0, 2)[:512]
        r = g.lm(v0, LMP(), ordering_kind=0)
        out["trace"] = r["trace"][:, :3]; out["final_values"] = r["values"]; out["iterations"] = r["iterations"]
        out["ref_seconds"] = r["seconds"]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "init", out["error"], "final", r["trace"][-1], "outer", r["iterations"])


def logfile():
    """LevenbergMarquardtParams::logFile: the CSV the reference's own optimize() writes (LevenbergMarquardtOptimizer.cpp:
    101-118, rows appended in iterate() :283-303) on dubrovnik-3-7-pre and on the noisy pose graph; columns
    (inner iterations, seconds, error, lambda, outer iterations) at the stream's default precision."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests import problems as PB
    g = dict(np.load(os.path.join(OUT, "dubrovnik_3_7.npz")))
    out = {}
    p, v0 = PB.dubrovnik_timesfm(g)
    out["timesfm_ceres"] = ref.RefGraph(p).lm_logfile(v0, LMP.CeresDefaults(), 1)
    out["timesfm_legacy"] = ref.RefGraph(p).lm_logfile(v0, LMP(), 0)
    p, v0 = PB.dubrovnik_sfmexample(g)
    out["sfmex_legacy"] = ref.RefGraph(p).lm_logfile(v0, LMP(), 0)
    np.savez_compressed(os.path.join(OUT, "lm_logfile.npz"), **out)
    for k, v in out.items():
        print("logfile", k, v.shape, "last row", v[-1])


if __name__ == "__main__":
    if "--logfile-only" in sys.argv:
        logfile()
    elif "--pose2-only" in sys.argv:
        pose2()
    elif "--robust-only" in sys.argv:
        robust()
    elif "--cal3ds2-only" in sys.argv:
        cal3ds2()
    elif "--smart-only" in sys.argv:
        smart()
    else:
        main()
        cal3ds2()
        smart()
        robust()
        pose2()
        logfile()
