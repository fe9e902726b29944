"""Shared builders: the Problem + initial values behind each golden fixture (tests only)."""
import numpy as np

from gtsam_amd import datasets as D
from gtsam_amd.problem import (NOISE_DIAGONAL, NOISE_ISOTROPIC, NOISE_UNIT, bal_problem, pose_graph_problem)


def dubrovnik_timesfm(g):
    """timing/timeSFMBAL.cpp:33-55 protocol on dubrovnik-3-7-pre: Unit(2) noise, no priors."""
    return bal_problem(g["cams"], g["pts"], g["obs_cam"], g["obs_pt"], g["obs_z"])


def dubrovnik_sfmexample(g):
    """examples/SFMExample_bal.cpp:52-76 protocol: priors sigma 0.1 on C(0) and P(0)."""
    p, v0 = bal_problem(g["cams"], g["pts"], g["obs_cam"], g["obs_pt"], g["obs_z"])
    n9 = p.add_noise(NOISE_ISOTROPIC, 9, [0.1]); n3 = p.add_noise(NOISE_ISOTROPIC, 3, [0.1])
    p.add_prior(0, g["cams"][0], n9); p.add_prior(g["cams"].shape[0], g["pts"][0], n3)
    return p, v0


def sphere2500(g):
    """examples/Pose3SLAMExample_g2o.cpp:28-60 protocol (prior Variances(1e-6 x3, 1e-4 x3) on key 0)."""
    n = int(max(g["v1"].max(), g["v2"].max())) + 1
    p = pose_graph_problem(n, g["v1"], g["v2"], g["z"], g["noise_kind"], g["noise"])
    npri = p.add_noise(NOISE_DIAGONAL, 6, np.sqrt([1e-6] * 3 + [1e-4] * 3))
    p.add_prior(0, np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0.0]), npri)
    return p, g["values0"]


SYNTH = {
    "posegraph_small": lambda: D.random_pose_graph(14, 6, seed=3),
    "posegraph_bigrot": lambda: D.random_pose_graph(10, 4, seed=5, rot_scale=1.8, init_noise=0.4),
    "projection_small": lambda: D.random_projection_graph(seed=2),
    "bal_small_unit": lambda: bal_problem(*D.synthetic_bal(12, 300, seed=1, n_loops=1), (NOISE_UNIT, ())),
    "bal_small_iso": lambda: bal_problem(*D.synthetic_bal(12, 300, seed=1, n_loops=1), (NOISE_ISOTROPIC, [0.7])),
}
SYNTH_ORDERING = {"posegraph_small": 0, "posegraph_bigrot": 0, "projection_small": 1, "bal_small_unit": 1, "bal_small_iso": 1}
