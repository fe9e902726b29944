"""Shared builders: the Problem + initial values behind each golden fixture (tests only)."""
import numpy as np

from gtsam_amd import datasets as D
from gtsam_amd.problem import (NOISE_DIAGONAL, NOISE_ISOTROPIC, NOISE_UNIT, ROBUST_CAUCHY, ROBUST_FAIR, ROBUST_HUBER,
                               ROBUST_TUKEY, ROBUST_WELSCH, ROBUST_GEMANMCCLURE, ROBUST_DCS, ROBUST_L2WITHDEADZONE, bal_problem,
                               pose_graph_problem)


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


def robustify(pv, rkind, k):
    """Wrap every noise model of the graph in noiseModel::Robust(mEstimator(k), base) (linear/NoiseModel.h:670-760)."""
    p, v0 = pv
    p.noise_robust = np.full(p.noise_kind.size, rkind, np.int32)
    p.noise_robust_param = np.full(p.noise_kind.size, k, np.float64)
    return p, v0


def _dubrovnik_robust(rkind, k):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dubrovnik_3_7.npz"))
    return robustify(dubrovnik_timesfm(g), rkind, k)


SYNTH = {
    "posegraph_small": lambda: D.random_pose_graph(14, 6, seed=3),
    "posegraph_bigrot": lambda: D.random_pose_graph(10, 4, seed=5, rot_scale=1.8, init_noise=0.4),
    "projection_small": lambda: D.random_projection_graph(seed=2),
    # GenericProjectionFactor<Pose3, Point3, Cal3DS2> (section 8(f) #3): calibration 0 with radial + tangential distortion,
    # calibration 1 a plain Cal3_S2 -- both branches of the projection in one graph
    "projection_ds2": lambda: D.random_projection_graph(seed=2, distortion=[[-0.12, 0.03, 1.5e-3, -2e-3], [0, 0, 0, 0]]),
    "bal_small_unit": lambda: bal_problem(*D.synthetic_bal(12, 300, seed=1, n_loops=1), (NOISE_UNIT, ())),
    "bal_small_iso": lambda: bal_problem(*D.synthetic_bal(12, 300, seed=1, n_loops=1), (NOISE_ISOTROPIC, [0.7])),
    # m-estimators (section 8(f) #2): every loss function of linear/LossFunctions.cpp on some graph
    "posegraph_huber": lambda: robustify(D.random_pose_graph(14, 6, seed=3), ROBUST_HUBER, 1.345),
    "posegraph_fair": lambda: robustify(D.random_pose_graph(14, 6, seed=3), ROBUST_FAIR, 1.3998),
    "posegraph_welsch": lambda: robustify(D.random_pose_graph(14, 6, seed=3), ROBUST_WELSCH, 2.9846),
    "projection_cauchy": lambda: robustify(D.random_projection_graph(seed=2), ROBUST_CAUCHY, 3.0),
    "projection_tukey": lambda: robustify(D.random_projection_graph(seed=2), ROBUST_TUKEY, 4.6851),
    "projection_gm": lambda: robustify(D.random_projection_graph(seed=2), ROBUST_GEMANMCCLURE, 5.0),
    # (round 6) DCS -- its parameter is compared with the SQUARED distance -- and the dead zone, inside which a factor drops out of the system
    "posegraph_dcs": lambda: robustify(D.random_pose_graph(14, 6, seed=3), ROBUST_DCS, 1.0),
    "projection_deadzone": lambda: robustify(D.random_projection_graph(seed=2), ROBUST_L2WITHDEADZONE, 0.75),
    "dubrovnik_huber": lambda: _dubrovnik_robust(ROBUST_HUBER, 1.345),
    "dubrovnik_cauchy": lambda: _dubrovnik_robust(ROBUST_CAUCHY, 5.0),
}


def smart_orbit(degenerate=False, enable_epi=False):
    """SmartProjectionFactor<PinholeCamera<Cal3Bundler>> graphs (section 8(f) #3), built like timing/timeSFMBALsmart.cpp from
    D.synthetic_orbit_scene: cameras are the only variables.  degenerate=True: ZERO_ON_DEGENERACY with tracks that do not
    triangulate -- single observations (m < 2), a landmark-distance threshold that rejects the far half of the cloud, a
    dynamic outlier threshold that rejects tracks with a bad measurement."""
    from gtsam_amd.problem import smart_bal_problem
    # (enable_epi: TriangulationParameters::enableEPI, every triangulation refined by the reference's LM on TriangulationFactors; four
    # times the pixel noise, so that the refinement moves the points)
    cams, pts, oc, op, oz = D.synthetic_orbit_scene(seed=3 if degenerate else 0, pixel_noise=2.0 if enable_epi else 0.5)
    if not degenerate:
        return smart_bal_problem(cams, oc, op, oz, enable_epi=enable_epi)
    oz = oz.copy()
    first = np.flatnonzero(np.r_[True, np.diff(op) != 0])
    oz[first[::9]] += 150.0                                              # a bad measurement in every 9th track
    keep = np.ones(oc.size, bool)
    for j in range(5, 120, 11):                                          # these tracks keep a single observation
        idx = np.flatnonzero(op == j); keep[idx[1:]] = False
    return smart_bal_problem(cams, oc[keep], op[keep], oz[keep], degeneracy_mode=1, landmark_distance_threshold=10.5,
                             dynamic_outlier_rejection_threshold=60.0, enable_epi=enable_epi)


def smart_far(degeneracy_mode, linearization_mode=0, arc=0.5, init_noise=(0.01, 0.05), spread=1.0, enable_epi=False):
    """The degeneracy modes that replace a failed track by a POINT AT INFINITY (SmartProjectionFactor.h:356-371, :419-427), on a
    scene where that is a sensible model: a quarter of the landmarks lie thousands of units behind the cloud and fail the
    landmark-distance threshold (FAR_POINT), some tracks keep a single observation (DEGENERATE).  (A NEAR track seen from infinity
    has residuals of hundreds of pixels and sends the reference's LM into steps where Cal3Bundler::calibrate throws; outlier
    tracks are in smart_orbit_degenerate.)  HANDLE_INFINITY (2): the point at infinity in the linearisation and in the error; IGNORE_DEGENERACY
    (0, the reference's default): in the linearisation only, error 0.0; with a Jacobian linearisation mode (2 JACOBIAN_Q,
    3 JACOBIAN_SVD) a failed track is an empty factor whatever the degeneracy mode.  The arc is narrow: the direction of a
    track's first measurement must lie in front of every camera of the track, or the reference throws a CheiralityException out of
    linearize() / error() (CalibratedCamera.cpp:146-149) -- arc = 0.9 is that case."""
    from gtsam_amd.problem import smart_bal_problem
    cams, pts, oc, op, oz = D.synthetic_orbit_scene(seed=7, arc=arc, far_points=30, init_noise=init_noise, spread=spread)
    keep = np.ones(oc.size, bool)
    for j in range(5, 120, 17):                                          # these tracks keep a single observation
        idx = np.flatnonzero(op == j); keep[idx[1:]] = False
    p, v0 = smart_bal_problem(cams, oc[keep], op[keep], oz[keep], degeneracy_mode=degeneracy_mode, linearization_mode=linearization_mode,
                              landmark_distance_threshold=100.0, enable_epi=enable_epi)
    # priors on the two end cameras fix the gauge: without them the end game of the legacy parameters (identity damping down to
    # lambda = 1e-8 on a system with seven flat directions) is decided by the rounding of the solve, in the reference as well
    ni = p.add_noise(NOISE_ISOTROPIC, 9, [0.05])
    p.add_prior(0, v0[:17], ni); p.add_prior(p.n_vars - 1, v0[-17:], ni)
    return p, v0


SMART = {"smart_orbit": lambda: smart_orbit(False), "smart_orbit_degenerate": lambda: smart_orbit(True),
         "smart_orbit_epi": lambda: smart_orbit(False, enable_epi=True),
         "smart_far_infinity": lambda: smart_far(2), "smart_far_ignore": lambda: smart_far(0),
         # (closer initial values: without the far tracks in the linear system the first steps from the noisier start leave the region
         # where Cal3Bundler::calibrate converges, and the reference throws)
         "smart_far_jacobian_q": lambda: smart_far(0, linearization_mode=2, init_noise=(0.002, 0.01), spread=1.8),
         "smart_far_jacobian_svd": lambda: smart_far(2, linearization_mode=3, init_noise=(0.002, 0.01), spread=1.8)}

ROBUST_SYNTH = ("posegraph_huber", "posegraph_fair", "posegraph_welsch", "projection_cauchy", "projection_tukey",
                "projection_gm", "dubrovnik_huber", "dubrovnik_cauchy", "posegraph_dcs", "projection_deadzone")
SYNTH_ORDERING = {"posegraph_small": 0, "posegraph_bigrot": 0, "projection_small": 1, "projection_ds2": 1, "bal_small_unit": 1, "bal_small_iso": 1,
                  "posegraph_huber": 0, "posegraph_fair": 0, "posegraph_welsch": 0, "projection_cauchy": 1,
                  "projection_tukey": 1, "projection_gm": 1, "dubrovnik_huber": 1, "dubrovnik_cauchy": 1, "posegraph_dcs": 0,
                  "projection_deadzone": 1}


def robust_prior_literal():
    """tests/testRobust.cpp:31-49 (RobustNoise.loss) carried to the path's types: a PriorFactor<Point3> at the origin
    with Robust(GemanMcClure(1.0), Unit) evaluated at (10, 0, 0): whitened distance 10 -> error 0.49505 (+-1e-5)."""
    from gtsam_amd.problem import Problem, VAR_POINT3
    p = Problem(var_type=np.array([VAR_POINT3], np.int32))
    n = p.add_noise(NOISE_UNIT, 3, (), robust=(ROBUST_GEMANMCCLURE, 1.0))
    p.add_prior(0, np.zeros(3), n)
    return p, np.array([10.0, 0.0, 0.0])


def pose2_graph(g):
    """Pose2SLAMExample_g2o.cpp:46-67 protocol: BetweenFactor<Pose2> edges from load2D + prior Variances(1e-6, 1e-6, 1e-8)
    on key 0."""
    from gtsam_amd.problem import pose2_graph_problem
    n = int(max(g["v1"].max(), g["v2"].max())) + 1
    p = pose2_graph_problem(n, g["v1"], g["v2"], g["z"], g["noise_kind"], g["noise"])
    npri = p.add_noise(NOISE_DIAGONAL, 3, np.sqrt([1e-6, 1e-6, 1e-8]))
    p.add_prior(0, np.zeros(3), npri)
    return p, g["values0"]
