"""The host API mirror (gtsam_amd/api.py): extractor + noise-model factories on CPU; on the GPU an LM run written the
way examples/SFMExample_bal.cpp / tests/testGeneralSFMFactorB.cpp write it."""
import numpy as np
import pytest

from gtsam_amd import api
from gtsam_amd.problem import NOISE_DIAGONAL, NOISE_GAUSSIAN, NOISE_ISOTROPIC, NOISE_UNIT
from tests import problems as PB
from tests.conftest import load_golden

C, P, X = api.symbol_shorthand.C, api.symbol_shorthand.P, api.symbol_shorthand.X


def _dubrovnik_graph(priors):
    """examples/SFMExample_bal.cpp:52-76, from the golden copy of what SfmData::FromBalFile returned."""
    g = load_golden("dubrovnik_3_7")
    graph = api.NonlinearFactorGraph(); initial = api.Values()
    noise = api.noiseModel.Isotropic.Sigma(2, 1.0)          # smart -> Unit, as in the reference
    for i, j, uv in zip(g["obs_cam"], g["obs_pt"], g["obs_z"]):
        graph.add(api.GeneralSFMFactorCal3Bundler(uv, noise, C(int(i)), P(int(j))))
    cams = [api.PinholeCameraCal3Bundler.from_packed(c) for c in g["cams"]]
    if priors:
        graph.addPriorPinholeCameraCal3Bundler(C(0), cams[0], api.noiseModel.Isotropic.Sigma(9, 0.1))
        graph.addPriorPoint3(P(0), g["pts"][0], api.noiseModel.Isotropic.Sigma(3, 0.1))
    for i, c in enumerate(cams):
        initial.insert(C(i), c)
    for j, p in enumerate(g["pts"]):
        initial.insert(P(j), np.array(p))
    return g, graph, initial


def test_symbol_and_noise_factories():
    assert C(3) == (ord("c") << 56) | 3 and P(0) > C(10 ** 6)          # Values order: cameras before points
    nm = api.noiseModel
    assert nm.Isotropic.Sigma(2, 1.0).kind == NOISE_UNIT and nm.Isotropic.Sigma(2, 0.5).kind == NOISE_ISOTROPIC
    assert nm.Diagonal.Sigmas([0.1, 0.1, 0.1]).kind == NOISE_ISOTROPIC and nm.Diagonal.Sigmas([1, 1]).kind == NOISE_UNIT
    d = nm.Diagonal.Variances([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4])
    assert d.kind == NOISE_DIAGONAL and np.allclose(d.params, np.sqrt([1e-6] * 3 + [1e-4] * 3))
    info = np.diag([10.0, 10, 10, 100, 100, 25])
    assert nm.Gaussian.Information(info).kind == NOISE_DIAGONAL        # sphere2500's edges
    info[0, 1] = info[1, 0] = 1.0
    ga = nm.Gaussian.Information(info)
    assert ga.kind == NOISE_GAUSSIAN and np.allclose(ga.params.reshape(6, 6).T @ ga.params.reshape(6, 6), info)
    with pytest.raises(ValueError):
        nm.Diagonal.Sigmas([0.1, 1e-9])                                  # constrained: out of scope, rejected loudly


def test_robust_noise_model_mirror():
    """noiseModel.Robust.Create(mEstimator.Huber.Create(k), base) -> (ROBUST_HUBER, k) on the base model's table row;
    literals of linear/tests/testNoiseModel.cpp:466-485 (robust_Huber / robust_Tukey weights)."""
    nm = api.noiseModel
    hub = nm.mEstimator.Huber.Create(5.0)
    assert abs(hub.weight(1.0) - 1.0) < 1e-8 and abs(hub.weight(10.0) - 0.5) < 1e-8
    tuk = nm.mEstimator.Tukey.Create(5.0)
    assert abs(tuk.weight(1.0) - 0.9216) < 1e-8 and abs(tuk.weight(10.0) - 0.0) < 1e-8
    with pytest.raises(ValueError):
        nm.mEstimator.Cauchy.Create(0.0)
    graph = api.NonlinearFactorGraph(); vals = api.Values()
    vals.insert(X(0), api.Pose3()); vals.insert(X(1), api.Pose3())
    base = nm.Diagonal.Sigmas([0.1, 0.1, 0.1, 0.3, 0.3, 0.2])
    graph.add(api.BetweenFactorPose3(X(0), X(1), api.Pose3(), nm.Robust.Create(hub, base)))
    graph.add(api.BetweenFactorPose3(X(0), X(1), api.Pose3(), base))
    graph.add(api.BetweenFactorPose3(X(1), X(0), api.Pose3(), nm.Robust.Create(nm.mEstimator.Huber.Create(5.0), base)))
    p, _, _ = api.extract(graph, vals)
    assert list(p.noise_robust) == [2, 0] and list(p.noise_robust_param) == [5.0, 0.0]
    assert list(p.between_noise) == [0, 1, 0]


def test_extractor_reproduces_the_soa_problem():
    g, graph, initial = _dubrovnik_graph(priors=True)
    p, v0, keys = api.extract(graph, initial)
    q, w0 = PB.dubrovnik_sfmexample(g)
    assert keys == [C(i) for i in range(3)] + [P(j) for j in range(7)]
    assert np.array_equal(p.var_type, q.var_type) and np.array_equal(v0, w0)
    assert np.array_equal(p.sfm_cam, q.sfm_cam) and np.array_equal(p.sfm_point, q.sfm_point) and np.array_equal(p.sfm_z, q.sfm_z)
    assert p.n_prior == 2 and np.array_equal(p.prior_data, q.prior_data)
    assert [int(k) for k in p.noise_kind] == [NOISE_UNIT, NOISE_ISOTROPIC, NOISE_ISOTROPIC]
    bad = api.NonlinearFactorGraph(); bad.add(api.BetweenFactorPose3(X(0), X(1), api.Pose3(), api.noiseModel.Unit.Create(5)))
    vals = api.Values(); vals.insert(X(0), api.Pose3()); vals.insert(X(1), api.Pose3())
    with pytest.raises(ValueError):
        api.extract(bad, vals)                                           # noise dimension mismatch (NonlinearFactor.cpp:97-104)
    with pytest.raises(KeyError):
        g2 = api.NonlinearFactorGraph(); g2.add(api.BetweenFactorPose3(X(0), X(7), api.Pose3(), api.noiseModel.Unit.Create(6)))
        api.extract(g2, vals)


def test_cal3ds2_projection_factors_through_the_mirror():
    """GenericProjectionFactor<Pose3, Point3, Cal3DS2> next to <..., Cal3_S2> in one graph: the extractor hands the library one
    calibration table with the distortion coefficients beside it (zero row for the plain Cal3_S2), shared objects deduplicated."""
    nm = api.noiseModel
    Kd = api.Cal3DS2(520, 515, 0.3, 320, 240, -0.12, 0.03, 1.5e-3, -2e-3); K = api.Cal3_S2(400, 400, 0, 300, 200)
    graph = api.NonlinearFactorGraph(); vals = api.Values()
    vals.insert(X(0), api.Pose3()); vals.insert(X(1), api.Pose3()); vals.insert(P(0), np.array([0.1, 0.2, 5.0]))
    graph.add(api.GenericProjectionFactorCal3DS2([330.0, 250.0], nm.Unit.Create(2), X(0), P(0), Kd))
    graph.add(api.GenericProjectionFactorCal3_S2([310.0, 210.0], nm.Unit.Create(2), X(1), P(0), K))
    graph.add(api.GenericProjectionFactorCal3DS2([331.0, 251.0], nm.Unit.Create(2), X(1), P(0), Kd, api.Pose3()))
    p, _, _ = api.extract(graph, vals)
    assert list(p.proj_calib) == [0, 1, 0] and list(p.proj_sensor) == [-1, -1, 0]
    assert np.array_equal(p.calib.reshape(2, 5), [[520, 515, 0.3, 320, 240], [400, 400, 0, 300, 200]])
    assert np.array_equal(p.calib_distortion.reshape(2, 4), [[-0.12, 0.03, 1.5e-3, -2e-3], [0, 0, 0, 0]])
    c = p.to_ctypes()
    assert c.n_calib == 2 and c.calib_distortion[0] == -0.12 and c.calib_distortion[7] == 0.0


def test_smart_factors_through_the_mirror_give_the_soa_problem():
    """timing/timeSFMBALsmart.cpp:33-58 written with the mirror: one smart factor per track, cameras the only variables; the
    extractor must produce the tables of gtsam_amd.problem.smart_bal_problem (the fixture generator's input)."""
    from gtsam_amd import datasets as D
    from gtsam_amd.problem import smart_bal_problem
    cams, pts, oc, op, oz = D.synthetic_orbit_scene(n_cams=5, n_points=12, seed=4)
    q, w0 = smart_bal_problem(cams, oc, op, oz, degeneracy_mode=1, landmark_distance_threshold=9.0)
    graph = api.NonlinearFactorGraph(); vals = api.Values()
    for i in range(5):
        vals.insert(C(i), api.PinholeCameraCal3Bundler.from_packed(cams[i]))
    sp = api.SmartProjectionParams(api.SmartProjectionParams.HESSIAN, api.SmartProjectionParams.ZERO_ON_DEGENERACY); sp.setLandmarkDistanceThreshold(9.0)
    for j in range(12):
        f = api.SmartProjectionFactorPinholeCameraCal3Bundler(api.noiseModel.Unit.Create(2), sp)
        for k in np.flatnonzero(op == j):
            f.add(oz[k], C(int(oc[k])))
        graph.add(f)
    p, v0, keys = api.extract(graph, vals)
    assert np.array_equal(v0, w0) and p.n_smart == q.n_smart == 12
    for name in ("smart_ptr", "smart_cam", "smart_z", "smart_noise", "smart_params"):
        assert np.array_equal(getattr(p, name), getattr(q, name)), name
    c = p.to_ctypes()
    assert c.n_smart == 12 and c.smart_ptr[12] == p.smart_cam.size


@pytest.mark.gpu
def test_general_sfm_factor_B_written_like_the_reference():
    """tests/testGeneralSFMFactorB.cpp:44-63: default LM on dubrovnik-3-7-pre, no priors -> 0.0199833 +- 1e-5."""
    g, graph, initial = _dubrovnik_graph(priors=False)
    assert abs(graph.error(initial) - 2764.21929281) < 1e-6
    lm = api.LevenbergMarquardtOptimizer(graph, initial)
    actual = lm.optimize()
    assert abs(lm.error() - 0.0199833) < 1e-5 and abs(graph.error(actual) - lm.error()) < 1e-12
    assert lm.iterations() == int(g["default_iterations"])
    assert isinstance(actual.at(C(0)), api.PinholeCameraCal3Bundler) and actual.at(P(0)).shape == (3,)


def test_pose2_slam_example_written_like_the_reference():
    """examples/Pose2SLAMExample.cpp:60-105 built through the mirror; the extractor must give the same SoA problem as
    the direct builder (gtsam_amd.problem.pose2_graph_problem)."""
    from gtsam_amd.problem import VAR_POSE2, pose2_graph_problem
    nm = api.noiseModel
    graph = api.NonlinearFactorGraph()
    graph.addPriorPose2(1, api.Pose2(0, 0, 0), nm.Diagonal.Sigmas([0.3, 0.3, 0.1]))
    model = nm.Diagonal.Sigmas([0.2, 0.2, 0.1])
    edges = [(1, 2, (2, 0, 0)), (2, 3, (2, 0, np.pi / 2)), (3, 4, (2, 0, np.pi / 2)), (4, 5, (2, 0, np.pi / 2)), (5, 2, (2, 0, np.pi / 2))]
    for a, b, z in edges:
        graph.add(api.BetweenFactorPose2(a, b, api.Pose2(*z), model))
    initial = api.Values()
    guess = [(0.5, 0.0, 0.2), (2.3, 0.1, -0.2), (4.1, 0.1, np.pi / 2), (4.0, 2.0, np.pi), (2.1, 2.1, -np.pi / 2)]
    for k, g in enumerate(guess):
        initial.insert(k + 1, api.Pose2(*g))
    p, v0, keys = api.extract(graph, initial)
    assert keys == [1, 2, 3, 4, 5] and list(p.var_type) == [VAR_POSE2] * 5 and np.allclose(v0, np.array(guess).reshape(-1))
    q = pose2_graph_problem(5, [e[0] - 1 for e in edges], [e[1] - 1 for e in edges], [e[2] for e in edges],
                            [NOISE_DIAGONAL] * 5, [[0.2, 0.2, 0.1, 0, 0, 0, 0, 0, 0]] * 5)
    assert np.array_equal(p.between_v1, q.between_v1) and np.array_equal(p.between_z, q.between_z)
    assert p.n_prior == 1 and list(p.noise_dim) == [3, 3]
    from oracle import gtsam_oracle as O
    from gtsam_amd.params import LevenbergMarquardtParams as LMP
    r = O.lm_optimize(p, v0, LMP())
    final = r["values"].reshape(5, 3)
    # the example's known answer (Pose2SLAMExample.cpp output): a square with side 2
    assert np.allclose(final[:, :2], [[0, 0], [2, 0], [4, 0], [4, 2], [2, 2]], atol=1e-3)
