"""Known-answer tests of the reference's own unit tests for the geometry, calibration and optimizer pieces of the hot path
(SURVEY.md section 8(c)), applied to the three restatements this repository holds: the numpy oracle
(oracle/gtsam_oracle.py), the device formulas compiled for the host (tests/hostmath, the same headers the HIP kernels
include) and the host mirrors (gtsam_amd/io.py, gtsam_amd/api.py).  Literals are the reference's, file:line cited."""
import ctypes as C
import os

import numpy as np
import pytest

from gtsam_amd import api, io
from gtsam_amd.params import LevenbergMarquardtParams as LMP
from gtsam_amd.problem import VAR_POSE3
from oracle import gtsam_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


@pytest.fixture(scope="module")
def hm():
    path = os.path.join(ROOT, "tests", "_build", "libhostmath.so")
    if not os.path.exists(path):
        pytest.skip("tests/_build/libhostmath.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    return C.CDLL(path)


def _hm_log(hm, R):
    R = np.ascontiguousarray(np.asarray(R, np.float64).reshape(-1, 9)); w = np.zeros((R.shape[0], 3))
    hm.hm_so3_logmap(C.c_long(R.shape[0]), P(R), P(w)); return w


def _hm_exp(hm, w):
    w = np.ascontiguousarray(np.asarray(w, np.float64).reshape(-1, 3)); R = np.zeros((w.shape[0], 9))
    hm.hm_so3_expmap(C.c_long(w.shape[0]), P(w), P(R)); return R.reshape(-1, 3, 3)


def _both(hm):
    return [("oracle", O.so3_expmap, lambda R: O.so3_logmap(R)), ("device formulas", lambda w: _hm_exp(hm, w), lambda R: _hm_log(hm, R))]


def test_rot3_log_suite(hm):
    """gtsam/geometry/tests/testRot3.cpp:195-262 (TEST(Rot3, log)): Logmap(Rodrigues(w)) == w to 1e-12 for zero, tiny
    (Taylor), normal and 180-degree rotations about the axes; -w for the 180-degree rotation about (1,4,2)/sqrt(21);
    zero for 360-degree rotations; the not-quite-orthogonal Lund matrix to 1e-8."""
    n = np.sqrt(21.0); x, y, z = 1 / n, 4 / n, 2 / n
    same = [(0, 0, 0)]
    for d in (0.0001, 0.1):
        same += [(d, 0, 0), (0, d, 0), (0, 0, d), (x * d, y * d, z * d)]
    same += [(np.pi, 0, 0), (0, np.pi, 0), (0, 0, np.pi)]
    lund = np.array([-0.98582676, -0.03958746, -0.16303092, -0.03997006, -0.88835923, 0.45740671, -0.16293753, 0.45743998, 0.87418537])
    for name, expm, logm in _both(hm):
        for w in same:
            assert np.abs(logm(expm(np.array([w], float)))[0] - w).max() <= 1e-12, (name, w)
        w = np.array([[x * np.pi, y * np.pi, z * np.pi]])
        assert np.abs(logm(expm(w))[0] + w[0]).max() <= 1e-12, name                       # sign flipped: Vector(-w)
        for w in [(2 * np.pi, 0, 0), (0, 2 * np.pi, 0), (0, 0, 2 * np.pi), (x * 2 * np.pi, y * 2 * np.pi, z * 2 * np.pi)]:
            assert np.abs(logm(expm(np.array([w], float)))[0]).max() <= 1e-9, (name, w)   # assert_equal default tolerance
        assert np.abs(logm(lund.reshape(1, 3, 3))[0] - [0.264452, -0.742197708, -3.04098184]).max() <= 1e-8, name


def test_rot3_expmap_logmap_stability(hm):
    """testRot3.cpp:534-561: Expmap of a 1e-6 rotation against the 7th-order series (1e-10); Logmap(Expmap((1e-8,0,0)))
    to 1e-15."""
    w = np.array([78e-9, 5e-8, 97e-7]); t2 = float(w @ w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    series = np.eye(3) + (1 - t2 / 6 + t2 * t2 / 120 - t2 ** 3 / 5040) * W + (0.5 - t2 / 24 + t2 * t2 / 720) * (W @ W)
    for name, expm, logm in _both(hm):
        assert np.abs(expm(w[None])[0] - series).max() <= 1e-10, name
        assert np.abs(logm(expm(np.array([[1e-8, 0, 0]])))[0] - [1e-8, 0, 0]).max() <= 1e-15, name


def test_rot3_quaternion_literals():
    """testRot3.cpp:564-589 (TEST(Rot3, quaternion)): the g2o reader's quaternion -> rotation and the g2o writer's rotation
    -> quaternion on the reference's two literal pairs (1e-9)."""
    q1 = (0.710997408193224, 0.360544029310185, 0.594459869568306, 0.105395217842782)      # w x y z
    R1 = np.array([0.271018623057411, 0.278786459830371, 0.921318086098018, 0.578529366719085, 0.717799701969298,
                   -0.387385285854279, -0.769319620053772, 0.637998195662053, 0.033250932803219]).reshape(3, 3)
    q2 = (0.263360579192421, 0.571813128030932, 0.494678363680335, 0.599136268678053)
    R2 = np.array([-0.207341903877828, 0.250149415542075, 0.945745528564780, 0.881304914479026, -0.371869043667957,
                   0.291573424846290, 0.424630407073532, 0.893945571198514, -0.143353873763946]).reshape(3, 3)
    for (w, x, y, z), R in ((q1, R1), (q2, R2)):
        assert np.abs(io._quat(x, y, z, w) - R).max() <= 1e-9
        assert np.abs(io._quaternion(R) - [x, y, z, w]).max() <= 1e-9


def test_pose3_expmap_literals(hm):
    """gtsam/geometry/tests/testPose3.cpp:82-134: expmap_a_full (Pose3(Rodrigues(0.3,0,0), (0.2,0.7,-2)) from
    xi = (0.3,0,0, 0.2,0.394742,-2.08998), 1e-5), expmap_b (retract of a far-away pose, 1e-2), the planar screw
    (expmap_c_full, 1e-6).  Pose3 retract is Expmap with the default flags, which is what the device retracts with."""
    Rx = O.so3_expmap(np.array([[0.3, 0, 0]]))[0]
    xi = np.array([[0.3, 0, 0, 0.2, 0.394742, -2.08998]])
    a, c, s = 0.3, np.cos(0.3), np.sin(0.3)
    screw = np.array([[0, 0, a, a, 0, 1.0]])
    Rs = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]); ts = np.array([0.29552, 0.0446635, 1])
    R, t = O.pose3_expmap(xi)
    assert np.abs(R[0] - Rx).max() <= 1e-9 and np.abs(t[0] - [0.2, 0.7, -2]).max() <= 1e-5
    R, t = O.pose3_expmap(screw)
    assert np.abs(R[0] - Rs).max() <= 1e-6 and np.abs(t[0] - ts).max() <= 1e-6
    assert np.abs(O.pose3_logmap(Rs[None], ts[None])[0] - screw[0]).max() <= 1e-6
    # the same through the device's retract from the identity / from Pose3(I, (100,0,0))
    ident = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])[None]
    for x0, d, Rexp, texp, tol in ((ident, xi, Rx, [0.2, 0.7, -2], 1e-5), (ident, screw, Rs, ts, 1e-6),
                                   (np.concatenate([np.eye(3).reshape(-1), [100.0, 0, 0]])[None], np.array([[0, 0, 0.1, 0, 0, 0.0]]),
                                    O.so3_expmap(np.array([[0, 0, 0.1]]))[0], [100.0, 0, 0], 1e-2)):
        y = np.zeros((1, 12)); x0 = np.ascontiguousarray(x0); d = np.ascontiguousarray(d)
        hm.hm_retract(C.c_int(VAR_POSE3), C.c_long(1), P(x0), P(d), P(y))
        assert np.abs(y[0, :9].reshape(3, 3) - Rexp).max() <= tol and np.abs(y[0, 9:] - texp).max() <= tol
        assert np.abs(O.pose_retract(x0, d)[0] - y[0]).max() <= 1e-14


def test_calibration_uncalibrate_literals():
    """gtsam/geometry/tests/testCal3Bundler.cpp:28-47 (K(500, 1e-3, 1e-3, 1000, 2000), p = (2, 3) -> (u0 + g x, v0 + g y),
    g = f (1 + k1 r + k2 r^2)) and testCal3_S2.cpp:28-49 (K(500, 500, 0.1, 320, 240), (2, 3) -> (1320.3, 1740)), through the
    projection of a point at depth 1 by a camera at the origin."""
    cam = np.concatenate([np.eye(3).reshape(-1), np.zeros(3), [500, 1e-3, 1e-3, 1000, 2000]])
    r = 2 * 2 + 3 * 3; g = 500 * (1 + 1e-3 * r + 1e-3 * r * r)
    pi, _, _, behind = O.sfm_project(cam, np.array([[2.0, 3.0, 1.0]]))
    assert not behind[0] and np.abs(pi[0] - [1000 + g * 2, 2000 + g * 3]).max() <= 1e-9
    pi, _, _, _ = O.s2_project(np.eye(3)[None], np.zeros((1, 3)), np.array([[500.0, 500, 0.1, 320, 240]]), np.array([[2.0, 3.0, 1.0]]))
    assert np.abs(pi[0] - [1320.3, 1740]).max() <= 1e-9


def _more_optimization(init):
    nm = api.noiseModel
    fg = api.NonlinearFactorGraph()
    fg.addPriorPose2(0, api.Pose2(0, 0, 0), nm.Isotropic.Sigma(3, 1))
    fg.add(api.BetweenFactorPose2(0, 1, api.Pose2(1, 0, np.pi / 2), nm.Isotropic.Sigma(3, 1)))
    fg.add(api.BetweenFactorPose2(1, 2, api.Pose2(1, 0, np.pi / 2), nm.Isotropic.Sigma(3, 1)))
    vals = api.Values()
    for k, p in enumerate(init):
        vals.insert(k, api.Pose2(*p))
    return api.extract(fg, vals)


def test_nonlinear_optimizer_more_optimization_literals():
    """tests/testNonlinearOptimizer.cpp:248-321 (MoreOptimization): legacy LM from a far-off start reaches the literal
    poses (0,0,0), (1,0,pi/2), (1,1,pi) (assert_equal's 1e-9) with a zero gradient; with diagonal damping the damped
    system's Hessian diagonal is d + lambda d (:296-308)."""
    p, v0, _ = _more_optimization([(3, 4, -np.pi), (10, 2, -np.pi), (11, 7, -np.pi)])
    r = O.lm_optimize(p, v0, LMP.LegacyDefaults())
    final = r["values"].reshape(3, 3)
    expected = np.array([[0, 0, 0], [1, 0, np.pi / 2], [1, 1, np.pi]])
    diff = final - expected
    diff[:, 2] = np.arctan2(np.sin(diff[:, 2]), np.cos(diff[:, 2]))       # Pose2 equality is on the rotation, pi == -pi
    assert np.abs(diff).max() <= 1e-9
    _, g, _ = O.hessian_dense(p, r["values"])
    assert np.abs(g).max() <= 1e-9                                          # linear->gradientAtZero() == 0
    # the diagonal-damping identity at the better start of the test
    p, v1, _ = _more_optimization([(3, 4, 0), (10, 2, np.pi / 3), (11, 7, np.pi / 2)])
    params = LMP.LegacyDefaults()
    d = O.hessian_diagonal(p, v1)
    st, delta, H, g, lin = O.solve_damped(p, v1, params.lambdaInitial, True, 0.0, 1e300)
    assert st == 0 and np.allclose(np.diag(H), d, rtol=1e-13)
    damped = H + np.diag(params.lambdaInitial * d)                        # Hessian diagonal d + lambda d, off-diagonals untouched
    assert np.abs(damped @ delta - g).max() <= 1e-9 * np.abs(g).max()      # the step solves exactly that system
    assert np.abs(H @ delta - g).max() > 1e-7 * np.abs(g).max()            # ... and not the undamped one


def test_noise_model_constructors_literals(hm):
    """gtsam/linear/tests/testNoiseModel.cpp:40-103 (TEST(NoiseModel, constructors)): nine ways to build sigma = 2 in 3-D --
    SqrtInformation(I/2), Covariance(4 I), Information(I/4), Diagonal Sigmas/Variances/Precisions, Isotropic
    Sigma/Variance/Precision, all with smart = false -- whiten (10, 20, 30) to (5, 10, 15), squared Mahalanobis distance
    5^2 + 10^2 + 15^2; :106-111 Unit leaves the vector alone.  Checked on the mirror's factories -> the noise table, the
    oracle's whitening, and the device's prior error 1/2 ||whiten(x - z)||^2."""
    from gtsam_amd.problem import NOISE_DIAGONAL, NOISE_GAUSSIAN, NOISE_ISOTROPIC, Problem, VAR_POINT3
    nm = api.noiseModel
    I3 = np.eye(3)
    models = [nm.Gaussian.SqrtInformation(I3 * 0.5, False), nm.Gaussian.Covariance(I3 * 4.0, False), nm.Gaussian.Information(I3 * 0.25, False),
              nm.Diagonal.Sigmas([2.0, 2.0, 2.0], False), nm.Diagonal.Variances([4.0, 4.0, 4.0], False), nm.Diagonal.Precisions([0.25] * 3, False),
              nm.Isotropic.Sigma(3, 2.0, False), nm.Isotropic.Variance(3, 4.0, False)]
    assert [m.kind for m in models] == [NOISE_GAUSSIAN] * 3 + [NOISE_DIAGONAL] * 3 + [NOISE_ISOTROPIC] * 2   # not down-cast
    unwhitened = np.array([10.0, 20.0, 30.0]); whitened = np.array([5.0, 10.0, 15.0])
    for m in models + [nm.Unit.Create(3)]:
        p = Problem(var_type=np.array([VAR_POINT3], np.int32)); ni = p.add_noise(m.kind, 3, m.params)
        W = O.noise_sqrt_info(p, ni)
        want = unwhitened if m.kind == 0 else whitened
        assert np.abs(W @ unwhitened - want).max() <= 1e-12
        # device: PriorFactor<Point3> error with x - z = unwhitened; noise rows as gtg_upload_problem derives them
        # (1/sigma, 1/sigmas, R) -- api.hip "noise table"
        dev = {0: [0.0], 1: [1.0 / m.params[0]] if m.kind == 1 else [0.0], 2: list(1.0 / np.asarray(m.params)) if m.kind == 2 else [0.0],
               3: list(m.params)}[m.kind]
        dev = np.array(dev, float); x = unwhitened.copy(); z = np.zeros(3)
        hm.hm_prior_error.restype = C.c_double
        e = hm.hm_prior_error(C.c_int(VAR_POINT3), P(x), P(z), C.c_int(m.kind), P(dev))
        assert abs(e - 0.5 * float(want @ want)) <= 1e-10 and (m.kind == 0 or abs(2 * e - (5 * 5 + 10 * 10 + 15 * 15)) <= 1e-9)
    # smart constructors down-cast (NoiseModel.cpp:97-110, 283-308, 624-633)
    assert nm.Gaussian.Information(I3 * 0.25).kind == NOISE_ISOTROPIC and nm.Gaussian.Information(I3).kind == 0
    assert nm.Isotropic.Variance(3, 1.0).kind == 0


def test_cholesky_underconstrained_and_bad_scaling_literals(live_ref):
    """gtsam/base/tests/testCholesky.cpp:70-81 (BadScalingCholesky: diag(1e-80, 1) factors with R00 / R11 = 1e-40) and :101-139
    (underconstrained: L D L^T with a 1e-12 pivot, with zero pivots, with negative pivots -- choleskyPartial must report
    failure for all three: the rank test of base/cholesky.cpp:144-157 and Eigen's LLT info).  On the oracle's restatement
    and, where it is built, on the reference's own choleskyPartial."""
    A = np.diag([1e-40, 1.0]); A = A.T @ A
    ok, R = O.cholesky_partial(A, 2)
    assert ok and abs(R[0, 0] / R[1, 1] - 1e-40) <= 1e-41
    L = np.array([[1, 0, 0, 0, 0, 0],
                  [1.11177808157954, 1.06204809504665, 0.507342638873381, 1.34953401829486, 1, 0],
                  [0.155864888199928, 1.10933048588373, 0.501255576961674, 1, 0, 0],
                  [1.12108665967793, 1.01584408366945, 1, 0, 0, 0],
                  [0.776164062474843, 0.117617236580373, -0.0236628691347294, 0.814118199972143, 0.694309975328922, 1],
                  [0.1197220685104, 1, 0, 0, 0, 0]])
    d = [0.814723686393179, 0.811780089277421, 1.82596950680844, 0.240287537694585]
    for tail in ([1.34342584865901, 1e-12], [0.0, 0.0], [-0.5, -0.6]):
        M = L @ np.diag(d + tail) @ L.T
        assert not O.cholesky_partial(M, 6)[0], tail
        if live_ref is not None:
            assert not live_ref.cholesky_partial(M, 6)[0], tail
    if live_ref is not None:
        ok, R = live_ref.cholesky_partial(A, 2)
        assert ok and abs(R[0, 0] / R[1, 1] - 1e-40) <= 1e-41


def test_general_sfm_factor_cal3bundler_error_literal(hm):
    """gtsam/slam/tests/testGeneralSFMFactor_Cal3Bundler.cpp:96-110 (TEST(GeneralSFMFactor_Cal3Bundler, error)): measurement
    (3, 0), default Cal3Bundler camera at (0, 0, -6) looking at the origin -> unwhitenedError = (-3, 0).  The linearized
    factor carries b = -error (GeneralSFMFactor.h:150-152)."""
    cam = np.concatenate([np.eye(3).reshape(-1), [0, 0, -6.0], [1.0, 0, 0, 0, 0]])
    pi, Dcam, Dpoint, behind = O.sfm_project(cam, np.zeros((1, 3)))
    assert not behind[0] and np.abs(pi[0] - np.array([3.0, 0.0]) - [-3.0, 0.0]).max() <= 1e-15
    J = np.zeros((1, 26)); z = np.array([[3.0, 0.0]]); pt = np.zeros((1, 3)); camc = np.ascontiguousarray(cam[None]); nd = np.zeros(1)
    hm.hm_sfm_linearize(C.c_long(1), P(camc), P(pt), P(z), C.c_int(0), P(nd), P(J))
    assert np.abs(J[0, 24:] - [3.0, 0.0]).max() <= 1e-15                       # b = z - h(x) = -(unwhitened error)
    assert np.abs(J[0, :18].reshape(2, 9) - Dcam[0]).max() <= 1e-14 and np.abs(J[0, 18:24].reshape(2, 3) - Dpoint[0]).max() <= 1e-14


def test_pcg_solver_test_literals():
    """tests/testPCGSolver.cpp:42-74 (llt: R^T R factors back to R, back substitution gives (6.5, 2.5, 3)) and :78-120
    (GaussianFactorGraphSystem::multiply / getb on a 3-variable graph with Diagonal sigmas (0.5, 0.3): A^T b and A^T A p
    literals for the first CG direction p = b); then the oracle's preconditioned CG (block Jacobi) on that system reaches the
    direct solution."""
    R = np.array([[1.0, -1, -1], [0, 2, -1], [0, 0, 1]]); AtA = R.T @ R
    ok, M = O.cholesky_partial(np.block([[AtA, np.array([[1.0], [2.0], [3.0]])], [np.zeros((1, 3)), np.zeros((1, 1))]]), 3)
    assert ok and np.abs(np.triu(M[:3, :3]) - R).max() <= 1e-12
    assert np.abs(np.linalg.solve(np.triu(M[:3, :3]), np.array([1.0, 2, 3])) - [6.5, 2.5, 3.0]).max() <= 1e-12
    # the factor graph of the test: (keys, blocks, b), every row whitened by 1 / sigma
    I2 = np.eye(2); w = np.diag([1 / 0.5, 1 / 0.3])
    factors = [({2: 10 * I2}, [-1, -1]), ({2: -10 * I2, 0: 10 * I2}, [2, -1]), ({2: -5 * I2, 1: 5 * I2}, [0, 1]),
               ({0: -5 * I2, 1: 5 * I2}, [-1, 1.5]), ({0: I2}, [0, 0]), ({1: I2}, [0, 0]), ({2: I2}, [0, 0])]
    A = np.zeros((2 * len(factors), 6)); b = np.zeros(2 * len(factors))
    for i, (blocks, rhs) in enumerate(factors):
        for key, blk in blocks.items():
            A[2 * i:2 * i + 2, 2 * key:2 * key + 2] = w @ blk
        b[2 * i:2 * i + 2] = w @ np.array(rhs, float)
    Atb = A.T @ b; AtA = A.T @ A
    assert np.abs(Atb - [100.0, -194.444, -20.0, 138.889, -120.0, -55.556]).max() <= 1e-3
    assert np.abs(AtA @ Atb - [100400, -249074.074, -2080, 148148.148, -146480, 37962.963]).max() <= 1e-3
    x, its, g0, g1 = O.preconditioned_conjugate_gradient(AtA, Atb, [(0, 2), (2, 4), (4, 6)], 500, 1, 1e-14, 1e-28)
    assert np.abs(x - np.linalg.solve(AtA, Atb)).max() <= 1e-10 and 1 <= its <= 6
