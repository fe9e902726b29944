"""The per-factor DEVICE formulas (gtsam_amd/csrc/geom.h, factors.h) compiled for the host with plain g++
(tests/hostmath/hostmath.cpp, test infrastructure) against the oracle: catches formula slips without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from gtsam_amd.problem import Problem
from oracle import gtsam_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hm():
    so = os.path.join(ROOT, "tests", "_build", "libhostmath.so")
    src = os.path.join(ROOT, "tests", "hostmath", "hostmath.cpp")
    hdrs = [os.path.join(ROOT, "gtsam_amd", "csrc", h) for h in ("geom.h", "factors.h")]
    if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in [src] + hdrs):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src], check=True)
    return C.CDLL(so)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _cams(rng, n):
    xi = rng.normal(size=(n, 6)); xi[:, :3] *= 1.2
    R, t = O.pose3_expmap(xi)
    cam = np.concatenate([O.pose_pack(R, t), np.stack([rng.uniform(400, 900, n), rng.normal(0, 1e-2, n), rng.normal(0, 1e-3, n),
                                                       rng.normal(0, 3, n), rng.normal(0, 3, n)], 1)], 1)
    pc = np.stack([rng.normal(0, 1, n), rng.normal(0, 1, n), rng.uniform(-0.5, 8, n)], 1)   # some behind the camera
    return cam, np.einsum("nij,nj->ni", R, pc) + t


@pytest.mark.parametrize("nk,nd", [(0, []), (1, [0.7]), (2, [0.5, 2.0]), (3, [1.2, 0.3, 0.0, 0.8])])
def test_sfm_factor(hm, nk, nd):
    rng = np.random.default_rng(3); n = 500
    cam, pw = _cams(rng, n); z = rng.normal(0, 100, (n, 2))
    p = Problem(var_type=np.array([1, 2], np.int32)); ni = p.add_noise(nk, 2, nd)
    W = O.noise_sqrt_info(p, ni)
    dev_nd = np.array({0: [0.0], 1: [1.0 / nd[0]] if nk == 1 else [0], 2: list(1.0 / np.array(nd)) if nk == 2 else [0], 3: nd}[nk], float)
    J = np.zeros((n, 26)); hm.hm_sfm_linearize(C.c_long(n), P(cam), P(pw), P(z), C.c_int(nk), P(dev_nd), P(J))
    pi, Dc, Dp, behind = O.sfm_project(cam, pw); b = z - pi; Dc[behind] = 0; Dp[behind] = 0; b[behind] = 0
    assert behind.sum() > 0
    ref = np.concatenate([np.einsum("ij,njk->nik", W, Dc).reshape(n, -1), np.einsum("ij,njk->nik", W, Dp).reshape(n, -1),
                          np.einsum("ij,nj->ni", W, b)], 1)
    assert rel(J, ref) <= 1e-13
    e = np.zeros(n); hm.hm_sfm_error(C.c_long(n), P(cam), P(pw), P(z), C.c_int(nk), P(dev_nd), P(e))
    assert rel(e, 0.5 * (ref[:, 24:] ** 2).sum(1)) <= 1e-13


def test_between_prior_retract(hm):
    rng = np.random.default_rng(4); n = 300
    from gtsam_amd.problem import pose_graph_problem
    R1, t1 = O.pose3_expmap(rng.normal(size=(n, 6)) * [1.5, 1.5, 1.5, 1, 1, 1]); R2, t2 = O.pose3_expmap(rng.normal(size=(n, 6)))
    Rz, tz = O.pose3_expmap(rng.normal(size=(n, 6)) * [1.5, 1.5, 1.5, 1, 1, 1])
    T1, T2, Z = O.pose_pack(R1, t1), O.pose_pack(R2, t2), O.pose_pack(Rz, tz)
    A = rng.normal(size=(6, 6)); Rn = np.ascontiguousarray(np.linalg.cholesky(A @ A.T + 6 * np.eye(6)).T)
    J = np.zeros((n, 78)); hm.hm_between_linearize(C.c_long(n), P(T1), P(T2), P(Z), C.c_int(3), P(Rn.reshape(-1)), P(J))
    p = pose_graph_problem(2 * n, np.arange(n), np.arange(n) + n, Z, [3] * n, np.tile(Rn.reshape(-1), (n, 1)))
    vals = np.concatenate([T1.reshape(-1), T2.reshape(-1)])
    assert rel(J, O.jacobians_flat(p, vals, 2)) <= 1e-11
    e = np.zeros(n); hm.hm_between_error(C.c_long(n), P(T1), P(T2), P(Z), C.c_int(3), P(Rn.reshape(-1)), P(e))
    assert rel(e, 0.5 * (O.jacobians_flat(p, vals, 2)[:, 72:] ** 2).sum(1)) <= 1e-11
    # retract / local for all three value types
    cam = np.concatenate([T1, rng.normal(size=(n, 5))], 1)
    for vt, x, dm in ((0, T1, 6), (1, cam, 9), (2, rng.normal(size=(n, 3)), 3)):
        d = np.ascontiguousarray(rng.normal(size=(n, dm)) * 0.4)
        x = np.ascontiguousarray(x); y = np.zeros_like(x)
        hm.hm_retract(C.c_int(vt), C.c_long(n), P(x), P(d), P(y))
        pp = Problem(var_type=np.full(n, vt, np.int32))
        assert rel(y, O.retract(pp, x.reshape(-1), d.reshape(-1)).reshape(x.shape)) <= 1e-14
        back = np.zeros((n, dm)); hm.hm_local(C.c_int(vt), C.c_long(n), P(x), P(y), P(back))
        assert rel(back, d) <= 1e-10


def test_logmap_branches(hm):
    """SO3::Logmap's three regimes (near pi by largest diagonal, acos, Taylor) against the oracle restatement."""
    rng = np.random.default_rng(5)
    ws = []
    for ang in [np.pi - 1e-5, np.pi - 5e-4, np.pi - 2e-2, 3.0, 1.0, 1e-3, 1e-5, 1e-9, 0.0]:
        for _ in range(6):
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ws.append(ax * ang)
    R = np.ascontiguousarray(O.so3_expmap(np.array(ws)))
    w = np.zeros((R.shape[0], 3)); hm.hm_so3_logmap(C.c_long(R.shape[0]), P(R), P(w))
    assert np.abs(w - O.so3_logmap(R)).max() <= 1e-9
    R2 = np.zeros_like(R); wa = np.ascontiguousarray(np.array(ws)); hm.hm_so3_expmap(C.c_long(R.shape[0]), P(wa), P(R2))
    assert np.abs(R2 - R).max() <= 1e-15


def test_m_estimators_host_compile_vs_oracle(hm):
    """robust_weight / robust_loss of csrc/factors.h (the device code, compiled for the host) against the oracle's
    restatement of linear/LossFunctions.cpp, across the threshold of every estimator."""
    hm.hm_robust_weight.restype = C.c_double; hm.hm_robust_loss.restype = C.c_double
    d = np.concatenate([np.linspace(0, 12, 97), [1e-12, 1.345, 4.6851, 1e3]])
    for rk in range(1, 7):
        for k in (0.5, 1.345, 4.6851):
            w = np.array([hm.hm_robust_weight(C.c_int(rk), C.c_double(k), C.c_double(x)) for x in d])
            l = np.array([hm.hm_robust_loss(C.c_int(rk), C.c_double(k), C.c_double(x)) for x in d])
            assert np.allclose(w, O.robust_weight(rk, k, d), rtol=1e-14, atol=0)
            assert np.allclose(l, O.robust_loss(rk, k, d), rtol=1e-13, atol=1e-300)
            # loss'(d) = d * weight(d)   (the defining relation, LossFunctions.h:33-60)
            h = 1e-6; dd = d[(d > 0.01) & (np.abs(d - k) > 0.01)]
            num = (O.robust_loss(rk, k, dd + h) - O.robust_loss(rk, k, dd - h)) / (2 * h)
            assert np.allclose(num, dd * O.robust_weight(rk, k, dd), rtol=1e-6, atol=1e-8)


# literals of linear/tests/testNoiseModel.cpp:460-569 (robustFunctionFair/Huber/Cauchy/GemanMcClure/Welsch/Tukey):
# (kind, k, [(error, weight, loss)...])
M_ESTIMATOR_LITERALS = [
    (1, 5.0, [(1.0, 0.8333333333333333, 0.441961080151135), (10.0, 0.3333333333333333, 22.534692783297260),
              (-10.0, 0.3333333333333333, 22.534692783297260), (-1.0, 0.8333333333333333, 0.441961080151135)]),
    (2, 5.0, [(1.0, 1.0, 0.5), (10.0, 0.5, 37.5), (-10.0, 0.5, 37.5), (-1.0, 1.0, 0.5)]),
    (3, 5.0, [(1.0, 0.961538461538461, 0.490258914416017), (10.0, 0.2, 20.117973905426254),
              (-10.0, 0.2, 20.117973905426254), (-1.0, 0.961538461538461, 0.490258914416017)]),
    (6, 1.0, [(1.0, 0.25, 0.25), (10.0, 9.80296e-5, 0.495049504950495), (-10.0, 9.80296e-5, 0.495049504950495),
              (-1.0, 0.25, 0.25)]),
    (5, 5.0, [(1.0, 0.960789439152323, 0.490132010595960), (10.0, 0.018315638888734, 12.271054513890823),
              (-10.0, 0.018315638888734, 12.271054513890823), (-1.0, 0.960789439152323, 0.490132010595960)]),
    (4, 5.0, [(1.0, 0.9216, 0.480266666666667), (10.0, 0.0, 4.166666666666667), (-10.0, 0.0, 4.166666666666667),
              (-1.0, 0.9216, 0.480266666666667)]),
]


def test_m_estimators_reference_literals(hm):
    hm.hm_robust_weight.restype = C.c_double; hm.hm_robust_loss.restype = C.c_double
    for rk, k, rows in M_ESTIMATOR_LITERALS:
        for e, w, l in rows:
            assert abs(hm.hm_robust_weight(C.c_int(rk), C.c_double(k), C.c_double(e)) - w) < 1e-8
            assert abs(hm.hm_robust_loss(C.c_int(rk), C.c_double(k), C.c_double(e)) - l) < 1e-8
            assert abs(float(O.robust_weight(rk, k, e)) - w) < 1e-8 and abs(float(O.robust_loss(rk, k, e)) - l) < 1e-8


def test_pose2_between_prior_retract(hm):
    """BetweenFactor<Pose2> / PriorFactor<Pose2> / Pose2 retract+local of the device code against the oracle (which is
    pinned to the live reference: identical LM trajectories on w100.graph and noisyToyGraph.txt)."""
    from gtsam_amd.problem import (NOISE_DIAGONAL, VAR_POSE2, pose2_graph_problem)
    rng = np.random.default_rng(5); n = 300
    poses = np.stack([rng.normal(0, 5, n), rng.normal(0, 5, n), rng.uniform(-np.pi, np.pi, n)], 1)
    v1 = rng.integers(0, n, 400); v2 = (v1 + rng.integers(1, n, 400)) % n
    z = np.stack([rng.normal(0, 2, 400), rng.normal(0, 2, 400), rng.uniform(-np.pi, np.pi, 400)], 1)
    sig = np.array([0.3, 0.5, 0.1])
    p = pose2_graph_problem(n, v1, v2, z, np.full(400, NOISE_DIAGONAL), np.tile(np.concatenate([sig, np.zeros(6)]), (400, 1)))
    v = poses.reshape(-1)
    Jo = O.jacobians_flat(p, v, 2)
    a = np.ascontiguousarray(poses[v1]); b = np.ascontiguousarray(poses[v2]); zc = np.ascontiguousarray(z)
    inv = np.ascontiguousarray(1.0 / sig)
    J = np.zeros((400, 78)); hm.hm_between2_linearize(C.c_long(400), P(a), P(b), P(zc), C.c_int(2), P(inv), P(J))
    assert rel(J, Jo) <= 1e-13
    e = np.zeros(400); hm.hm_between2_error(C.c_long(400), P(a), P(b), P(zc), C.c_int(2), P(inv), P(e))
    assert abs(e.sum() - O.error(p, v)) <= 1e-12 * e.sum()
    d = rng.normal(0, 0.4, (n, 3)); d[:, 2] = rng.uniform(-3.5, 3.5, n)
    y = np.zeros((n, 3)); hm.hm_retract(C.c_int(3), C.c_long(n), P(poses), P(np.ascontiguousarray(d)), P(y))
    assert rel(y, O.pose2_retract(poses, d)) <= 1e-14
    back = np.zeros((n, 3)); hm.hm_local(C.c_int(3), C.c_long(n), P(poses), P(y), P(back))
    assert rel(back, O.pose2_local(poses, y)) <= 1e-13
    # local(retract(x, d)) = d up to the wrap of theta: Pose2's chart is first order, exact for this composition
    assert np.allclose(np.angle(np.exp(1j * (back[:, 2] - d[:, 2]))), 0, atol=1e-12) and np.allclose(back[:, :2], d[:, :2], atol=1e-12)


def _numerical_columns(f_b, x, vtype, dim, hm, eps=1e-6):
    """-d b / d delta of a record's right-hand side b(x (+) delta) by central differences through the device's own retract."""
    cols = []
    for i in range(dim):
        d = np.zeros((1, dim)); d[0, i] = eps
        xp = np.zeros_like(x); xm = np.zeros_like(x)
        hm.hm_retract(C.c_int(vtype), C.c_long(1), P(x), P(d), P(xp))
        d[0, i] = -eps
        hm.hm_retract(C.c_int(vtype), C.c_long(1), P(x), P(d), P(xm))
        cols.append(-(f_b(xp) - f_b(xm)) / (2 * eps))
    return np.stack(cols, 1)


def test_cal3ds2_uncalibrate_literal(hm):
    """geometry/tests/testCal3DS2.cpp:28-45 (TEST(Cal3DS2, Uncalibrate)): K(500, 100, 0.1, 320, 240, 1e-3, 2e-3, 3e-3, 4e-3),
    intrinsic point (2, 3) -> K * [g x + tx, g y + ty, 1]; here through the device's projection with the camera at the origin
    looking at (2, 3, 1), i.e. the same intrinsic point; and a zero-distortion entry is the plain Cal3_S2."""
    K = np.array([500, 100, 0.1, 320, 240, 1e-3, 2e-3, 3e-3, 4e-3], np.float64)
    x, y = 2.0, 3.0
    r = x * x + y * y; g = 1 + K[5] * r + K[6] * r * r
    tx = 2 * K[7] * x * y + K[8] * (r + 2 * x * x); ty = K[7] * (r + 2 * y * y) + 2 * K[8] * x * y
    expect = np.array([500 * (g * x + tx) + 0.1 * (g * y + ty) + 320, 100 * (g * y + ty) + 240])
    pose = np.ascontiguousarray(np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])[None]); pw = np.array([[x, y, 1.0]]); z = np.zeros((1, 2)); unit = np.zeros(1)
    J = np.zeros((1, 20)); hm.hm_proj_linearize(C.c_long(1), P(pose), P(K), None, P(pw), P(z), C.c_int(0), P(unit), P(J))
    assert np.abs(-J[0, 18:] - expect).max() <= 1e-12 * np.abs(expect).max()     # b = -(h(x) - z), z = 0
    K2 = K.copy(); K2[5:] = 0
    J2 = np.zeros((1, 20)); hm.hm_proj_linearize(C.c_long(1), P(pose), P(K2), None, P(pw), P(z), C.c_int(0), P(unit), P(J2))
    assert np.allclose(-J2[0, 18:], [500 * x + 0.1 * y + 320, 100 * y + 240], rtol=1e-15)


def test_device_jacobians_against_numerical_derivatives(hm):
    """The reference's own way of testing a factor (EXPECT_CORRECT_FACTOR_JACOBIANS, numericalDerivative: SURVEY section 4),
    applied to the device formulas: central differences of the residual through the device's retract against the analytic
    blocks of the record.  GeneralSFMFactor and GenericProjectionFactor (with and without body_P_sensor) at random
    configurations in front of the camera; BetweenFactor<Pose3> / <Pose2> at zero error, where the reference's Jacobians
    (not multiplied by dLog, BetweenFactor.h:115-123) are exact -- the configuration testBetweenFactor.cpp:99-113 uses."""
    rng = np.random.default_rng(11)
    unit = np.zeros(1)
    for trial in range(20):
        # ---- GeneralSFMFactor<PinholeCamera<Cal3Bundler>, Point3>
        xi = rng.normal(size=(1, 6)) * [0.8, 0.8, 0.8, 1, 1, 1]
        R, t = O.pose3_expmap(xi)
        cam = np.ascontiguousarray(np.concatenate([O.pose_pack(R, t), [[rng.uniform(400, 900), rng.normal(0, 1e-2), rng.normal(0, 1e-3), 0, 0]]], 1))
        pc = np.array([rng.normal(0, 1), rng.normal(0, 1), rng.uniform(2, 8)])
        pw = np.ascontiguousarray((R[0] @ pc + t[0])[None]); z = np.ascontiguousarray(rng.normal(0, 50, (1, 2)))

        def sfm_b(c, p=pw):
            J = np.zeros((1, 26)); hm.hm_sfm_linearize(C.c_long(1), P(np.ascontiguousarray(c)), P(np.ascontiguousarray(p)), P(z), C.c_int(0), P(unit), P(J)); return J[0, 24:].copy()
        J = np.zeros((1, 26)); hm.hm_sfm_linearize(C.c_long(1), P(cam), P(pw), P(z), C.c_int(0), P(unit), P(J))
        A1, A2 = J[0, :18].reshape(2, 9), J[0, 18:24].reshape(2, 3)
        N1 = _numerical_columns(sfm_b, cam, 1, 9, hm)
        N2 = _numerical_columns(lambda p: sfm_b(cam, p), pw, 2, 3, hm)
        assert np.abs(N1 - A1).max() <= 1e-6 * max(1.0, np.abs(A1).max()) and np.abs(N2 - A2).max() <= 1e-6 * max(1.0, np.abs(A2).max())
        # ---- GenericProjectionFactor<Pose3, Point3, Cal3_S2>, optional body_P_sensor
        # (the calibration entry is 9 doubles: fx fy s u0 v0 + k1 k2 p1 p2 -- zero for a Cal3_S2, the second pass is a Cal3DS2)
        pose = np.ascontiguousarray(O.pose_pack(R, t)); K0 = np.array([rng.uniform(400, 900), rng.uniform(400, 900), rng.normal(0, 0.5), 320.0, 240.0, 0, 0, 0, 0])
        Kd = K0.copy(); Kd[5:] = [rng.normal(0, 5e-2), rng.normal(0, 1e-2), rng.normal(0, 5e-3), rng.normal(0, 5e-3)]
        Rs, ts = O.pose3_expmap(rng.normal(size=(1, 6)) * 0.1); sensor = np.ascontiguousarray(O.pose_pack(Rs, ts)[0])
        for K, sen in ((K0, None), (K0, sensor), (Kd, None), (Kd, sensor)):
            sp = P(sen) if sen is not None else None
            pw2 = pw if sen is None else np.ascontiguousarray((R[0] @ (Rs[0] @ pc + ts[0]) + t[0])[None])

            def proj_b(x, p=pw2):
                J = np.zeros((1, 20)); hm.hm_proj_linearize(C.c_long(1), P(np.ascontiguousarray(x)), P(K), sp, P(np.ascontiguousarray(p)), P(z), C.c_int(0), P(unit), P(J)); return J[0, 18:].copy()
            J = np.zeros((1, 20)); hm.hm_proj_linearize(C.c_long(1), P(pose), P(K), sp, P(pw2), P(z), C.c_int(0), P(unit), P(J))
            A1, A2 = J[0, :12].reshape(2, 6), J[0, 12:18].reshape(2, 3)
            N1 = _numerical_columns(proj_b, pose, 0, 6, hm)
            N2 = _numerical_columns(lambda p: proj_b(pose, p), pw2, 2, 3, hm)
            assert np.abs(N1 - A1).max() <= 1e-6 * max(1.0, np.abs(A1).max()) and np.abs(N2 - A2).max() <= 1e-6 * max(1.0, np.abs(A2).max())
        # ---- BetweenFactor<Pose3> at zero error
        R2, t2 = O.pose3_expmap(rng.normal(size=(1, 6))); T1 = pose; T2 = np.ascontiguousarray(O.pose_pack(R2, t2))
        Ri, ti = O.pose_inverse(R, t); Rz, tz = O.pose_compose(Ri, ti, R2, t2); Z = np.ascontiguousarray(O.pose_pack(Rz, tz))

        def btw_b(a, b):
            J = np.zeros((1, 78)); hm.hm_between_linearize(C.c_long(1), P(np.ascontiguousarray(a)), P(np.ascontiguousarray(b)), P(Z), C.c_int(0), P(unit), P(J)); return J[0, 72:].copy()
        J = np.zeros((1, 78)); hm.hm_between_linearize(C.c_long(1), P(T1), P(T2), P(Z), C.c_int(0), P(unit), P(J))
        assert np.abs(J[0, 72:]).max() <= 1e-12
        assert np.abs(_numerical_columns(lambda a: btw_b(a, T2), T1, 0, 6, hm) - J[0, :36].reshape(6, 6)).max() <= 1e-6 * max(1.0, np.abs(J[0, :36]).max())
        assert np.abs(_numerical_columns(lambda b: btw_b(T1, b), T2, 0, 6, hm) - J[0, 36:72].reshape(6, 6)).max() <= 1e-6
        # ---- BetweenFactor<Pose2> at zero error (3x3 blocks inside the 78-double record, rows of 6)
        a = np.ascontiguousarray(np.array([[rng.normal(0, 3), rng.normal(0, 3), rng.uniform(-3, 3)]])); b = np.ascontiguousarray(np.array([[rng.normal(0, 3), rng.normal(0, 3), rng.uniform(-3, 3)]]))
        ca, sa = np.cos(a[0, 2]), np.sin(a[0, 2]); dx, dy = b[0, 0] - a[0, 0], b[0, 1] - a[0, 1]
        zz = np.ascontiguousarray(np.array([[ca * dx + sa * dy, -sa * dx + ca * dy, b[0, 2] - a[0, 2]]]))       # a^-1 b

        def btw2(aa, bb):
            J = np.zeros((1, 78)); hm.hm_between2_linearize(C.c_long(1), P(np.ascontiguousarray(aa)), P(np.ascontiguousarray(bb)), P(zz), C.c_int(0), P(unit), P(J)); return J[0]
        J2 = btw2(a, b)                                   # H1 at [0:9], H2 at [36:45] (3x3 row-major), b at [72:75]
        assert np.abs(J2[72:75]).max() <= 1e-12
        N1 = _numerical_columns(lambda aa: btw2(aa, b)[72:75], a, 3, 3, hm)
        N2 = _numerical_columns(lambda bb: btw2(a, bb)[72:75], b, 3, 3, hm)
        assert np.abs(N1 - J2[0:9].reshape(3, 3)).max() <= 1e-6 * max(1.0, np.abs(J2[0:9]).max())
        assert np.abs(N2 - J2[36:45].reshape(3, 3)).max() <= 1e-6


def test_smart_triangulation_against_the_reference(hm, live_ref):
    """geom.h::smart_triangulate (what k_smart_triangulate runs per smart factor) against gtsam::triangulateSafe of the live
    reference: same status on well-posed, rank-deficient (two cameras at one place), behind-camera, far and outlier tracks, and
    the same point to 1e-9 (the reference takes the smallest right singular vector of the DLT matrix by JacobiSVD, the device
    the smallest eigenvector of A^T A by Jacobi rotations)."""
    if live_ref is None:
        pytest.skip("oracle/_ref not present")
    from gtsam_amd import datasets as D
    lib = live_ref.lib()
    lib.ref_triangulate_safe.restype = C.c_int
    hm.hm_smart_triangulate.restype = C.c_int
    cams, pts, oc, op, oz = D.synthetic_orbit_scene(n_cams=8, n_points=60, seed=11)
    rng = np.random.default_rng(2)
    seen = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0}
    for j in range(60):
        idx = np.flatnonzero(op == j)
        for variant in range(5):
            cj = np.ascontiguousarray(cams[oc[idx]]); zj = np.ascontiguousarray(oz[idx]).copy()
            rank_tol, dist, outl = 1.0, -1.0, -1.0
            if variant == 1:                                   # the same camera twice and nothing else: rank < 3
                cj = np.ascontiguousarray(np.stack([cj[0], cj[0]])); zj = np.ascontiguousarray(np.stack([zj[0], zj[0]]))
            elif variant == 2:                                 # two cameras with exchanged, exaggerated disparities: the rays meet behind them
                cj = np.ascontiguousarray(cj[[0, -1]]); d = zj[-1] - zj[0]
                zj = np.ascontiguousarray(np.stack([zj[0] + 3.0 * d, zj[-1] - 3.0 * d]))
            elif variant == 3:
                dist = 7.5 + 0.5 * rng.uniform()
            elif variant == 4:
                zj[0] += 40.0; outl = 15.0
            m = cj.shape[0]
            pr = np.zeros(3); pd = np.zeros(3)
            sr = lib.ref_triangulate_safe(C.c_int(m), P(cj), P(zj), C.c_double(rank_tol), C.c_double(dist), C.c_double(outl), P(pr))
            sd = hm.hm_smart_triangulate(C.c_int(m), P(cj), P(zj), C.c_double(rank_tol), C.c_double(dist), C.c_double(outl), P(pd))
            assert sr == sd, (j, variant, sr, sd)
            seen[sr] += 1
            if sr == 0:
                assert np.abs(pr - pd).max() <= 1e-9 * max(1.0, np.abs(pr).max()), (j, variant, pr, pd)
    assert all(v > 0 for v in seen.values()), seen              # every status occurred


def test_smart_triangulation_with_epi(hm, live_ref):
    """TriangulationParameters::enableEPI: geom.h::triangulate_refine -- the reference's LM on one TriangulationFactor per camera,
    restated decision by decision -- against the oracle's restatement (always) and the live reference's triangulateSafe (when
    present), on clean, rank-deficient, behind-camera, far, outlier and NOISY tracks (there the refinement moves the point by up
    to tens of units): the same status, the same point to 1e-9, and the same CheiralityException where the refinement linearises
    at a point behind a camera (status 6 on all three sides)."""
    from gtsam_amd import datasets as D
    hm.hm_smart_triangulate_epi.restype = C.c_int
    lib = None
    if live_ref is not None:
        lib = live_ref.lib(); lib.ref_triangulate_safe_epi.restype = C.c_int
    cams, pts, oc, op, oz = D.synthetic_orbit_scene(n_cams=8, n_points=60, seed=11)
    rng = np.random.default_rng(2)
    seen = {}; moved = 0.0
    for j in range(60):
        idx = np.flatnonzero(op == j)
        for variant in range(6):
            cj = np.ascontiguousarray(cams[oc[idx]]); zj = np.ascontiguousarray(oz[idx]).copy()
            rank_tol, dist, outl = 1.0, -1.0, -1.0
            if variant == 1:
                cj = np.ascontiguousarray(np.stack([cj[0], cj[0]])); zj = np.ascontiguousarray(np.stack([zj[0], zj[0]]))
            elif variant == 2:
                cj = np.ascontiguousarray(cj[[0, -1]]); d = zj[-1] - zj[0]
                zj = np.ascontiguousarray(np.stack([zj[0] + 3.0 * d, zj[-1] - 3.0 * d]))
            elif variant == 3:
                dist = 7.5 + 0.5 * rng.uniform()
            elif variant == 4:
                zj[0] += 40.0; outl = 15.0
            elif variant == 5:
                zj += rng.normal(0, 6.0, zj.shape)
            m = cj.shape[0]
            pd = np.zeros(3); p0 = np.zeros(3)
            sd = hm.hm_smart_triangulate_epi(C.c_int(m), P(cj), P(zj), C.c_double(rank_tol), C.c_double(dist), C.c_double(outl), C.c_int(1), P(pd))
            s0 = hm.hm_smart_triangulate_epi(C.c_int(m), P(cj), P(zj), C.c_double(rank_tol), C.c_double(dist), C.c_double(outl), C.c_int(0), P(p0))
            try:
                so, po = O.triangulate_safe(cj, zj, rank_tol, dist, outl, enable_epi=True)
            except RuntimeError as e:
                assert "Cheirality" in str(e); so, po = 6, None
            assert sd == so, (j, variant, sd, so)
            seen[sd] = seen.get(sd, 0) + 1
            if sd == 0:
                assert np.abs(pd - po).max() <= 1e-9 * max(1.0, np.abs(po).max()), (j, variant, pd, po)
                if s0 == 0:
                    moved = max(moved, float(np.abs(pd - p0).max()))
            if lib is not None:
                pr = np.zeros(3)
                sr = lib.ref_triangulate_safe_epi(C.c_int(m), P(cj), P(zj), C.c_double(rank_tol), C.c_double(dist), C.c_double(outl), C.c_int(1), P(pr))
                assert sr == sd, (j, variant, sr, sd)
                if sr == 0:
                    assert np.abs(pr - pd).max() <= 1e-9 * max(1.0, np.abs(pr).max()), (j, variant, pr, pd)
    assert seen.get(0, 0) > 100 and seen.get(1, 0) and seen.get(3, 0) and seen.get(4, 0) and seen.get(6, 0), seen
    assert moved > 1.0                                          # the refinement is not a no-op on these inputs


def test_smart_point_at_infinity(hm):
    """The records of a smart factor whose landmark is a point at infinity (geom.h sfm_backproject_at_infinity /
    sfm_project_at_infinity, factors.h sfm_linearize_at_infinity) against the oracle's restatement, which multiplies the chain of
    Jacobians the way the reference does (with both tangent bases) and is itself pinned on the reference's fixtures
    (tests/golden/smart_far_*.npz).  The device drops the basis of the camera-frame direction (it cancels) and keeps the world
    basis: the landmark block must agree entry by entry, the third (padding) column must be zero, and the derivative with
    respect to the translation must be zero."""
    rng = np.random.default_rng(11); n = 400
    cam, _ = _cams(rng, n)
    z = rng.normal(0, 120, (n, 2))
    dirs = np.zeros((n, 3))
    for i in range(n):
        assert hm.hm_sfm_backproject_at_infinity(P(cam[i]), P(z[i]), P(dirs[i])) == 1
        assert rel(dirs[i], O.backproject_point_at_infinity(cam[i], z[i])) <= 1e-14
    # look at each direction from OTHER cameras as well: a rotation of up to ~60 degrees keeps most of them in front
    cam2 = cam.copy()
    dR, _ = O.pose3_expmap(np.concatenate([rng.normal(0, 0.35, (n, 3)), np.zeros((n, 3))], 1))
    cam2[:, :9] = np.einsum("nij,njk->nik", cam[:, :9].reshape(n, 3, 3), dR).reshape(n, 9)
    z2 = z + rng.normal(0, 40, (n, 2))
    nd = np.array([1.0 / 0.7])
    J = np.zeros((n, 26)); e = np.zeros(n)
    bad = hm.hm_sfm_linearize_at_infinity(C.c_long(n), P(cam2), P(dirs), P(z2), C.c_int(1), P(nd), P(J), P(e))
    n_behind = 0
    for i in range(n):
        try:
            pi, Dc, Dp = O.sfm_project_at_infinity(cam2[i], dirs[i])
        except RuntimeError:
            n_behind += 1
            assert not J[i].any() and e[i] == 0.0
            continue
        ref = np.concatenate([(Dc / 0.7).reshape(-1), np.concatenate([Dp, np.zeros((2, 1))], 1).reshape(-1) / 0.7, (z2[i] - pi) / 0.7])
        assert np.abs(J[i] - ref).max() <= 1e-11 * np.abs(ref).max(), i
        assert not J[i, 3:6].any() and not J[i, 12:15].any() and J[i, 20] == 0.0 and J[i, 23] == 0.0
        assert abs(e[i] - 0.5 * (ref[24:] ** 2).sum()) <= 1e-12 * max(e[i], 1.0)
    assert bad == n_behind and n_behind < n // 4
    # and the Jacobian IS the derivative: central differences on the rotation and on the direction's tangent plane
    i = int(np.flatnonzero(J.any(1))[0]); h = 1e-6
    B = O.unit3_basis(dirs[i])
    for k in range(2):
        dp = dirs[i] + h * B[:, k]; dm = dirs[i] - h * B[:, k]
        num = (O.sfm_project_at_infinity(cam2[i], dp / np.linalg.norm(dp))[0] - O.sfm_project_at_infinity(cam2[i], dm / np.linalg.norm(dm))[0]) / (2 * h)
        assert np.abs(num / 0.7 - J[i, [18 + k, 21 + k]]).max() <= 1e-5 * np.abs(J[i, 18:24]).max()
    for k in range(3):
        w = np.zeros((1, 6)); w[0, k] = h
        Rp, _ = O.pose3_expmap(w); Rm, _ = O.pose3_expmap(-w)
        cp = cam2[i].copy(); cp[:9] = (cam2[i, :9].reshape(3, 3) @ Rp[0]).reshape(-1)
        cm = cam2[i].copy(); cm[:9] = (cam2[i, :9].reshape(3, 3) @ Rm[0]).reshape(-1)
        num = (O.sfm_project_at_infinity(cp, dirs[i])[0] - O.sfm_project_at_infinity(cm, dirs[i])[0]) / (2 * h)
        assert np.abs(num / 0.7 - J[i, [k, 9 + k]]).max() <= 1e-5 * np.abs(J[i, :18]).max()
