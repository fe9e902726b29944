"""oracle/gtsam_oracle.py -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).

A plain numpy restatement of borglab/gtsam's Levenberg-Marquardt hot path
(linearize -> damp -> eliminate -> solve -> retract -> error), vectorised over factors, every
function citing the reference file:line it follows (paths relative to /root/reference/gtsam).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product
path (gtsam_amd/) never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
  * literals of the reference's own tests (testGeneralSFMFactorB.cpp:44-63 -> 0.0199833 +- 1e-5,
    testProjectionFactor.cpp:96-189, testCholesky.cpp:26-67, testBetweenFactor.cpp, testSO3/testPose3
    identities), and
  * golden fixtures in tests/golden/*.npz produced by the real reference built from
    /root/reference (oracle/Makefile -> oracle/_ref, generator tests/golden/make_golden.py), and
  * the live reference (oracle/ref.py) whenever oracle/_ref is present.

All arithmetic is IEEE double, like the reference (base/Matrix.h:39).
"""
from __future__ import annotations

import math

import numpy as np

from gtsam_amd.problem import (FAC_BETWEEN_POSE3, FAC_GENERAL_SFM, FAC_PRIOR, FAC_PROJECTION,
                               NOISE_DIAGONAL, NOISE_GAUSSIAN, NOISE_ISOTROPIC, NOISE_UNIT,
                               ROBUST_CAUCHY, ROBUST_FAIR, ROBUST_GEMANMCCLURE, ROBUST_HUBER, ROBUST_NONE,
                               ROBUST_TUKEY, ROBUST_WELSCH, ROBUST_DCS, ROBUST_L2WITHDEADZONE,
                               STORAGE, TANGENT, VAR_POINT3, VAR_POSE2, VAR_POSE3, VAR_SFM_CAMERA, Problem)

EPS = np.finfo(np.float64).eps


# ------------------------------------------------------------------------------------------------
# SO(3) / SE(3)   (geometry/SO3.cpp, geometry/Pose3.cpp, base/Lie.h)
# ------------------------------------------------------------------------------------------------
def skew(w):
    """skewSymmetric(wx,wy,wz) (base/Matrix.h). w: [n,3] -> [n,3,3]."""
    w = np.asarray(w, np.float64)
    z = np.zeros(w.shape[0])
    return np.stack([np.stack([z, -w[:, 2], w[:, 1]], -1),
                     np.stack([w[:, 2], z, -w[:, 0]], -1),
                     np.stack([-w[:, 1], w[:, 0], z], -1)], -2)


def so3_expmap(omega):
    """SO3::Expmap via so3::ExpmapFunctor (geometry/SO3.cpp:50-88,200-208):
    near zero (theta^2 <= eps) R = I + W, else R = I + sin(theta) K + 2 sin^2(theta/2) K^2."""
    omega = np.asarray(omega, np.float64).reshape(-1, 3)
    theta2 = np.einsum("ni,ni->n", omega, omega)
    theta = np.sqrt(theta2)
    W = skew(omega)
    near = theta2 <= EPS
    th = np.where(near, 1.0, theta)
    K = W / th[:, None, None]
    KK = K @ K
    s2 = np.sin(th / 2.0)
    R = np.eye(3)[None] + np.sin(th)[:, None, None] * K + (2.0 * s2 * s2)[:, None, None] * KK
    R[near] = np.eye(3)[None] + W[near]
    return R


def so3_logmap(R):
    """SO3::Logmap (geometry/SO3.cpp:247-323): three regimes -- trace near -1 (largest-diagonal
    branch), the normal acos branch (tr-3 < -1e-6) and the Taylor branch near the identity."""
    R = np.asarray(R, np.float64).reshape(-1, 3, 3)
    n = R.shape[0]
    out = np.zeros((n, 3))
    for i in range(n):
        R11, R12, R13 = R[i, 0]; R21, R22, R23 = R[i, 1]; R31, R32, R33 = R[i, 2]
        tr = R11 + R22 + R33
        if tr + 1.0 < 1e-3:
            if R33 > R22 and R33 > R11:
                W = R21 - R12; Q1 = 2.0 + 2.0 * R33; Q2 = R31 + R13; Q3 = R23 + R32; perm = (1, 2, 0)
                vec = (Q2, Q3, Q1)
            elif R22 > R11:
                W = R13 - R31; Q1 = 2.0 + 2.0 * R22; Q2 = R23 + R32; Q3 = R12 + R21
                vec = (Q3, Q1, Q2)
            else:
                W = R32 - R23; Q1 = 2.0 + 2.0 * R11; Q2 = R12 + R21; Q3 = R31 + R13
                vec = (Q1, Q2, Q3)
            r = math.sqrt(Q1)
            one_over_r = 1 / r
            norm = math.sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W)
            sgn_w = -1.0 if W < 0 else 1.0
            mag = math.pi - (2 * sgn_w * W) / norm
            scale = 0.5 * one_over_r * mag
            out[i] = sgn_w * scale * np.array(vec)
        else:
            tr_3 = tr - 3.0
            if tr_3 < -1e-6:
                theta = math.acos((tr - 1.0) / 2.0)
                magnitude = theta / (2.0 * math.sin(theta))
            else:
                magnitude = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0
            out[i] = magnitude * np.array([R32 - R23, R13 - R31, R21 - R12])
    return out


def pose3_expmap(xi):
    """Pose3::Expmap (geometry/Pose3.cpp:169-185). xi = [omega; v] -> (R [n,3,3], t [n,3])."""
    xi = np.asarray(xi, np.float64).reshape(-1, 6)
    omega, v = xi[:, :3], xi[:, 3:]
    R = so3_expmap(omega)
    theta2 = np.einsum("ni,ni->n", omega, omega)
    big = theta2 > EPS
    t_parallel = omega * np.einsum("ni,ni->n", omega, v)[:, None]
    oxv = np.cross(omega, v)
    th2 = np.where(big, theta2, 1.0)
    t = (oxv - np.einsum("nij,nj->ni", R, oxv) + t_parallel) / th2[:, None]
    t[~big] = v[~big]
    return R, t


def pose3_logmap(R, t):
    """Pose3::Logmap (geometry/Pose3.cpp:188-208)."""
    R = np.asarray(R, np.float64).reshape(-1, 3, 3)
    T = np.asarray(t, np.float64).reshape(-1, 3)
    w = so3_logmap(R)
    th = np.linalg.norm(w, axis=1)
    out = np.concatenate([w, T], 1)
    big = ~(th < 1e-10)
    if big.any():
        wb, Tb, tb = w[big], T[big], th[big]
        W = skew(wb / tb[:, None])
        Tan = np.tan(0.5 * tb)
        WT = np.einsum("nij,nj->ni", W, Tb)
        u = Tb - (0.5 * tb)[:, None] * WT + (1 - tb / (2.0 * Tan))[:, None] * np.einsum("nij,nj->ni", W, WT)
        out[big, 3:] = u
    return out


def pose_unpack(p12):
    p12 = np.asarray(p12, np.float64).reshape(-1, 12)
    return p12[:, :9].reshape(-1, 3, 3), p12[:, 9:12]


def pose_pack(R, t):
    return np.concatenate([R.reshape(-1, 9), t.reshape(-1, 3)], 1)


def pose_compose(R1, t1, R2, t2):
    """Pose3::operator* (geometry/Pose3.h:114-116)."""
    return R1 @ R2, t1 + np.einsum("nij,nj->ni", R1, t2)


def pose_inverse(R, t):
    """Pose3::inverse (geometry/Pose3.cpp:49-52)."""
    Rt = np.swapaxes(R, 1, 2)
    return Rt, np.einsum("nij,nj->ni", Rt, -t)


def pose_adjoint(R, t):
    """Pose3::AdjointMap (geometry/Pose3.cpp:57-63): [R 0; [t]x R  R]."""
    n = R.shape[0]
    A = skew(t) @ R
    adj = np.zeros((n, 6, 6))
    adj[:, :3, :3] = R; adj[:, 3:, :3] = A; adj[:, 3:, 3:] = R
    return adj


def pose_retract(p12, xi):
    """LieGroup::retract = compose(Expmap(v)) (base/Lie.h:131-133, GTSAM_POSE3_EXPMAP)."""
    R, t = pose_unpack(p12)
    dR, dt = pose3_expmap(xi)
    return pose_pack(*pose_compose(R, t, dR, dt))


def pose_local(p12, q12):
    """LieGroup::localCoordinates = Logmap(between(g)) (base/Lie.h:136-138)."""
    R1, t1 = pose_unpack(p12); R2, t2 = pose_unpack(q12)
    return pose3_logmap(*pose_compose(*pose_inverse(R1, t1), R2, t2))


# ------------------------------------------------------------------------------------------------
# cameras   (geometry/CalibratedCamera.cpp, PinholePose.h, PinholeCamera.h, Cal3Bundler.cpp, Cal3_S2.cpp)
# ------------------------------------------------------------------------------------------------
def _project2(R, t, pw):
    """PinholeBase::project2 (CalibratedCamera.cpp:116-135) with Pose3::transformTo (Pose3.cpp:371-388),
    Project (:88-94), Dpose (:27-34), Dpoint (:37-46).  Returns pn, Dpose[n,2,6], Dpoint[n,2,3], behind."""
    Rt = np.swapaxes(R, 1, 2)
    q = np.einsum("nij,nj->ni", Rt, pw - t)
    behind = q[:, 2] <= 0  # CheiralityException (GTSAM_THROW_CHEIRALITY_EXCEPTION default ON)
    qz = np.where(behind, 1.0, q[:, 2])
    d = 1.0 / qz
    u, v = q[:, 0] * d, q[:, 1] * d
    uv, uu, vv = u * v, u * u, v * v
    z = np.zeros_like(u)
    Dpose = np.stack([np.stack([uv, -1 - uu, v, -d, z, d * u], -1),
                      np.stack([1 + vv, -uv, -u, z, -d, d * v], -1)], -2)
    Dpoint = np.stack([Rt[:, 0, :] - u[:, None] * Rt[:, 2, :],
                       Rt[:, 1, :] - v[:, None] * Rt[:, 2, :]], -2) * d[:, None, None]
    return np.stack([u, v], -1), Dpose, Dpoint, behind


def sfm_project(cam17, pw):
    """PinholeCamera<Cal3Bundler>::project2 (PinholeCamera.h:230-248) = PinholeBaseK::_project
    (PinholePose.h:89-109) + Cal3Bundler::uncalibrate (Cal3Bundler.cpp:66-92).
    Returns pi[n,2], Dcam[n,2,9] = [Dpose | Dcal], Dpoint[n,2,3], behind[n]."""
    cam17 = np.asarray(cam17, np.float64).reshape(-1, 17)
    R, t = pose_unpack(cam17[:, :12])
    f, k1, k2, u0, v0 = (cam17[:, 12 + i] for i in range(5))
    pn, Dpose, Dpoint, behind = _project2(R, t, np.asarray(pw, np.float64).reshape(-1, 3))
    x, y = pn[:, 0], pn[:, 1]
    r = x * x + y * y
    g = 1. + (k1 + k2 * r) * r
    u, v = g * x, g * y
    rx, ry = r * x, r * y
    Dcal = np.stack([np.stack([u, f * rx, f * r * rx], -1), np.stack([v, f * ry, f * r * ry], -1)], -2)
    a = 2. * (k1 + 2. * k2 * r)
    axx, axy, ayy = a * x * x, a * x * y, a * y * y
    Dp = np.stack([np.stack([g + axx, axy], -1), np.stack([axy, g + ayy], -1)], -2) * f[:, None, None]
    pi = np.stack([u0 + f * u, v0 + f * v], -1)
    Dcam = np.concatenate([Dp @ Dpose, Dcal], 2)
    return pi, Dcam, Dp @ Dpoint, behind


def s2_project(R, t, K5, pw, dist=None):
    """PinholeCamera<Cal3_S2>::project (PinholePose.h:89-109) + Cal3_S2::uncalibrate (Cal3_S2.cpp:44-50); with dist = (k1, k2,
    p1, p2) per row: + Cal3DS2_Base::uncalibrate (geometry/Cal3DS2_Base.cpp:93-132), point derivative D2dintrinsic (:71-91)."""
    pn, Dpose, Dpoint, behind = _project2(R, t, pw)
    fx, fy, s, u0, v0 = (K5[:, i] for i in range(5))
    x, y = pn[:, 0], pn[:, 1]
    z = np.zeros_like(fx)
    DK = np.stack([np.stack([fx, s], -1), np.stack([z, fy], -1)], -2)
    if dist is None:
        pi = np.stack([fx * x + s * y + u0, fy * y + v0], -1)
        return pi, DK @ Dpose, DK @ Dpoint, behind
    k1, k2, p1, p2 = (dist[:, i] for i in range(4))
    xy, xx, yy = x * y, x * x, y * y
    rr = xx + yy; r4 = rr * rr
    g = 1. + k1 * rr + k2 * r4
    dx = 2. * p1 * xy + p2 * (rr + 2. * xx)
    dy = 2. * p2 * xy + p1 * (rr + 2. * yy)
    pnx = g * x + dx; pny = g * y + dy
    pi = np.stack([fx * pnx + s * pny + u0, fy * pny + v0], -1)
    drdx, drdy = 2. * x, 2. * y
    dgdx = k1 * drdx + k2 * 2. * rr * drdx; dgdy = k1 * drdy + k2 * 2. * rr * drdy
    dDxdx = 2. * p1 * y + p2 * (drdx + 4. * x); dDxdy = 2. * p1 * x + p2 * drdy
    dDydx = 2. * p2 * y + p1 * drdx; dDydy = 2. * p2 * x + p1 * (drdy + 4. * y)
    DR = np.stack([np.stack([g + x * dgdx + dDxdx, x * dgdy + dDxdy], -1),
                   np.stack([y * dgdx + dDydx, g + y * dgdy + dDydy], -1)], -2)
    Dp = DK @ DR
    return pi, Dp @ Dpose, Dp @ Dpoint, behind


# ------------------------------------------------------------------------------------------------
# Pose2   (geometry/Pose2.{h,cpp}, Rot2.{h,cpp}); values are (x, y, theta), the reference keeps (c, s)
# ------------------------------------------------------------------------------------------------
def _rot2_normalize(c, s):
    """Rot2::normalize (Rot2.cpp:56-64): only when |c^2+s^2-1| > 1e-10."""
    scale = c * c + s * s
    bad = np.abs(scale - 1.0) > 1e-10
    k = np.where(bad, 1.0 / np.sqrt(np.where(bad, scale, 1.0)), 1.0)
    return c * k, s * k


def pose2_between_cs(xa, ya, ca, sa, xb, yb, cb, sb):
    """a^-1 b as (x, y, c, s): inverse (Pose2.cpp:201-203), compose (Pose2.h:131-133), Rot2 product through
    fromCosSin (Rot2.h:116-118), rotate / unrotate (Rot2.cpp:100-116)."""
    ix = ca * (-xa) + sa * (-ya); iy = -sa * (-xa) + ca * (-ya)
    c, s = _rot2_normalize(ca * cb - (-sa) * sb, (-sa) * cb + ca * sb)
    return ix + (ca * xb + sa * yb), iy + (-sa * xb + ca * yb), c, s


def pose2_local(a, b):
    """traits<Pose2>::Local(a, b) = ChartAtOrigin::Local(between(a, b)) = (x, y, theta) (Lie.h:136-138,
    Pose2.cpp:111-121 with GTSAM_SLOW_BUT_CORRECT_EXPMAP off)."""
    x, y, c, s = pose2_between_cs(a[:, 0], a[:, 1], np.cos(a[:, 2]), np.sin(a[:, 2]),
                                  b[:, 0], b[:, 1], np.cos(b[:, 2]), np.sin(b[:, 2]))
    return np.stack([x, y, np.arctan2(s, c)], 1)


def pose2_retract(a, v):
    """a * Pose2(v0, v1, v2) (Lie.h:131-133, Pose2.cpp:99-109)."""
    ca, sa = np.cos(a[:, 2]), np.sin(a[:, 2]); cv, sv = np.cos(v[:, 2]), np.sin(v[:, 2])
    c, s = _rot2_normalize(ca * cv - sa * sv, sa * cv + ca * sv)
    return np.stack([a[:, 0] + (ca * v[:, 0] + -sa * v[:, 1]), a[:, 1] + (sa * v[:, 0] + ca * v[:, 1]), np.arctan2(s, c)], 1)


# ------------------------------------------------------------------------------------------------
# noise models   (linear/NoiseModel.cpp)
# ------------------------------------------------------------------------------------------------
def noise_sqrt_info(p: Problem, idx: int):
    """The matrix W with whiten(v) = W v: Unit I; Isotropic invsigma*I (:641-663, invsigma_=1/sigma);
    Diagonal diag(1/sigmas) (:275-281,:311-325); Gaussian R (:163-181)."""
    kind, dim, off = int(p.noise_kind[idx]), int(p.noise_dim[idx]), int(p.noise_off[idx])
    d = p.noise_data
    if kind == NOISE_UNIT:
        return np.eye(dim)
    if kind == NOISE_ISOTROPIC:
        return np.eye(dim) * (1.0 / d[off])
    if kind == NOISE_DIAGONAL:
        return np.diag(1.0 / d[off:off + dim])
    return d[off:off + dim * dim].reshape(dim, dim).copy()


def _whiten_many(p, noise_idx, *arrays):
    """Apply whiten per factor to vectors [n,m] / matrices [n,m,k]."""
    outs = [np.array(a, np.float64, copy=True) for a in arrays]
    for ni in np.unique(noise_idx):
        sel = noise_idx == ni
        if int(p.noise_kind[ni]) == NOISE_UNIT:
            continue
        W = noise_sqrt_info(p, int(ni))
        for a in outs:
            if a.ndim == 2:
                a[sel] = np.einsum("ij,nj->ni", W, a[sel])
            else:
                a[sel] = np.einsum("ij,njk->nik", W, a[sel])
    return outs


# ------------------------------------------------------------------------------------------------
# factors: unwhitened error + Jacobians, then NoiseModelFactor::linearize
# ------------------------------------------------------------------------------------------------
def _values_views(p: Problem, values):
    values = np.asarray(values, np.float64)
    return values, p.val_offsets()


def _gather(values, off, ids, size):
    idx = off[ids][:, None] + np.arange(size)[None, :]
    return values[idx]


# ------------------------------------------------------------------------------------------------
# m-estimators   (linear/LossFunctions.cpp) and noiseModel::Robust (linear/NoiseModel.h:670-760)
# ------------------------------------------------------------------------------------------------
def robust_weight(rkind, k, d):
    """mEstimator::*::weight(distance): Fair :146, Huber :179, Cauchy :217, Tukey :250, Welsch :289, GemanMcClure :320."""
    d = np.asarray(d, np.float64); a = np.abs(d)
    if rkind == ROBUST_FAIR:
        return 1.0 / (1.0 + a / k)
    if rkind == ROBUST_HUBER:
        return np.where(a <= k, 1.0, k / np.where(a > 0, a, 1.0))
    if rkind == ROBUST_CAUCHY:
        return (k * k) / (k * k + d * d)
    if rkind == ROBUST_TUKEY:
        return np.where(a <= k, (1.0 - d * d / (k * k)) ** 2, 0.0)
    if rkind == ROBUST_WELSCH:
        return np.exp(-(d * d) / (k * k))
    if rkind == ROBUST_GEMANMCCLURE:
        return (k ** 4) / (k * k + d * d) ** 2
    if rkind == ROBUST_DCS:                                  # :355-364 (the parameter is compared with the SQUARED distance)
        return np.where(d * d > k, (2.0 * k / (k + d * d)) ** 2, 1.0)
    if rkind == ROBUST_L2WITHDEADZONE:                       # :402-409 (distance >= 0 on this path)
        return np.where(a <= k, 0.0, (a - k) / np.where(a > 0, a, 1.0))
    return np.ones_like(d)


def robust_loss(rkind, k, d):
    """mEstimator::*::loss(distance): Fair :150, Huber :184, Cauchy :221, Tukey :258, Welsch :294, GemanMcClure :327."""
    d = np.asarray(d, np.float64); a = np.abs(d)
    if rkind == ROBUST_FAIR:
        return k * k * (a / k - np.log1p(a / k))
    if rkind == ROBUST_HUBER:
        return np.where(a <= k, d * d / 2, k * (a - k / 2))
    if rkind == ROBUST_CAUCHY:
        return k * k * np.log1p(d * d / (k * k)) * 0.5
    if rkind == ROBUST_TUKEY:
        return np.where(a <= k, k * k * (1 - (1 - d * d / (k * k)) ** 3) / 6.0, k * k / 6.0)
    if rkind == ROBUST_WELSCH:
        return k * k * 0.5 * -np.expm1(-(d * d) / (k * k))
    if rkind == ROBUST_GEMANMCCLURE:
        return 0.5 * (k * k * d * d) / (k * k + d * d)
    if rkind == ROBUST_DCS:                                  # :366-375
        e2 = d * d
        return (k * k * e2 + k * e2 * e2) / ((e2 + k) * (e2 + k))
    if rkind == ROBUST_L2WITHDEADZONE:                       # :411-414
        return np.where(a < k, 0.0, 0.5 * (k - a) ** 2)
    return 0.5 * d * d


def _robust_of(p, ni):
    rk = getattr(p, "noise_robust", None)
    if rk is None or ni >= len(rk):
        return ROBUST_NONE, 0.0
    return int(rk[ni]), float(p.noise_robust_param[ni])


def _reweight_many(p, noise_idx, A1, A2, b):
    """Robust::WhitenSystem = base WhitenSystem, then robust_->reweight(A.., b) with the Block scheme
    (LossFunctions.cpp:43-88): every block and b times sqrt(weight(||b||)).  In place on whitened data."""
    for ni in np.unique(noise_idx):
        rk, k = _robust_of(p, int(ni))
        if rk == ROBUST_NONE:
            continue
        sel = noise_idx == ni
        w = np.sqrt(robust_weight(rk, k, np.linalg.norm(b[sel], axis=1)))
        A1[sel] *= w[:, None, None]
        if A2 is not None:
            A2[sel] *= w[:, None, None]
        b[sel] *= w[:, None]


def _loss_many(p, noise_idx, b):
    """sum of noiseModel->loss(squaredMahalanobisDistance) over factors; b = whitened (un-reweighted) residual."""
    sq = np.sum(b * b, axis=1)
    e = 0.0
    for ni in np.unique(noise_idx):
        sel = noise_idx == ni
        rk, k = _robust_of(p, int(ni))
        if rk == ROBUST_NONE:
            e += 0.5 * float(np.sum(sq[sel]))
        else:                                                    # NoiseModel.h:716-718: robust_->loss(sqrt(d2))
            e += float(np.sum(robust_loss(rk, k, np.sqrt(sq[sel]))))
    return e


def linearize(p: Problem, values):
    """As _linearize_whitened, followed by the m-estimator re-weighting of Robust noise models for the factors that
    linearize through NoiseModelFactor::linearize -> WhitenSystem (NonlinearFactor.cpp:150-182).  GeneralSFMFactor is
    NOT re-weighted by its residual: its own linearize whitens H1, H2, b one by one through Robust::Whiten(Matrix), whose internal
    WhitenSystem sees an empty b and hence weight(0) -- 1 for every estimator but L2WithDeadZone (GeneralSFMFactor.h:162-168,
    NoiseModel.h:705-709; measured on the live reference)."""
    lin = _linearize_whitened(p, values)
    if getattr(p, "noise_robust", None) is None or not np.any(p.noise_robust):
        return lin
    if FAC_GENERAL_SFM in lin:          # weight(0): 1 for every estimator but L2WithDeadZone (0: the factor leaves the linear system)
        A1, A2, b = lin[FAC_GENERAL_SFM][:3]
        for ni in np.unique(p.sfm_noise):
            rk, k = _robust_of(p, int(ni))
            if rk != ROBUST_NONE:
                w = float(np.sqrt(robust_weight(rk, k, 0.0)))
                if w != 1.0:
                    sel = p.sfm_noise == ni
                    A1[sel] *= w; A2[sel] *= w; b[sel] *= w
    if FAC_PROJECTION in lin:
        _reweight_many(p, p.proj_noise, *lin[FAC_PROJECTION])
    if FAC_BETWEEN_POSE3 in lin:
        _reweight_many(p, p.between_noise, *lin[FAC_BETWEEN_POSE3])
    if FAC_PRIOR in lin:
        A, _, b, dims = lin[FAC_PRIOR]
        _reweight_many(p, p.prior_noise, A, None, b)
    return lin


def _linearize_whitened(p: Problem, values):
    """NonlinearFactorGraph::linearize (nonlinear/NonlinearFactorGraph.cpp:239-278): per factor the
    whitened [A1 A2 b] of NoiseModelFactor::linearize (NonlinearFactor.cpp:150-182) /
    GeneralSFMFactor::linearize (slam/GeneralSFMFactor.h:141-177).
    Returns dict type -> (A1, A2 or None, b), arrays stacked over factors."""
    values, off = _values_views(p, values)
    out = {}
    if p.n_sfm:
        cam = _gather(values, off, p.sfm_cam, 17); pt = _gather(values, off, p.sfm_point, 3)
        pi, Dc, Dp, behind = sfm_project(cam, pt)
        b = p.sfm_z.reshape(-1, 2) - pi
        Dc[behind] = 0; Dp[behind] = 0; b[behind] = 0          # GeneralSFMFactor.h:153-158
        Dc, Dp, b = _whiten_many(p, p.sfm_noise, Dc, Dp, b)
        out[FAC_GENERAL_SFM] = (Dc, Dp, b)
    if p.n_proj:
        R, t = pose_unpack(_gather(values, off, p.proj_pose, 12)); pt = _gather(values, off, p.proj_point, 3)
        K5 = p.calib.reshape(-1, 5)[p.proj_calib]
        cd = getattr(p, "calib_distortion", np.zeros(0))
        dist = cd.reshape(-1, 4)[p.proj_calib] if cd.size else None     # Cal3DS2 entries (zero rows = plain Cal3_S2)
        H0 = None
        has_s = p.proj_sensor >= 0 if p.proj_sensor.size else np.zeros(p.n_proj, bool)
        if has_s.any():                                          # ProjectionFactor.h:142-148
            sR, st = pose_unpack(p.sensor.reshape(-1, 12)[np.where(has_s, p.proj_sensor, 0)])
            cR, ct = pose_compose(R, t, sR, st)
            R = np.where(has_s[:, None, None], cR, R); t = np.where(has_s[:, None], ct, t)
            H0 = pose_adjoint(*pose_inverse(sR, st))             # compose H1 = g.inverse().AdjointMap() (Lie.h:56-61)
        pi, Dpose, Dpt, behind = s2_project(R, t, K5, pt, dist)
        if H0 is not None:
            Dpose = np.where(has_s[:, None, None], Dpose @ H0, Dpose)
        err = pi - p.proj_z.reshape(-1, 2)
        Dpose[behind] = 0; Dpt[behind] = 0                       # ProjectionFactor.h:157-165
        err[behind] = (2.0 * K5[behind, 0])[:, None]
        Dpose, Dpt, b = _whiten_many(p, p.proj_noise, Dpose, Dpt, -err)
        out[FAC_PROJECTION] = (Dpose, Dpt, b)
    if p.n_between and np.all(p.var_type[p.between_v1] == VAR_POSE2):
        # BetweenFactor<Pose2>: generic LieGroup::between (Lie.h:63-69), H1 = -Ad(h^-1) (Pose2.cpp:126-135), H2 = I,
        # r = Local(z, h); measurement = first 3 of the factor's 12 doubles
        a = _gather(values, off, p.between_v1, 3); b2 = _gather(values, off, p.between_v2, 3)
        z = p.between_z.reshape(-1, 12)[:, :3]
        hx, hy, hc, hs = pose2_between_cs(a[:, 0], a[:, 1], np.cos(a[:, 2]), np.sin(a[:, 2]),
                                          b2[:, 0], b2[:, 1], np.cos(b2[:, 2]), np.sin(b2[:, 2]))
        gx, gy, gc, gs = pose2_between_cs(z[:, 0], z[:, 1], np.cos(z[:, 2]), np.sin(z[:, 2]), hx, hy, hc, hs)
        err = np.stack([gx, gy, np.arctan2(gs, gc)], 1)
        ic, isn = hc, -hs                                           # h^-1
        ixx = hc * (-hx) + hs * (-hy); iyy = -hs * (-hx) + hc * (-hy)
        n = a.shape[0]
        H1 = np.zeros((n, 3, 3))
        H1[:, 0, 0] = -ic; H1[:, 0, 1] = isn; H1[:, 0, 2] = -iyy
        H1[:, 1, 0] = -isn; H1[:, 1, 1] = -ic; H1[:, 1, 2] = ixx
        H1[:, 2, 2] = -1.0
        H2 = np.broadcast_to(np.eye(3), H1.shape).copy()
        H1, H2, b = _whiten_many(p, p.between_noise, H1, H2, -err)
        out[FAC_BETWEEN_POSE3] = (H1, H2, b)
    elif p.n_between:
        if np.any(p.var_type[p.between_v1] == VAR_POSE2):
            raise NotImplementedError("oracle: a graph mixing BetweenFactor<Pose2> and <Pose3>")
        R1, t1 = pose_unpack(_gather(values, off, p.between_v1, 12))
        R2, t2 = pose_unpack(_gather(values, off, p.between_v2, 12))
        hR, ht = pose_compose(*pose_inverse(R1, t1), R2, t2)     # Lie.h:63-69 between
        H1 = -pose_adjoint(*pose_inverse(hR, ht))
        H2 = np.broadcast_to(np.eye(6), H1.shape).copy()
        zR, zt = pose_unpack(p.between_z.reshape(-1, 12))
        err = pose3_logmap(*pose_compose(*pose_inverse(zR, zt), hR, ht))  # BetweenFactor.h:113-123 (no Hlocal)
        H1, H2, b = _whiten_many(p, p.between_noise, H1, H2, -err)
        out[FAC_BETWEEN_POSE3] = (H1, H2, b)
    if p.n_prior:
        A = np.zeros((p.n_prior, 9, 9)); b = np.zeros((p.n_prior, 9)); dims = np.zeros(p.n_prior, np.int32)
        for k in range(p.n_prior):
            v = int(p.prior_var[k]); t = int(p.var_type[v]); d = TANGENT[t]
            x = values[off[v]:off[v] + STORAGE[t]]
            z = p.prior_data[p.prior_off[k]:p.prior_off[k] + STORAGE[t]]
            e = -local_coordinates(t, x, z)                      # PriorFactor.h:98-102, H = I
            W = noise_sqrt_info(p, int(p.prior_noise[k]))
            A[k, :d, :d] = W @ np.eye(d); b[k, :d] = W @ (-e); dims[k] = d
        out[FAC_PRIOR] = (A, None, b, dims)
    return out


def local_coordinates(vtype, x, z):
    """traits<T>::Local(x, z): Pose3 Logmap(between) ; PinholeCamera [pose local; calib diff]
    (PinholeCamera.h:208-213, Cal3Bundler.h:150-152); Point3 z - x."""
    if vtype == VAR_POSE3:
        return pose_local(x[None, :12], z[None, :12])[0]
    if vtype == VAR_POSE2:
        return pose2_local(x[None, :3], z[None, :3])[0]
    if vtype == VAR_SFM_CAMERA:
        return np.concatenate([pose_local(x[None, :12], z[None, :12])[0], z[12:15] - x[12:15]])
    return z - x


def jacobians_flat(p: Problem, values, ftype):
    """Same row layout as gtg_get_jacobians / ref_graph_jacobians."""
    lin = linearize(p, values)
    if ftype not in lin:
        return np.zeros((0, 0))
    if ftype == FAC_PRIOR:
        A, _, b, dims = lin[ftype]
        out = np.zeros((A.shape[0], 90))
        for k in range(A.shape[0]):
            d = dims[k]
            out[k, :d * d] = A[k, :d, :d].reshape(-1); out[k, 81:81 + d] = b[k, :d]
        return out
    A1, A2, b = lin[ftype]
    n = A1.shape[0]
    if ftype == FAC_BETWEEN_POSE3 and A1.shape[1] == 3:     # Pose2: 3x3 blocks inside the 78-double between record
        out = np.zeros((n, 78))
        out[:, 0:9] = A1.reshape(n, -1); out[:, 36:45] = A2.reshape(n, -1); out[:, 72:75] = b
        return out
    return np.concatenate([A1.reshape(n, -1), A2.reshape(n, -1), b], 1)


# ------------------------------------------------------------------------------------------------
# SmartProjectionFactor<PinholeCamera<Cal3Bundler>> (slam/SmartProjectionFactor.h, SmartFactorBase.h, geometry/triangulation.*)
# Restated WITHOUT the factor's triangulation cache (SmartProjectionFactor.h:127-183): every call triangulates afresh, which is
# what the reference does whenever some camera pose moved by more than retriangulationThreshold -- true for the probes the
# fixtures hold (error, Hessian diagonal, one damped solve and its trial error from the initial values).
# ------------------------------------------------------------------------------------------------
TRI_VALID, TRI_DEGENERATE, TRI_BEHIND, TRI_OUTLIER, TRI_FAR = 0, 1, 2, 3, 4


def bundler_calibrate(f, k1, k2, u0, v0, pi):
    """Cal3Bundler::calibrate (geometry/Cal3Bundler.cpp:95-128): the reference's fixed-point iteration, tol 1e-5, <= 10 rounds."""
    px, py = (pi[0] - u0) / f, (pi[1] - v0) / f
    ix, iy = px, py
    for _ in range(10):
        rr = px * px + py * py
        g = 1 + k1 * rr + k2 * rr * rr
        pn = np.array([ix / g, iy / g])
        r = pn @ pn; g2 = 1. + (k1 + k2 * r) * r
        if np.hypot(u0 + f * g2 * pn[0] - pi[0], v0 + f * g2 * pn[1] - pi[1]) <= 1e-5:
            return pn
        px, py = pn
    raise RuntimeError("Cal3Bundler::calibrate fails to converge")


def triangulate_nonlinear(cams17, z, point0):
    """gtsam::triangulateNonlinear (geometry/triangulation.h:211-221): the DLT point refined by LevenbergMarquardtOptimizer on one
    TriangulationFactor per camera (slam/TriangulationFactor.h:121-136: h(x) - z with the camera's full projection, unit noise; a
    point behind the camera is the constant error (2 fx, 2 fx), PinholeCamera.h:323-325; LINEARISING there throws) with the parameters of
    triangulation.cpp:177-195: lambdaInitial 1, lambdaFactor 10, at most 100 iterations, absoluteErrorTol 1.0, the other defaults
    (relativeErrorTol 1e-5, lambdaUpperBound 1e5, minModelFidelity 1e-3, identity damping, fixed factor).  The state machine is
    lm_optimize's (LM.cpp:121-308, NonlinearOptimizer.cpp:62-117), on a single 3-dimensional variable."""
    m = cams17.shape[0]

    def residuals(pt, jac):
        pi, Dc, Dp, behind = sfm_project(cams17, np.repeat(pt[None], m, 0))
        r = pi - z
        r[behind] = 2.0 * cams17[behind, 12:13]
        if jac:           # TriangulationFactor::linearize projects without evaluateError's try / catch (TriangulationFactor.h:148-170)
            if behind.any():
                raise RuntimeError("CheiralityException")
            return r.reshape(-1), Dp.reshape(-1, 3)
        return r.reshape(-1), None

    def err_of(pt):
        r, _ = residuals(pt, False)
        return 0.5 * float(r @ r)
    pt = np.asarray(point0, np.float64).copy()
    err = err_of(pt)
    lam, factor = 1.0, 10.0
    iterations = 0
    if err <= 0.0:
        return pt
    new_error = err
    while True:
        current_error = new_error
        r, A = residuals(pt, True)
        b = -r
        H = A.T @ A; g = A.T @ b
        while True:
            ABC = np.zeros((4, 4)); ABC[:3, :3] = H + lam * np.eye(3); ABC[:3, 3] = g; ABC[3, :3] = g; ABC[3, 3] = b @ b
            ok = cholesky_partial(ABC, 3)[0]                    # the one clique of the damped system (Eigen LLT + the rank test)
            step_ok = False; stop = False
            trial_err = math.inf; trial = None
            if ok:
                delta = np.linalg.solve(H + lam * np.eye(3), g)
                old_lin = 0.5 * float(b @ b); new_lin = 0.5 * float((A @ delta - b) @ (A @ delta - b))
                lin_change = old_lin - new_lin
                if lin_change >= 0:
                    trial = pt + delta
                    trial_err = err_of(trial)
                    cost_change = err - trial_err
                    if lin_change > EPS * old_lin:
                        step_ok = cost_change / lin_change > 1e-3
                    if abs(cost_change) < 1e-5 * err:
                        stop = True
            if step_ok:
                lam = max(0.0, lam / factor)
                pt, err = trial, trial_err
                iterations += 1
                break
            elif not stop:
                lam *= factor
                if lam >= 1e5:
                    break
            else:
                break
        new_error = err
        if not (iterations < 100 and not check_convergence(1e-5, 1.0, 0.0, current_error, new_error) and math.isfinite(current_error)):
            break
    return pt


def triangulate_safe(cams17, z, rank_tol=1.0, dist_thr=-1.0, outlier_thr=-1.0, enable_epi=False):
    """gtsam::triangulateSafe (geometry/triangulation.h:697-752) with useLOST = false: undistort, DLT by SVD
    (triangulation.cpp:27-57, base/Matrix.cpp:566-584), the nonlinear refinement when enableEPI is set (triangulation.h:531-534),
    cheirality, distance and outlier checks.  -> (status, point)."""
    m = cams17.shape[0]
    if m < 2:
        return TRI_DEGENERATE, None
    A = np.zeros((2 * m, 4))
    for k in range(m):
        c = cams17[k]; R = c[:9].reshape(3, 3); t = c[9:12]; f, k1, k2, u0, v0 = c[12:17]
        pn = bundler_calibrate(f, k1, k2, u0, v0, z[k])
        zu = np.array([f * pn[0] + u0, f * pn[1] + v0])
        K = np.array([[f, 0, u0], [0, f, v0], [0, 0, 1.0]])
        P = K @ np.concatenate([R.T, (-R.T @ t)[:, None]], 1)
        A[2 * k] = zu[0] * P[2] - P[0]; A[2 * k + 1] = zu[1] * P[2] - P[1]
    _, sv, Vt = np.linalg.svd(A)
    if int(np.sum(sv[:min(2 * m, 4)] > rank_tol)) < 3:
        return TRI_DEGENERATE, None
    v = Vt[-1]; pt = v[:3] / v[3]
    if enable_epi:
        pt = triangulate_nonlinear(cams17, z, pt)
    for k in range(m):
        R = cams17[k, :9].reshape(3, 3)
        if (R.T @ (pt - cams17[k, 9:12]))[2] <= 0:
            return TRI_BEHIND, None
    max_err = 0.0
    for k in range(m):
        if dist_thr > 0 and np.linalg.norm(pt - cams17[k, 9:12]) > dist_thr:
            return TRI_FAR, None
        if outlier_thr > 0:
            pi, _, _, _ = sfm_project(cams17[k:k + 1], pt[None])
            max_err = max(max_err, float(np.linalg.norm(pi[0] - z[k])))
    if outlier_thr > 0 and max_err > outlier_thr:
        return TRI_OUTLIER, None
    return TRI_VALID, pt


def unit3_basis(n):
    """Unit3::basis (geometry/Unit3.cpp:73-135): b1 = normalize(n x axis) with the coordinate axis of the smallest |n_i| (ties: x
    first, then y), b2 = n x b1.  -> 3x2."""
    mx, my, mz = np.abs(n)
    axis = np.array([1.0, 0, 0]) if (mx <= my and mx <= mz) else (np.array([0, 1.0, 0]) if (my <= mx and my <= mz) else np.array([0, 0, 1.0]))
    b1 = np.cross(n, axis); b1 = b1 / np.linalg.norm(b1)
    return np.stack([b1, np.cross(n, b1)], 1)


def backproject_point_at_infinity(cam17, z):
    """PinholeBaseK::backprojectPointAtInfinity (geometry/PinholePose.h:164-168): rotate Unit3(calibrate(z), 1) into the world."""
    pn = bundler_calibrate(*cam17[12:17], z)
    pc = np.array([pn[0], pn[1], 1.0]); pc /= np.linalg.norm(pc)
    w = cam17[:9].reshape(3, 3) @ pc
    return w / np.linalg.norm(w)                                             # Rot3::rotate(Unit3) normalises again (Rot3.cpp:109-116)


def sfm_project_at_infinity(cam17, d):
    """PinholeCamera<Cal3Bundler>::project2(Unit3) (geometry/PinholeCamera.h:251-254 -> PinholePose.h:89-109 ->
    CalibratedCamera.cpp:138-165, Rot3::unrotate(Unit3) Rot3.cpp:119-126, PinholeBase::Project(Unit3) CalibratedCamera.cpp:97-106),
    the chain of Jacobians as the reference multiplies it.  -> pi, Dcam 2x9, Dpoint 2x2; raises on the CheiralityException."""
    R = cam17[:9].reshape(3, 3); f, k1, k2, u0, v0 = cam17[12:17]
    q = R.T @ d; q = q / np.linalg.norm(q)
    if q[2] <= 0:
        raise RuntimeError("CheiralityException")
    Bq, Bd = unit3_basis(q), unit3_basis(d)
    Dpc_rot = Bq.T @ skew(q[None])[0]                                               # 2x3
    Dpc_point = Bq.T @ R.T @ Bd                                              # 2x2
    dz = 1.0 / q[2]; u, v = q[0] * dz, q[1] * dz
    Dpn_pc = np.array([[dz, 0, -u * dz], [0, dz, -v * dz]]) @ Bq             # 2x2
    Dpose = np.zeros((2, 6)); Dpose[:, :3] = Dpn_pc @ Dpc_rot
    Dpoint = Dpn_pc @ Dpc_point
    r = u * u + v * v; g = 1. + (k1 + k2 * r) * r                            # Cal3Bundler::uncalibrate (Cal3Bundler.cpp:66-92)
    pi = np.array([u0 + f * g * u, v0 + f * g * v])
    a = 2. * (k1 + 2. * k2 * r)
    Dp = f * np.array([[g + a * u * u, a * u * v], [a * u * v, g + a * v * v]])
    Dcal = np.array([[g * u, f * r * u, f * r * r * u], [g * v, f * r * v, f * r * r * v]])
    return pi, np.concatenate([Dp @ Dpose, Dcal], 1), Dp @ Dpoint


SMART_HESSIAN, SMART_IMPLICIT_SCHUR, SMART_JACOBIAN_Q, SMART_JACOBIAN_SVD = 0, 1, 2, 3      # LinearizationMode (SmartFactorParams.h:31-33)


def _smart_factors(p: Problem, values, for_error=False):
    """Per smart factor: (camera ids, whitened F blocks [m,2,9], E [2m,N], b [2m]) or None where the factor contributes nothing.
    N = 3 for a VALID triangulation.  Otherwise (the optional<Point3> of TriangulationResult is empty for every other status):
      linearising, HESSIAN mode: ZERO_ON_DEGENERACY -> nothing (SmartProjectionFactor.h:212-219); IGNORE_DEGENERACY and
        HANDLE_INFINITY -> the point at infinity of the first measurement, N = 2 (computeJacobiansWithTriangulatedPoint, :356-371);
      linearising, JACOBIAN_Q / JACOBIAN_SVD: an empty factor whatever the degeneracy mode (:245-272);
      error(): HANDLE_INFINITY -> the point at infinity, every other mode 0.0 (totalReprojectionError, :407-427)."""
    off = p.val_offsets(); values = np.asarray(values, np.float64)
    out = []
    for i in range(p.n_smart):
        k0, k1 = int(p.smart_ptr[i]), int(p.smart_ptr[i + 1])
        cams = p.smart_cam[k0:k1]; z = p.smart_z.reshape(-1, 2)[k0:k1]
        c17 = np.stack([values[off[c]:off[c] + 17] for c in cams])
        prm = p.smart_params.reshape(-1, 8)[i]
        st, pt = triangulate_safe(c17, z, prm[0], prm[1], prm[2], enable_epi=bool(prm[6]))
        W = noise_sqrt_info(p, int(p.smart_noise[i]))                      # 2x2 (Unit / Isotropic)
        if st != TRI_VALID:
            at_infinity = (prm[4] == 2.0) if for_error else (prm[4] != 1.0 and prm[5] == SMART_HESSIAN)
            if not at_infinity:
                out.append(None); continue
            d = backproject_point_at_infinity(c17[0], z[0])
            pr = [sfm_project_at_infinity(c17[k], d) for k in range(len(cams))]
            pi = np.stack([q[0] for q in pr]); Dc = np.stack([q[1] for q in pr]); Dp = np.stack([q[2] for q in pr])
        else:
            pi, Dc, Dp, behind = sfm_project(c17, np.repeat(pt[None], len(cams), 0))
        F = np.einsum("ij,mjk->mik", W, Dc); E = np.einsum("ij,mjk->mik", W, Dp).reshape(-1, Dp.shape[2])
        b = (-(pi - z) @ W.T).reshape(-1)                                   # b = -whiten(h(x) - z), SmartFactorBase.h:296-316
        out.append(([int(c) for c in cams], F, E, b))
    return out


def _smart_hessians(p: Problem, values):
    """createHessianFactor (SmartProjectionFactor.h:190-233) = CameraSet::SchurComplement (CameraSet.h:174-226) with lambda = 0:
    G = F^T F - F^T E P E^T F, g = F^T (b - E P E^T b), f = b^T b, P = (E^T E)^-1 (3x3, or 2x2 at infinity).
    JACOBIAN_Q (JacobianFactorQ.h:53-79: Q F, Q b with the projector Q = I - E P E^T) and JACOBIAN_SVD (JacobianFactorSVD.h:
    Enull^T F, Enull^T b, Enull Enull^T = Q) have the same G and g, and the constant b^T Q b.  -> list of (camera ids, G, g, f)."""
    res = []
    modes = p.smart_params.reshape(-1, 8)[:, 5]
    for i, sf in enumerate(_smart_factors(p, values)):
        if sf is None:
            continue
        cams, F, E, b = sf
        m = len(cams)
        P = np.linalg.inv(E.T @ E)
        Fd = np.zeros((2 * m, 9 * m))
        for k in range(m):
            Fd[2 * k:2 * k + 2, 9 * k:9 * k + 9] = F[k]
        Q = np.eye(2 * m) - E @ P @ E.T
        res.append((cams, Fd.T @ Q @ Fd, Fd.T @ (Q @ b), float(b @ b) if modes[i] == SMART_HESSIAN else float(b @ Q @ b)))
    return res


def error(p: Problem, values):
    """NonlinearFactorGraph::error (NonlinearFactorGraph.cpp:170-179) = sum 0.5*||whiten(r)||^2
    (NonlinearFactor.cpp:136-147), or the m-estimator loss of ||whiten(r)|| for Robust models.  b of the un-reweighted
    linearization is -whiten(r)."""
    lin = _linearize_whitened(p, values)
    nz = {FAC_GENERAL_SFM: p.sfm_noise, FAC_PROJECTION: p.proj_noise, FAC_BETWEEN_POSE3: p.between_noise,
          FAC_PRIOR: p.prior_noise}
    e = 0.0
    for ft, tup in lin.items():
        e += _loss_many(p, nz[ft], tup[2])
    if getattr(p, "n_smart", 0):                          # totalReprojectionError (SmartProjectionFactor.h:407-427): 0.5 |b|^2 or 0
        for sf in _smart_factors(p, values, for_error=True):
            if sf is not None:
                e += 0.5 * float(sf[3] @ sf[3])
    return e


# ------------------------------------------------------------------------------------------------
# linear algebra of the solve
# ------------------------------------------------------------------------------------------------
def _factor_blocks(p: Problem, lin):
    """Yield (var ids tuple, [A blocks], b) per linear factor."""
    if FAC_GENERAL_SFM in lin:
        A1, A2, b = lin[FAC_GENERAL_SFM]
        for k in range(A1.shape[0]):
            yield (int(p.sfm_cam[k]), int(p.sfm_point[k])), (A1[k], A2[k]), b[k]
    if FAC_PROJECTION in lin:
        A1, A2, b = lin[FAC_PROJECTION]
        for k in range(A1.shape[0]):
            yield (int(p.proj_pose[k]), int(p.proj_point[k])), (A1[k], A2[k]), b[k]
    if FAC_BETWEEN_POSE3 in lin:
        A1, A2, b = lin[FAC_BETWEEN_POSE3]
        for k in range(A1.shape[0]):
            yield (int(p.between_v1[k]), int(p.between_v2[k])), (A1[k], A2[k]), b[k]
    if FAC_PRIOR in lin:
        A, _, b, dims = lin[FAC_PRIOR]
        for k in range(A.shape[0]):
            d = dims[k]
            yield (int(p.prior_var[k]),), (A[k, :d, :d],), b[k, :d]


def hessian_dense(p: Problem, values):
    """Dense information matrix J^T J and gradient J^T b in variable-id order (the sum of
    JacobianFactor::updateHessian contributions, linear/JacobianFactor.cpp:563-598).  Small problems."""
    lin = linearize(p, values)
    doff = p.dim_offsets(); n = int(doff[-1])
    H = np.zeros((n, n)); g = np.zeros(n)
    for vids, As, b in _factor_blocks(p, lin):
        for i, vi in enumerate(vids):
            si = slice(doff[vi], doff[vi + 1])
            g[si] += As[i].T @ b
            for j, vj in enumerate(vids):
                H[si, slice(doff[vj], doff[vj + 1])] += As[i].T @ As[j]
    if getattr(p, "n_smart", 0):
        lin = dict(lin); lin["smart"] = _smart_hessians(p, values)
        for cams, G, gg, f in lin["smart"]:
            for a, ca in enumerate(cams):
                g[doff[ca]:doff[ca + 1]] += gg[9 * a:9 * a + 9]
                for bb, cb in enumerate(cams):
                    H[doff[ca]:doff[ca + 1], doff[cb]:doff[cb + 1]] += G[9 * a:9 * a + 9, 9 * bb:9 * bb + 9]
    return H, g, lin


def hessian_diagonal(p: Problem, values):
    """GaussianFactorGraph::hessianDiagonal (GaussianFactorGraph.cpp:279-287) = sum of squared
    column norms (JacobianFactor::hessianDiagonalAdd JacobianFactor.cpp:516-541)."""
    lin = linearize(p, values)
    doff = p.dim_offsets(); d = np.zeros(int(doff[-1]))
    for vids, As, b in _factor_blocks(p, lin):
        for i, vi in enumerate(vids):
            d[doff[vi]:doff[vi + 1]] += np.sum(As[i] * As[i], 0)
    if getattr(p, "n_smart", 0):                          # HessianFactor::hessianDiagonal: the diagonal of the Schur-complemented G
        for cams, G, gg, f in _smart_hessians(p, values):
            for a, ca in enumerate(cams):
                d[doff[ca]:doff[ca + 1]] += np.diag(G)[9 * a:9 * a + 9]
    return d


def cholesky_partial(ABC, n_frontal):
    """gtsam::choleskyPartial (base/cholesky.cpp:107-158) on a symmetric matrix (upper triangle
    used): A = R^T R (Eigen LLT; fails on a non-positive pivot), B <- R^-T B, C <- C - B^T B, then
    the exponent test on the last two pivots (underconstrainedExponentDifference = 12).
    Returns (ok, matrix with [R S; . C'])."""
    M = np.array(ABC, np.float64, copy=True)
    n = M.shape[0]; nf = n_frontal
    if nf == 0:
        return True, M
    A = np.triu(M[:nf, :nf]); A = A + np.triu(A, 1).T
    R = np.zeros((nf, nf))
    for k in range(nf):                      # Eigen llt_inplace unblocked: x <= 0 -> NumericalIssue
        x = A[k, k] - R[:k, k] @ R[:k, k]
        if not (x > 0.0):
            return False, M
        R[k, k] = math.sqrt(x)
        if k + 1 < nf:
            R[k, k + 1:] = (A[k, k + 1:] - R[:k, k] @ R[:k, k + 1:]) / R[k, k]
    M[:nf, :nf] = R
    if nf < n:
        B = np.linalg.solve(R.T, M[:nf, nf:]) if nf > 0 else M[:nf, nf:]
        # forward substitution (TRSM) -- np.linalg.solve on a triangular system is the same math
        M[:nf, nf:] = B
        C = np.triu(M[nf:, nf:]) - np.triu(B.T @ B)
        M[nf:, nf:] = C
    if nf >= 2:
        e2 = math.frexp(R[nf - 2, nf - 2])[1]; e1 = math.frexp(R[nf - 1, nf - 1])[1]
        return (e2 - e1 < 12), M
    e1 = math.frexp(R[0, 0])[1]
    return (e1 > -12), M


def preconditioned_conjugate_gradient(A, b, blocks, max_iterations=500, min_iterations=1, epsilon_rel=1e-3, epsilon_abs=1e-3):
    """preconditionedConjugateGradient (linear/ConjugateGradientSolver.h:106-169) from x0 = 0 with a block-Jacobi
    preconditioner M = L L^T, L = Cholesky factors of the diagonal blocks `blocks` = [(start, stop)] of A
    (linear/Preconditioner.cpp BlockJacobiPreconditioner): split form r = L^-1 (b - A x), p = L^-T r; stop when
    |r|^2 <= max(epsilon_abs, epsilon_rel^2 |r0|^2).  Returns (x, iterations, gamma0, gamma)."""
    Ls = [np.linalg.cholesky(A[a:b_, a:b_]) for a, b_ in blocks]

    def left(v):
        out = np.zeros_like(v)
        for (a, b_), L in zip(blocks, Ls):
            out[a:b_] = np.linalg.solve(L, v[a:b_])
        return out

    def right(v):
        out = np.zeros_like(v)
        for (a, b_), L in zip(blocks, Ls):
            out[a:b_] = np.linalg.solve(L.T, v[a:b_])
        return out

    x = np.zeros_like(b)
    r = left(b - A @ x)
    pdir = right(r)
    gamma = float(r @ r); gamma0 = gamma
    threshold = max(epsilon_abs, epsilon_rel * epsilon_rel * gamma)
    k = 1
    while k <= max_iterations and (gamma > threshold or k <= min_iterations):
        q1 = A @ pdir
        alpha = gamma / float(pdir @ q1)
        x = x + alpha * pdir
        r = r - alpha * left(q1)
        prev = gamma
        gamma = float(r @ r)
        pdir = right(r) + (gamma / prev) * pdir
        k += 1
    return x, k - 1, gamma0, gamma


def solve_damped(p: Problem, values, lam, diagonal_damping=False, min_diag=1e-6, max_diag=1e32, pcg=None):
    """One solve of LevenbergMarquardtOptimizer::tryLambda (LM.cpp:146-160):
    buildDampedSystem (internal/LevenbergMarquardtState.h:125-156: one prior per variable, sigma =
    1/sqrt(lambda), A = I or diag(sqrt(clamp(hessianDiagonal))) ) then
    GaussianFactorGraph::optimize with the Schur ordering (points first; timing/timeSFMBAL.h:74-83):
    every POINT3 is a clique with 3 frontals eliminated by choleskyPartial
    (HessianFactor.cpp:459-487), the separators are summed into the reduced system, which is then
    eliminated as one dense root clique; back-substitution x_F = R^-1 (d - S x_S)
    (linearAlgorithms-inst.h:49-155).  Returns (status, delta, H, g, lin); status 1 =
    IndeterminantLinearSystemException."""
    H, g, lin = hessian_dense(p, values)
    n = H.shape[0]
    if diagonal_damping:
        dd = np.sqrt(np.minimum(np.maximum(np.diag(H).copy(), min_diag), max_diag))  # LM.cpp:293-299
        Hd = H + np.diag(lam * dd * dd)
    else:
        Hd = H + lam * np.eye(n)
    doff = p.dim_offsets()
    pts = [v for v in range(p.n_vars) if p.var_type[v] == VAR_POINT3]
    rest = [v for v in range(p.n_vars) if p.var_type[v] != VAR_POINT3]
    idx_rest = np.concatenate([np.arange(doff[v], doff[v + 1]) for v in rest]) if rest else np.zeros(0, int)
    S = Hd[np.ix_(idx_rest, idx_rest)].copy(); gr = g[idx_rest].copy()
    elim = []
    for v in pts:
        ip = np.arange(doff[v], doff[v + 1])
        W = Hd[np.ix_(idx_rest, ip)]
        nz = np.where(np.abs(W).sum(1) > 0)[0]
        sep = np.unique(nz)                              # separator entries of this clique
        m = 3 + sep.size + 1
        aug = np.zeros((m, m))
        aug[:3, :3] = Hd[np.ix_(ip, ip)]; aug[:3, 3:3 + sep.size] = W[sep].T; aug[:3, -1] = g[ip]
        ok, aug = cholesky_partial(aug, 3)
        if not ok:
            return 1, None, H, g, lin
        Rp = np.triu(aug[:3, :3]); Sp = aug[:3, 3:3 + sep.size]; dp = aug[:3, -1]
        S[np.ix_(sep, sep)] -= Sp.T @ Sp                 # separator HessianFactor -> parent
        gr[sep] -= Sp.T @ dp
        elim.append((ip, sep, Rp, Sp, dp))
    delta = np.zeros(n)
    if idx_rest.size and pcg is not None:
        # NonlinearOptimizerParams::Iterative: PCG with block Jacobi on the Schur complement of the landmarks (the GPU path
        # applies it implicitly).  pcg = dict(max_iterations, min_iterations, epsilon_rel, epsilon_abs); the iteration
        # count is left in pcg["iterations"].
        blocks = []; o = 0
        for v in rest:
            d = int(doff[v + 1] - doff[v]); blocks.append((o, o + d)); o += d
        try:
            xr, its, g0, g1 = preconditioned_conjugate_gradient(S, gr, blocks, pcg.get("max_iterations", 500), pcg.get("min_iterations", 1),
                                                                pcg.get("epsilon_rel", 1e-3), pcg.get("epsilon_abs", 1e-3))
        except np.linalg.LinAlgError:
            return 1, None, H, g, lin
        pcg["iterations"] = its; pcg["gamma0"] = g0; pcg["gamma"] = g1
        if not np.all(np.isfinite(xr)):
            return 1, None, H, g, lin
        delta[idx_rest] = xr
    elif idx_rest.size:
        m = idx_rest.size
        aug = np.zeros((m + 1, m + 1)); aug[:m, :m] = S; aug[:m, m] = gr
        ok, aug = cholesky_partial(aug, m)
        if not ok:
            return 1, None, H, g, lin
        Rr = np.triu(aug[:m, :m]); dr = aug[:m, m]
        xr = np.linalg.solve(Rr, dr)
        if not np.all(np.isfinite(xr)):
            return 1, None, H, g, lin
        delta[idx_rest] = xr
    else:
        xr = np.zeros(0)
    for ip, sep, Rp, Sp, dp in elim:
        delta[ip] = np.linalg.solve(Rp, dp - Sp @ xr[sep])
    return 0, delta, H, g, lin


def linear_error(p: Problem, lin, delta):
    """GaussianFactorGraph::error(delta) = sum 0.5*||A delta - b||^2 on the UNDAMPED graph
    (linear/GaussianFactorGraph.cpp:71-78, JacobianFactor.cpp:486-491)."""
    doff = p.dim_offsets(); e = 0.0
    for vids, As, b in _factor_blocks(p, lin):
        r = -b.copy()
        for i, vi in enumerate(vids):
            r += As[i] @ delta[doff[vi]:doff[vi + 1]]
        e += 0.5 * float(r @ r)
    for cams, G, gg, f in (lin.get("smart", ()) if isinstance(lin, dict) else ()):   # HessianFactor::error: 0.5 (f - 2 x^T g + x^T G x)
        x = np.concatenate([delta[doff[c]:doff[c + 1]] for c in cams])
        e += 0.5 * (f - 2.0 * float(x @ gg) + float(x @ G @ x))
    return e


def retract(p: Problem, values, delta):
    """Values::retract (nonlinear/Values.cpp:52-63): Pose3 T*Expmap(xi); PinholeCamera pose retract +
    Cal3Bundler::retract (PinholeCamera.h:199-205, Cal3Bundler.h:145-147); Point3 p + d."""
    values = np.asarray(values, np.float64); out = values.copy()
    off = p.val_offsets(); doff = p.dim_offsets()
    for t, st, dm in ((VAR_POSE3, 12, 6), (VAR_SFM_CAMERA, 17, 9), (VAR_POINT3, 3, 3), (VAR_POSE2, 3, 3)):
        ids = np.where(p.var_type == t)[0]
        if not ids.size:
            continue
        x = _gather(values, off, ids, st); d = _gather(delta, doff, ids, dm)
        if t == VAR_POINT3:
            y = x + d
        elif t == VAR_POSE2:
            y = pose2_retract(x, d)
        else:
            y = x.copy(); y[:, :12] = pose_retract(x[:, :12], d[:, :6])
            if t == VAR_SFM_CAMERA:
                y[:, 12:15] = x[:, 12:15] + d[:, 6:9]
        out[off[ids][:, None] + np.arange(st)[None, :]] = y
    return out


# ------------------------------------------------------------------------------------------------
# the LM loop   (nonlinear/LevenbergMarquardtOptimizer.cpp, NonlinearOptimizer.cpp)
# ------------------------------------------------------------------------------------------------
def check_convergence(rel_tol, abs_tol, err_tol, current_error, new_error):
    """checkConvergence (nonlinear/NonlinearOptimizer.cpp:182-231)."""
    if new_error <= err_tol:
        return True
    absolute_decrease = current_error - new_error
    relative_decrease = absolute_decrease / current_error
    return bool((rel_tol and (relative_decrease <= rel_tol)) or (absolute_decrease <= abs_tol))


def lm_optimize(p: Problem, values0, params, max_trace=100000):
    """LevenbergMarquardtOptimizer::optimize = defaultOptimize (NonlinearOptimizer.cpp:62-117) around
    iterate()/tryLambda() (LM.cpp:121-308) with the lambda policy of
    internal/LevenbergMarquardtState.h:70-94.  Returns dict(values, trace rows (inner, error, lambda),
    iterations)."""
    values = np.asarray(values0, np.float64).copy()
    err = error(p, values)
    lam, factor = params.lambdaInitial, params.lambdaFactor
    iterations, inner = 0, 0
    trace = [(inner, err, lam)]
    if err <= params.errorTol or iterations >= params.maxIterations:
        return dict(values=values, trace=np.array(trace), iterations=iterations)
    new_error = err
    while True:
        current_error = new_error
        # ---- iterate(): linearize once, then try lambdas --------------------------------------
        while True:
            status, delta, H, g, lin = solve_damped(p, values, lam, params.diagonalDamping,
                                                    params.minDiagonal, params.maxDiagonal)
            step_ok = False; stop = False; model_fidelity = 0.0
            trial_err = math.inf; trial = None
            if status == 0:
                old_lin = linear_error(p, lin, np.zeros_like(delta)); new_lin = linear_error(p, lin, delta)
                lin_change = old_lin - new_lin
                if lin_change >= 0:
                    trial = retract(p, values, delta)
                    trial_err = error(p, trial)
                    cost_change = err - trial_err
                    if lin_change > EPS * old_lin:
                        model_fidelity = cost_change / lin_change
                        step_ok = model_fidelity > params.minModelFidelity
                    if abs(cost_change) < params.relativeErrorTol * err:
                        stop = True
            if step_ok:
                # decreaseLambda (LMState.h:81-94)
                if params.useFixedLambdaFactor:
                    lam = lam / factor
                else:
                    lam = lam * max(1.0 / 3.0, 1.0 - (2.0 * model_fidelity - 1.0) ** 3)
                    factor = 2.0 * factor
                lam = max(params.lambdaLowerBound, lam)
                values, err = trial, trial_err
                iterations += 1; inner += 1
                break
            elif not stop:
                lam *= factor; inner += 1                   # increaseLambda (LMState.h:70-76)
                if not params.useFixedLambdaFactor:
                    factor *= 2.0
                if lam >= params.lambdaUpperBound:
                    break
            else:
                break
        new_error = err
        if len(trace) < max_trace:
            trace.append((inner, err, lam))
        if not (iterations < params.maxIterations and
                not check_convergence(params.relativeErrorTol, params.absoluteErrorTol, params.errorTol,
                                      current_error, new_error) and math.isfinite(current_error)):
            break
    return dict(values=values, trace=np.array(trace), iterations=iterations)
