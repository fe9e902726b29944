"""Property-based checks (hypothesis) of the group / chart identities the LM step relies on, on the oracle restatement and on
the device formulas compiled for the host: retract and local coordinates are inverse to each other for every value type of
the path (Pose3 = Expmap chart, PinholeCamera<Cal3Bundler>, Point3, Pose2 = first-order chart with wrapped angle), rotations
stay orthonormal, Logmap inverts Expmap.  Rotation angles stay below 3.0 rad: beyond 3.11 rad (trace + 1 < 1e-3) the
reference's SO3::Logmap switches to a first-order formula around pi (SO3.cpp:262-303) that is exact at pi (the literal tests of
test_reference_known_answers.py) but only ~1e-6 accurate next to it, and whose choice of the largest diagonal entry is decided
by rounding when two are equal; and the step's rotation is either zero or at least 1e-3 rad: Pose3::Expmap divides
(w x v - R (w x v)) by theta^2 (Pose3.cpp:176-183), so between the near-zero branch (theta^2 <= eps) and ~1e-3 rad two
correctly rounded implementations differ by eps |v| / theta (3e-11 at 1e-5 rad).  Both are properties of the reference's
algorithms, found by this very test, not identities.  The reference states the same identities in its Testable / Manifold
concept checks (GTSAM_CONCEPT_MANIFOLD_INST, base/Manifold.h) and testPose3 / testPose2 / testRot3."""
import ctypes as C
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import gtsam_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
_f = lambda lo, hi: st.floats(lo, hi, allow_nan=False, allow_infinity=False)  # noqa: E731
vec3 = st.tuples(_f(-1, 1), _f(-1, 1), _f(-1, 1)).map(np.array)


def _hm():
    path = os.path.join(ROOT, "tests", "_build", "libhostmath.so")
    if not os.path.exists(path):
        pytest.skip("tests/_build/libhostmath.so not built")
    return C.CDLL(path)


def _retract(hm, vt, x, d):
    x = np.ascontiguousarray(x[None], np.float64); d = np.ascontiguousarray(d[None], np.float64); y = np.zeros_like(x)
    hm.hm_retract(C.c_int(vt), C.c_long(1), P(x), P(d), P(y)); return y[0]


def _local(hm, vt, x, y, dim):
    x = np.ascontiguousarray(x[None], np.float64); y = np.ascontiguousarray(y[None], np.float64); d = np.zeros((1, dim))
    hm.hm_local(C.c_int(vt), C.c_long(1), P(x), P(y), P(d)); return d[0]


@settings(max_examples=200, deadline=None, derandomize=True, database=None)
@given(axis=vec3, angle=st.one_of(st.just(0.0), _f(1e-3, 3.0)), axis2=vec3, angle2=_f(0.0, 3.0), t=vec3, v=vec3)
def test_pose3_chart_identities(axis, angle, axis2, angle2, t, v):
    hm = _hm()
    if np.linalg.norm(axis) < 1e-3 or np.linalg.norm(axis2) < 1e-3:
        return
    w = axis / np.linalg.norm(axis) * angle
    R = O.so3_expmap(w[None])[0]
    assert np.abs(R @ R.T - np.eye(3)).max() <= 1e-14 and abs(np.linalg.det(R) - 1) <= 1e-14
    assert np.abs(O.so3_logmap(R[None])[0] - w).max() <= 1e-9
    x = np.concatenate([O.so3_expmap((axis2 / np.linalg.norm(axis2) * angle2)[None])[0].reshape(-1), 5 * t])
    d = np.concatenate([w, 3 * v])
    for name, y in (("oracle", O.pose_retract(x[None], d[None])[0]), ("device", _retract(hm, 0, x, d))):
        Ry = y[:9].reshape(3, 3)
        assert np.abs(Ry @ Ry.T - np.eye(3)).max() <= 1e-13, name
    y = _retract(hm, 0, x, d)
    assert np.abs(y - O.pose_retract(x[None], d[None])[0]).max() <= 1e-12 * max(1.0, np.abs(y).max())
    back = _local(hm, 0, x, y, 6)
    scale = max(1.0, np.abs(y).max())
    assert np.abs(back - O.pose_local(x[None], y[None])[0]).max() <= 1e-8 * scale
    assert np.abs(back - d).max() <= 1e-7 * scale                                             # local(retract(x, d)) = d
    assert np.abs(_retract(hm, 0, x, back) - y).max() <= 1e-8 * scale                        # retract(x, local(x, y)) = y


@settings(max_examples=200, deadline=None, derandomize=True, database=None)
@given(x=st.tuples(_f(-50, 50), _f(-50, 50), _f(-np.pi, np.pi)).map(np.array), d=st.tuples(_f(-3, 3), _f(-3, 3), _f(-3.0, 3.0)).map(np.array))
def test_pose2_chart_identities(x, d):
    hm = _hm()
    y = _retract(hm, 3, x, d)
    assert np.abs(y - O.pose2_retract(x[None], d[None])[0]).max() <= 1e-12 * max(1.0, np.abs(y).max())
    assert -np.pi - 1e-12 <= y[2] <= np.pi + 1e-12                                             # theta() is wrapped
    back = _local(hm, 3, x, y, 3)
    assert np.abs(back[:2] - d[:2]).max() <= 1e-9 * max(1.0, np.abs(y).max()) and abs(np.angle(np.exp(1j * (back[2] - d[2])))) <= 1e-9
    y2 = _retract(hm, 3, x, back)
    assert np.abs(y2[:2] - y[:2]).max() <= 1e-9 * max(1.0, np.abs(y).max()) and abs(np.angle(np.exp(1j * (y2[2] - y[2])))) <= 1e-9


@settings(max_examples=200, deadline=None, derandomize=True, database=None)
@given(axis=vec3, angle=_f(0.0, 3.0), t=vec3, f=_f(300, 1500), k=st.tuples(_f(-0.05, 0.05), _f(-0.005, 0.005)), d=st.lists(_f(-0.5, 0.5), min_size=9, max_size=9), dp=vec3)
def test_camera_and_point_chart_identities(axis, angle, t, f, k, d, dp):
    hm = _hm()
    if np.linalg.norm(axis) < 1e-3:
        return
    cam = np.concatenate([O.so3_expmap((axis / np.linalg.norm(axis) * angle)[None])[0].reshape(-1), 10 * t, [f, k[0], k[1], 0.0, 0.0]])
    d = np.array(d)
    y = _retract(hm, 1, cam, d)
    assert np.abs(y[12:15] - (cam[12:15] + d[6:9])).max() <= 1e-12 and np.abs(y[15:] - cam[15:]).max() == 0      # Cal3Bundler::retract adds (f, k1, k2)
    assert np.abs(y[:12] - _retract(hm, 0, cam[:12], d[:6])).max() <= 1e-13 * max(1.0, np.abs(y[:12]).max())     # pose part = Pose3 retract
    assert np.abs(_local(hm, 1, cam, y, 9) - d).max() <= 1e-7 * max(1.0, np.abs(y[:12]).max())
    pt = 5 * t
    assert np.abs(_retract(hm, 2, pt, dp) - (pt + dp)).max() == 0 and np.abs(_local(hm, 2, pt, pt + dp, 3) - dp).max() <= 1e-15
