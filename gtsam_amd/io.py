"""Host-side dataset readers: the two wire formats either side of the hot path.

Restated from the reference's loaders so that a file parsed here gives the same numbers the reference's own
loader hands to its optimizer:
  * read_bal     <-> SfmData::FromBalFile   gtsam/sfm/SfmData.cpp:189-246  (numbers go through float32 temporaries;
                                            Rodrigues rotation, openGL2gtsam :79-85, measurement (u, -v))
  * read_g2o3d   <-> load3D / readG2o(is3D) gtsam/slam/dataset.cpp:738-944  (VERTEX3 / VERTEX_SE3:QUAT /
                                            EDGE3 (roll pitch yaw) / EDGE_SE3:QUAT incl. the t,R -> R,t information
                                            reshuffle :848-853; Gaussian::Information's smart down-casting
                                            gtsam/linear/NoiseModel.cpp:97-110,283-308,624-633)
  * read_2d      <-> load2D                 gtsam/slam/dataset.cpp:179-330
and the writers of results:
  * write_bal    <-> writeBAL               gtsam/sfm/SfmData.cpp:249-327
  * write_g2o    <-> writeG2o               gtsam/slam/dataset.cpp:636-735
BAL files are large (Ladybug-1723: 2.7 M numbers, Venice-1778: 20 M): read_bal / write_bal go through the native parser of
the C ABI (gtg_io_read_bal / gtg_io_write_bal, gtsam_amd/csrc/io.cpp); read_bal_py is the token-by-token restatement the
native one is tested against.  Pure host bookkeeping; nothing here runs per iteration.
"""
from __future__ import annotations

import numpy as np

from .problem import NOISE_DIAGONAL, NOISE_GAUSSIAN, NOISE_ISOTROPIC, NOISE_UNIT


def _rodrigues(w):
    """Rot3::Rodrigues = SO3::Expmap (geometry/SO3.cpp:50-88)."""
    w = np.asarray(w, np.float64)
    theta2 = float(w @ w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if theta2 <= np.finfo(np.float64).eps:
        return np.eye(3) + W
    theta = np.sqrt(theta2)
    K = W / theta
    s2 = np.sin(theta / 2.0)
    return np.eye(3) + np.sin(theta) * K + (2.0 * s2 * s2) * (K @ K)


def read_bal(path):
    """-> (cams17, pts3, obs_cam, obs_pt, obs_z); observations ordered by track then file order.  Native parser
    (gtg_io_read_bal); raises like SfmData::FromBalFile when the file is missing."""
    import ctypes as C
    from . import lib as L
    lib = L.load()
    nc, npt, nobs = C.c_int64(), C.c_int64(), C.c_int64()
    if lib.gtg_io_bal_sizes(str(path).encode(), C.byref(nc), C.byref(npt), C.byref(nobs)) != 0:
        raise RuntimeError(lib.gtg_io_last_error().decode())
    cams = np.zeros((nc.value, 17)); pts = np.zeros((npt.value, 3))
    oc = np.zeros(nobs.value, np.int32); op = np.zeros(nobs.value, np.int32); oz = np.zeros((nobs.value, 2))
    if lib.gtg_io_read_bal(str(path).encode(), nc.value, npt.value, nobs.value, cams.ctypes.data, pts.ctypes.data,
                           oc.ctypes.data, op.ctypes.data, oz.ctypes.data) != 0:
        raise RuntimeError(lib.gtg_io_last_error().decode())
    return cams, pts, oc, op, oz


def write_bal(path, cams, pts, obs_cam, obs_pt, obs_z):
    """writeBAL (sfm/SfmData.cpp:249-327) of packed SfmCameras (n,17), points (n,3) and track-ordered observations."""
    from . import lib as L
    lib = L.load()
    cams = np.ascontiguousarray(cams, np.float64).reshape(-1, 17); pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
    oz = np.ascontiguousarray(obs_z, np.float64).reshape(-1, 2)
    if lib.gtg_io_write_bal(str(path).encode(), cams.shape[0], pts.shape[0], oc.size, cams.ctypes.data, pts.ctypes.data,
                            oc.ctypes.data, op.ctypes.data, oz.ctypes.data) != 0:
        raise RuntimeError(lib.gtg_io_last_error().decode())


def read_bal_py(path):
    """Token-by-token restatement of SfmData::FromBalFile (the check of the native parser; slow on large files)."""
    toks = open(path).read().split()
    n_cam, n_pt, n_obs = int(toks[0]), int(toks[1]), int(toks[2])
    pos = 3
    f32 = lambda s: float(np.float32(s))      # noqa: E731  (`float u; is >> u`)
    oc = np.zeros(n_obs, np.int64); op = np.zeros(n_obs, np.int64); oz = np.zeros((n_obs, 2))
    for k in range(n_obs):
        oc[k] = int(toks[pos]); op[k] = int(toks[pos + 1])
        oz[k] = (f32(toks[pos + 2]), -f32(toks[pos + 3]))
        pos += 4
    cams = np.zeros((n_cam, 17))
    R90 = np.diag([1.0, -1.0, -1.0])
    for i in range(n_cam):
        v = [f32(t) for t in toks[pos:pos + 9]]; pos += 9
        R = _rodrigues(v[0:3])
        wRc = R.T @ R90                                       # openGL2gtsam: (R.inverse()).compose(R90)
        t = R.T @ (-np.array(v[3:6]))                         # R.unrotate(-t)
        cams[i, :9] = wRc.reshape(-1); cams[i, 9:12] = t
        cams[i, 12:15] = v[6:9]                               # Cal3Bundler(f, k1, k2), u0 = v0 = 0
    pts = np.array([[f32(t) for t in toks[pos + 3 * j:pos + 3 * j + 3]] for j in range(n_pt)]).reshape(n_pt, 3)
    order = np.argsort(op, kind="stable")                     # tracks[j].measurements in file order
    return cams, pts, oc[order].astype(np.int32), op[order].astype(np.int32), oz[order]


def _ypr(yaw, pitch, roll):
    """Rot3::Ypr(y,p,r) = RzRyRx(r, p, y) (geometry/Rot3.h)."""
    cx, sx, cy, sy, cz, sz = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _quat(x, y, z, w):
    """operator>>(Quaternion) normalises (dataset.cpp:738-744); Eigen quaternion -> rotation matrix."""
    n = np.sqrt(w * w + x * x + y * y + z * z); x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def information_to_noise(m):
    """noiseModel::Gaussian::Information(m, smart=true): diagonal -> Diagonal::Precisions -> Variances ->
    (all equal -> Isotropic::Variance -> (|var-1|<1e-9 -> Unit)); else R = upper Cholesky factor of m.
    Returns (kind, 36 parameters)."""
    out = np.zeros(36)
    if np.all(m == np.diag(np.diag(m))):
        var = 1.0 / np.diag(m)
        if np.all(var == var[0]):
            if abs(var[0] - 1.0) < 1e-9:
                return NOISE_UNIT, out
            out[0] = np.sqrt(var[0])
            return NOISE_ISOTROPIC, out
        out[:6] = np.sqrt(var)
        return NOISE_DIAGONAL, out
    out[:] = np.linalg.cholesky(m).T.reshape(-1)
    return NOISE_GAUSSIAN, out


def read_g2o3d_py(path):
    """-> dict(v1, v2, z [n,12], noise_kind, noise [n,36], vertex_keys, vertex_poses [m,12]) like the reference's
    load3D: vertices only when the file has VERTEX lines (it does *not* create missing ones, dataset.cpp:922-944)."""
    v1, v2, zs, nk, nd, vk, vp = [], [], [], [], [], [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        tag = t[0]
        if tag in ("VERTEX3", "VERTEX_SE3:QUAT"):
            x, y, z = (float(a) for a in t[2:5])
            R = _ypr(float(t[7]), float(t[6]), float(t[5])) if tag == "VERTEX3" else _quat(*(float(a) for a in t[5:9]))
            vk.append(int(t[1])); vp.append(np.concatenate([R.reshape(-1), [x, y, z]]))
        elif tag in ("EDGE3", "EDGE_SE3:QUAT"):
            x, y, z = (float(a) for a in t[3:6])
            if tag == "EDGE3":
                R = _ypr(float(t[8]), float(t[7]), float(t[6])); rest = t[9:30]
            else:
                R = _quat(*(float(a) for a in t[6:10])); rest = t[10:31]
            m = np.zeros((6, 6)); k = 0
            for i in range(6):
                for j in range(i, 6):
                    m[i, j] = m[j, i] = float(rest[k]); k += 1
            if tag == "EDGE_SE3:QUAT":                        # g2o stores t,R order (dataset.cpp:848-853)
                mg = np.zeros((6, 6))
                mg[:3, :3] = m[3:, 3:]; mg[3:, 3:] = m[:3, :3]; mg[3:, :3] = m[:3, 3:]; mg[:3, 3:] = m[3:, :3]
                m = mg
            kind, params = information_to_noise(m)
            v1.append(int(t[1])); v2.append(int(t[2])); zs.append(np.concatenate([R.reshape(-1), [x, y, z]]))
            nk.append(kind); nd.append(params)
    order = np.argsort(vk, kind="stable") if vk else np.zeros(0, int)
    return dict(v1=np.array(v1, np.int64), v2=np.array(v2, np.int64), z=np.array(zs).reshape(-1, 12),
                noise_kind=np.array(nk, np.int32), noise=np.array(nd).reshape(-1, 36),
                vertex_keys=np.array(vk, np.int64)[order], vertex_poses=np.array(vp).reshape(-1, 12)[order])


def _gaussian3_to_noise(M, covariance):
    """noiseModel::Gaussian::{Information, Covariance}(M, smart=true) for a 3x3 matrix (linear/NoiseModel.cpp:83-131): a
    diagonal matrix becomes Diagonal::Variances (-> Isotropic -> Unit when all equal / equal to 1), otherwise the upper
    Cholesky factor R of the information matrix.  Returns (kind, 9 parameters: sigma | sigmas[3] | R row-major)."""
    out = np.zeros(9)
    if np.all(M == np.diag(np.diag(M))):
        var = np.diag(M) if covariance else 1.0 / np.diag(M)
        if np.all(var == var[0]):
            if abs(var[0] - 1.0) < 1e-9:
                return NOISE_UNIT, out
            out[0] = np.sqrt(var[0])
            return NOISE_ISOTROPIC, out
        out[:3] = np.sqrt(var)
        return NOISE_DIAGONAL, out
    info = np.linalg.inv(M) if covariance else M
    out[:] = np.linalg.cholesky(info).T.reshape(-1)
    return NOISE_GAUSSIAN, out


NOISE_FORMAT_G2O, NOISE_FORMAT_TORO, NOISE_FORMAT_GRAPH, NOISE_FORMAT_COV, NOISE_FORMAT_AUTO = "G2O", "TORO", "GRAPH", "COV", "AUTO"


def _noise_matrix3(v, noise_format):
    """createNoiseModel's matrix (slam/dataset.cpp:215-262): the 6 numbers of an edge line -> (3x3 matrix, is_covariance).
    G2O / COV: upper-triangular order [v0 v1 v2; . v3 v4; . . v5]; TORO / GRAPH: [v0 v1 v4; . v2 v5; . . v3].
    G2O and TORO store the INFORMATION matrix, GRAPH and COV the covariance.  AUTO guesses GRAPH or COV from the zero pattern."""
    if noise_format == NOISE_FORMAT_AUTO:
        if v[0] != 0 and v[1] == 0 and v[2] != 0 and v[3] != 0 and v[4] == 0 and v[5] == 0:
            noise_format = NOISE_FORMAT_GRAPH
        elif v[0] != 0 and v[1] == 0 and v[2] == 0 and v[3] != 0 and v[4] == 0 and v[5] != 0:
            noise_format = NOISE_FORMAT_COV
        else:
            raise ValueError("load2D: unrecognized covariance matrix format in dataset file. Please specify the noise format.")
    if noise_format in (NOISE_FORMAT_G2O, NOISE_FORMAT_COV):
        if v[0] == 0 or v[3] == 0 or v[5] == 0:
            raise ValueError("load2D::readNoiseModel looks like this is not G2O matrix order")
        M = np.array([[v[0], v[1], v[2]], [v[1], v[3], v[4]], [v[2], v[4], v[5]]])
    elif noise_format in (NOISE_FORMAT_TORO, NOISE_FORMAT_GRAPH):
        if v[0] == 0 or v[2] == 0 or v[3] == 0:
            raise ValueError("load2D::readNoiseModel looks like this is not TORO matrix order")
        M = np.array([[v[0], v[1], v[4]], [v[1], v[2], v[5]], [v[4], v[5], v[3]]])
    else:
        raise ValueError("load2D: invalid noise format")
    return M, noise_format in (NOISE_FORMAT_GRAPH, NOISE_FORMAT_COV)


def read_g2o(path, is_3d=False):
    """readG2o (slam/dataset.cpp:621-633): a 3-D file goes to load3D, a 2-D file to load2D with NoiseFormatG2O -- the six
    numbers of an EDGE_SE2 line are the upper triangle of the INFORMATION matrix (what write_g2o writes)."""
    return read_g2o3d(path) if is_3d else read_2d(path, noise_format=NOISE_FORMAT_G2O)


# ---- the native readers / writer (gtsam_amd/csrc/io_g2o.cpp behind the C ABI: gtg_io_g2o_sizes / gtg_io_read_g2o / gtg_io_write_g2o);
# the *_py functions of this module are the token-by-token restatements they are checked against (tests/test_io.py)
_NOISE_FORMAT_ID = {NOISE_FORMAT_AUTO: 0, NOISE_FORMAT_G2O: 1, NOISE_FORMAT_TORO: 2, NOISE_FORMAT_GRAPH: 3, NOISE_FORMAT_COV: 4}


def _read_g2o_native(path, is_3d, noise_format):
    import ctypes as C
    from .lib import load
    lib = load()
    if noise_format not in _NOISE_FORMAT_ID:
        raise ValueError("load2D: invalid noise format")
    fmt = _NOISE_FORMAT_ID[noise_format]
    ne, nv = C.c_int64(), C.c_int64()
    if lib.gtg_io_g2o_sizes(str(path).encode(), int(is_3d), fmt, C.byref(ne), C.byref(nv)) != 0:
        raise ValueError(lib.gtg_io_last_error().decode())
    ne, nv = ne.value, nv.value
    dz, dn = (12, 36) if is_3d else (3, 9)
    v1 = np.zeros(ne, np.int64); v2 = np.zeros(ne, np.int64); z = np.zeros((ne, dz)); nk = np.zeros(ne, np.int32); nd = np.zeros((ne, dn))
    vk = np.zeros(nv, np.int64); vp = np.zeros((nv, dz))
    if lib.gtg_io_read_g2o(str(path).encode(), int(is_3d), fmt, ne, nv, v1.ctypes.data, v2.ctypes.data, z.ctypes.data, nk.ctypes.data,
                           nd.ctypes.data, vk.ctypes.data, vp.ctypes.data) != 0:
        raise ValueError(lib.gtg_io_last_error().decode())
    return dict(v1=v1, v2=v2, z=z, noise_kind=nk, noise=nd, vertex_keys=vk, vertex_poses=vp)


def read_g2o3d(path):
    """load3D / readG2o(path, true) (slam/dataset.cpp:738-944) -> dict(v1, v2, z [n,12], noise_kind, noise [n,36], vertex_keys,
    vertex_poses [m,12]); vertices only where the file has VERTEX lines.  Native parser (csrc/io_g2o.cpp)."""
    return _read_g2o_native(path, True, NOISE_FORMAT_AUTO)


def read_2d(path, noise_format=NOISE_FORMAT_AUTO):
    """load2D (slam/dataset.cpp:179-330, 505-570; `noise_format` = load2D's parameter of that name: "AUTO" | "G2O" | "TORO" | "GRAPH" |
    "COV") -> dict(v1, v2, z [n,3], noise_kind, noise [n,9], vertex_keys, vertex_poses [m,3]), vertices a pure odometry file does not
    list created along the chain.  Native parser (csrc/io_g2o.cpp)."""
    return _read_g2o_native(path, False, noise_format)


def write_g2o(path, d, vertex_keys=None, vertex_poses=None, full_precision=False):
    """writeG2o (slam/dataset.cpp:636-735) through the native writer (csrc/io_g2o.cpp); arguments as write_g2o_py."""
    from .lib import load
    lib = load()
    vk = np.ascontiguousarray(d["vertex_keys"] if vertex_keys is None else vertex_keys, np.int64)
    is3d = d["z"].shape[1] == 12
    vp = np.ascontiguousarray(d["vertex_poses"] if vertex_poses is None else vertex_poses, np.float64).reshape(len(vk), 12 if is3d else 3)
    v1 = np.ascontiguousarray(d["v1"], np.int64); v2 = np.ascontiguousarray(d["v2"], np.int64)
    z = np.ascontiguousarray(d["z"], np.float64); nk = np.ascontiguousarray(d["noise_kind"], np.int32)
    nd = np.ascontiguousarray(d["noise"], np.float64).reshape(len(v1), 36 if is3d else 9)
    if lib.gtg_io_write_g2o(str(path).encode(), int(is3d), len(v1), v1.ctypes.data, v2.ctypes.data, z.ctypes.data, nk.ctypes.data, nd.ctypes.data,
                            len(vk), vk.ctypes.data, vp.ctypes.data, int(full_precision)) != 0:
        raise RuntimeError(lib.gtg_io_last_error().decode())


def read_2d_py(path, noise_format=NOISE_FORMAT_AUTO):
    """load2D (slam/dataset.cpp:179-330, defaults: maxIndex 0, smart noise, NoiseFormatAUTO, no kernel; `noise_format` =
    load2D's parameter of that name: "AUTO" | "G2O" | "TORO" | "GRAPH" | "COV"): VERTEX2 /
    VERTEX_SE2 / VERTEX lines -> initial Pose2 (x, y, theta); EDGE2 / EDGE / EDGE_SE2 / ODOMETRY lines ->
    BetweenFactor<Pose2> with the 6 noise numbers interpreted by their zero pattern (dataset.cpp:218-232): GRAPH order
    (covariance, [v0 v1 v4; v1 v2 v5; v4 v5 v3]) or COV order (covariance, [v0 v1 v2; v1 v3 v4; v2 v4 v5]).
    -> dict(v1, v2, z [n,3], noise_kind, noise [n,9], vertex_keys, vertex_poses [m,3])."""
    import math
    v1, v2, zs, nk, nd = [], [], [], [], []
    poses = {}          # key -> (x, y, cos, sin): first pass = the VERTEX lines (dataset.cpp:511-522)
    lines = [line.split() for line in open(path)]
    for t in lines:
        if t and t[0] in ("VERTEX2", "VERTEX_SE2", "VERTEX"):
            if int(t[1]) in poses:     # Values::insert throws ValuesKeyAlreadyExists (nonlinear/Values.cpp:140-145)
                raise ValueError(f"load2D: vertex {int(t[1])} appears twice (ValuesKeyAlreadyExists in the reference)")
            poses[int(t[1])] = (float(t[2]), float(t[3]), math.cos(float(t[4])), math.sin(float(t[4])))
    for t in lines:
        if not t:
            continue
        tag = t[0]
        if tag in ("EDGE2", "EDGE", "EDGE_SE2", "ODOMETRY"):
            v = [float(a) for a in t[6:12]]
            M, is_cov = _noise_matrix3(v, noise_format)
            kind, params = _gaussian3_to_noise(M, covariance=is_cov)
            k1, k2 = int(t[1]), int(t[2])
            zx, zy, zt = float(t[3]), float(t[4]), float(t[5])
            v1.append(k1); v2.append(k2); zs.append([zx, zy, zt])
            nk.append(kind); nd.append(params)
            # vertices a pure odometry file does not list: identity for key1, key1's pose * measurement for key2
            # (dataset.cpp:541-546; Pose2 product: Rot2 through fromCosSin, Pose2.h:131-133)
            if k1 not in poses:
                poses[k1] = (0.0, 0.0, 1.0, 0.0)
            if k2 not in poses:
                x, y, c, s_ = poses[k1]
                zc, zsn = math.cos(zt), math.sin(zt)
                nc, ns = c * zc - s_ * zsn, s_ * zc + c * zsn
                scale = nc * nc + ns * ns
                if abs(scale - 1.0) > 1e-10:
                    scale = 1.0 / math.sqrt(scale); nc *= scale; ns *= scale
                poses[k2] = (x + (c * zx + -s_ * zy), y + (s_ * zx + c * zy), nc, ns)
    # Pose2(x, y, yaw) keeps (cos, sin): theta() = atan2(sin yaw, cos yaw) is the file's angle wrapped into (-pi, pi]
    zs = np.array(zs, np.float64).reshape(-1, 3); zs[:, 2] = np.arctan2(np.sin(zs[:, 2]), np.cos(zs[:, 2]))
    vk = sorted(poses)                                                                        # Values iterate by key
    vp = np.array([[poses[k][0], poses[k][1], math.atan2(poses[k][3], poses[k][2])] for k in vk], np.float64).reshape(-1, 3)
    order = np.arange(len(vk))
    return dict(v1=np.array(v1, np.int64), v2=np.array(v2, np.int64), z=zs,
                noise_kind=np.array(nk, np.int32), noise=np.array(nd, np.float64).reshape(-1, 9),
                vertex_keys=np.array(vk, np.int64)[order], vertex_poses=vp[order])


def _quaternion(R):
    """Rot3::toQuaternion = Eigen::Quaternion(Matrix3) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl):
    -> (x, y, z, w)."""
    R = np.asarray(R, np.float64).reshape(3, 3)
    q = np.zeros(4)                                           # x y z w
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t
        q[0] = (R[2, 1] - R[1, 2]) * t; q[1] = (R[0, 2] - R[2, 0]) * t; q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3; k = (j + 1) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0); q[i] = 0.5 * t; t = 0.5 / t
        q[3] = (R[k, j] - R[j, k]) * t; q[j] = (R[j, i] + R[i, j]) * t; q[k] = (R[k, i] + R[i, k]) * t
    return q


def _information(kind, params, dim):
    """R^T R of a noise-table row (kind, parameters as read_g2o3d / read_2d return them)."""
    params = np.asarray(params, np.float64)
    if kind == NOISE_UNIT:
        return np.eye(dim)
    if kind == NOISE_ISOTROPIC:
        return np.eye(dim) / (params[0] * params[0])
    if kind == NOISE_DIAGONAL:
        return np.diag(1.0 / (params[:dim] * params[:dim]))
    R = params[:dim * dim].reshape(dim, dim)
    return R.T @ R


def write_g2o_py(path, d, vertex_keys=None, vertex_poses=None, full_precision=False):
    """writeG2o (slam/dataset.cpp:636-735): VERTEX_SE2 / VERTEX_SE3:QUAT lines of the estimate, then EDGE_SE2 /
    EDGE_SE3:QUAT lines of the BetweenFactors with the upper triangle of their information matrix (EDGE_SE3:QUAT in g2o's
    t,R block order), numbers at the stream's default precision.  `d` is what read_2d / read_g2o3d return; the estimate
    defaults to d's vertices (pass the optimised poses to write a result).  full_precision: shortest round-trip digits instead of the
    stream's six (not what writeG2o does: for files that carry a problem to another program without rounding it)."""
    vk = d["vertex_keys"] if vertex_keys is None else np.asarray(vertex_keys)
    vp = d["vertex_poses"] if vertex_poses is None else np.asarray(vertex_poses, np.float64)
    is3d = d["z"].shape[1] == 12
    g = (lambda x: repr(float(x))) if full_precision else (lambda x: f"{float(x):g}")      # noqa: E731  (`stream << double`)
    with open(path, "w") as f:
        for k, p in zip(vk, vp.reshape(len(vk), 12 if is3d else 3)):
            if is3d:
                q = _quaternion(p[:9])
                f.write("VERTEX_SE3:QUAT " + " ".join([str(int(k))] + [g(x) for x in (p[9], p[10], p[11], q[0], q[1], q[2], q[3])]) + "\n")
            else:
                f.write("VERTEX_SE2 " + " ".join([str(int(k))] + [g(x) for x in p[:3]]) + "\n")
        for a, b, z, kind, params in zip(d["v1"], d["v2"], d["z"], d["noise_kind"], d["noise"]):
            if is3d:
                info = _information(int(kind), params, 6)
                ig = np.eye(6)
                ig[:3, :3] = info[3:, 3:]; ig[3:, 3:] = info[:3, :3]; ig[:3, 3:] = info[3:, :3]; ig[3:, :3] = info[:3, 3:]
                q = _quaternion(z[:9])
                nums = [z[9], z[10], z[11], q[0], q[1], q[2], q[3]] + [ig[i, j] for i in range(6) for j in range(i, 6)]
                f.write("EDGE_SE3:QUAT " + " ".join([str(int(a)), str(int(b))] + [g(x) for x in nums]) + "\n")
            else:
                info = _information(int(kind), params, 3)
                nums = [z[0], z[1], z[2]] + [info[i, j] for i in range(3) for j in range(i, 3)]
                f.write("EDGE_SE2 " + " ".join([str(int(a)), str(int(b))] + [g(x) for x in nums]) + "\n")
