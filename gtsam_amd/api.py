"""Host mirror of the slice of GTSAM's wrapped API that feeds the LM hot path, so that user code and the parity
tests read like the reference's own (`python/gtsam` wrapper generated from `gtsam/*.i`):

    graph = NonlinearFactorGraph(); initial = Values()
    graph.add(GeneralSFMFactorCal3Bundler(uv, noiseModel.Isotropic.Sigma(2, 1.0), C(i), P(j)))
    graph.addPriorPose3(X(0), Pose3(), noiseModel.Diagonal.Variances(v))
    initial.insert(C(i), camera)
    result = LevenbergMarquardtOptimizer(graph, initial, params).optimize()

Containers and the extractor only (Key -> dense id in Values order, noise table, SoA factor tables); all arithmetic
of the path runs on the GPU through gtsam_amd.lib.  Reference anchors: Values `nonlinear/Values.h:74-79`,
`NonlinearFactorGraph::addPrior` `nonlinear/NonlinearFactorGraph.h:199-215`, Symbol `inference/Symbol.cpp:29-46`,
noise-model factories incl. their "smart" down-casting `linear/NoiseModel.cpp:83-131,283-308,624-633`.
"""
from __future__ import annotations

import numpy as np

from .optimizer import DeviceLevenbergMarquardt
from .params import LevenbergMarquardtParams
from .problem import (NOISE_DIAGONAL, NOISE_GAUSSIAN, NOISE_ISOTROPIC, NOISE_UNIT, STORAGE, TANGENT, VAR_POINT3, VAR_POSE2, VAR_POSE3,
                      VAR_SFM_CAMERA, Problem)


# ---- keys ---------------------------------------------------------------------------------------------------
def symbol(c: str, j: int) -> int:
    """Symbol(c, j): char << 56 | index (inference/Symbol.cpp:29-46)."""
    return (ord(c) << 56) | int(j)


class symbol_shorthand:
    C = staticmethod(lambda j: symbol("c", j)); P = staticmethod(lambda j: symbol("p", j))
    X = staticmethod(lambda j: symbol("x", j)); L = staticmethod(lambda j: symbol("l", j))


# ---- geometry value types (storage only) ----------------------------------------------------------------------
class Rot3:
    def __init__(self, R=None):
        self.R = np.eye(3) if R is None else np.asarray(R, np.float64).reshape(3, 3)

    def matrix(self): return self.R


def Point3(x=0.0, y=0.0, z=0.0):
    return np.array([x, y, z], np.float64)


def Point2(x=0.0, y=0.0):
    return np.array([x, y], np.float64)


class Pose3:
    def __init__(self, R: Rot3 | None = None, t=None):
        self.R = (R or Rot3()).R; self.t = np.zeros(3) if t is None else np.asarray(t, np.float64).reshape(3)

    def rotation(self): return Rot3(self.R)
    def translation(self): return self.t
    def packed(self): return np.concatenate([self.R.reshape(-1), self.t])

    @staticmethod
    def from_packed(p): return Pose3(Rot3(np.asarray(p[:9]).reshape(3, 3)), p[9:12])


class Pose2:
    """gtsam.Pose2(x, y, theta) (geometry/Pose2.h); packed as (x, y, theta)."""

    def __init__(self, x=0.0, y=0.0, theta=0.0):
        self.v = np.array([x, y, theta], np.float64)

    def x(self): return self.v[0]
    def y(self): return self.v[1]
    def theta(self): return self.v[2]
    def packed(self): return self.v.copy()

    @staticmethod
    def from_packed(p): return Pose2(p[0], p[1], p[2])


class Cal3Bundler:
    def __init__(self, f=1.0, k1=0.0, k2=0.0, u0=0.0, v0=0.0):
        self.v = np.array([f, k1, k2, u0, v0], np.float64)

    def fx(self): return self.v[0]
    def k1(self): return self.v[1]
    def k2(self): return self.v[2]


class Cal3_S2:
    def __init__(self, fx=1.0, fy=1.0, s=0.0, u0=0.0, v0=0.0):
        self.v = np.array([fx, fy, s, u0, v0], np.float64)


class Cal3DS2:
    """fx, fy, s, u0, v0 + radial (k1, k2) and tangential (p1, p2) distortion (geometry/Cal3DS2_Base.h:43-66)."""

    def __init__(self, fx=1.0, fy=1.0, s=0.0, u0=0.0, v0=0.0, k1=0.0, k2=0.0, p1=0.0, p2=0.0):
        self.v = np.array([fx, fy, s, u0, v0, k1, k2, p1, p2], np.float64)


class PinholeCameraCal3Bundler:
    """gtsam.PinholeCameraCal3Bundler = SfmCamera (sfm/SfmData.h:33)."""

    def __init__(self, pose: Pose3 | None = None, K: Cal3Bundler | None = None):
        self._pose = pose or Pose3(); self._K = K or Cal3Bundler()

    def pose(self): return self._pose
    def calibration(self): return self._K
    def packed(self): return np.concatenate([self._pose.packed(), self._K.v])

    @staticmethod
    def from_packed(p): return PinholeCameraCal3Bundler(Pose3.from_packed(p[:12]), Cal3Bundler(*p[12:17]))


# ---- noise models ------------------------------------------------------------------------------------------------
def _robust_weight(kind, k, d):
    a = abs(d)
    if kind == 1: return 1.0 / (1.0 + a / k)
    if kind == 2: return 1.0 if a <= k else k / a
    if kind == 3: return k * k / (k * k + d * d)
    if kind == 4: return (1.0 - d * d / (k * k)) ** 2 if a <= k else 0.0
    if kind == 5: return float(np.exp(-(d * d) / (k * k)))
    if kind == 6: return k ** 4 / (k * k + d * d) ** 2
    return 1.0


class _Noise:
    def __init__(self, kind, dim, params=()):
        self.kind, self._dim, self.params = kind, int(dim), np.asarray(params, np.float64).reshape(-1)

    robust = (0, 0.0)       # (ROBUST_*, parameter): set by noiseModel.Robust.Create

    def dim(self): return self._dim
    def key(self): return (self.kind, self._dim, self.params.tobytes(), self.robust)


class noiseModel:
    class Unit:
        @staticmethod
        def Create(dim): return _Noise(NOISE_UNIT, dim)

    class Isotropic:
        @staticmethod
        def Sigma(dim, sigma, smart=True):
            if smart and abs(sigma - 1.0) < 1e-9:
                return noiseModel.Unit.Create(dim)
            return _Noise(NOISE_ISOTROPIC, dim, [sigma])

        @staticmethod
        def Variance(dim, variance, smart=True):
            if smart and abs(variance - 1.0) < 1e-9:
                return noiseModel.Unit.Create(dim)
            return _Noise(NOISE_ISOTROPIC, dim, [np.sqrt(variance)])

    class Diagonal:
        @staticmethod
        def Sigmas(sigmas, smart=True):
            s = np.asarray(sigmas, np.float64).reshape(-1)
            if smart and s.size:
                if np.any(s < 1e-8):
                    raise ValueError("Constrained noise models (sigma < 1e-8) are outside the GPU path")
                if np.all(s == s[0]):
                    return noiseModel.Isotropic.Sigma(s.size, s[0], True)
            return _Noise(NOISE_DIAGONAL, s.size, s)

        @staticmethod
        def Variances(variances, smart=True):
            v = np.asarray(variances, np.float64).reshape(-1)
            if smart and np.all(v == v[0]):
                return noiseModel.Isotropic.Variance(v.size, v[0], True)
            return _Noise(NOISE_DIAGONAL, v.size, np.sqrt(v))

        @staticmethod
        def Precisions(precisions, smart=True):
            return noiseModel.Diagonal.Variances(1.0 / np.asarray(precisions, np.float64), smart)

    class Gaussian:
        @staticmethod
        def SqrtInformation(R, smart=True):
            R = np.asarray(R, np.float64)
            if smart and np.all(R == np.diag(np.diag(R))):
                return noiseModel.Diagonal.Sigmas(1.0 / np.diag(R), True)
            return _Noise(NOISE_GAUSSIAN, R.shape[0], R.reshape(-1))

        @staticmethod
        def Information(M, smart=True):
            M = np.asarray(M, np.float64)
            if smart and np.all(M == np.diag(np.diag(M))):
                return noiseModel.Diagonal.Precisions(np.diag(M), True)
            return _Noise(NOISE_GAUSSIAN, M.shape[0], np.linalg.cholesky(M).T.reshape(-1))

        @staticmethod
        def Covariance(S, smart=True):
            S = np.asarray(S, np.float64)
            if smart and np.all(S == np.diag(np.diag(S))):
                return noiseModel.Diagonal.Variances(np.diag(S), True)
            return noiseModel.Gaussian.Information(np.linalg.inv(S), False)

    class mEstimator:
        """linear/LossFunctions.h: the six m-estimators of the GPU path (Block re-weighting, the default scheme)."""
        class _Est:
            kind = 0
            def __init__(self, k):
                if not k > 0:
                    raise ValueError("mEstimator parameter must be > 0")        # LossFunctions.cpp constructors
                self.k = float(k)
            @classmethod
            def Create(cls, k): return cls(k)
            def weight(self, distance):
                return _robust_weight(self.kind, self.k, float(distance))
        class Fair(_Est): kind = 1
        class Huber(_Est): kind = 2
        class Cauchy(_Est): kind = 3
        class Tukey(_Est): kind = 4
        class Welsch(_Est): kind = 5
        class GemanMcClure(_Est): kind = 6

    class Robust:
        @staticmethod
        def Create(robust, noise):
            """noiseModel::Robust::Create(robust, noise) (linear/NoiseModel.cpp:731-734)."""
            n = _Noise(noise.kind, noise.dim(), noise.params)
            n.robust = (robust.kind, robust.k)
            return n


# ---- factors ---------------------------------------------------------------------------------------------------------
class GeneralSFMFactorCal3Bundler:
    def __init__(self, measured, model, cameraKey, landmarkKey):
        self.z, self.model, self.keys_ = np.asarray(measured, np.float64), model, (cameraKey, landmarkKey)


class GenericProjectionFactorCal3_S2:
    def __init__(self, measured, model, poseKey, pointKey, K: Cal3_S2, body_P_sensor: Pose3 | None = None):
        self.z, self.model, self.keys_, self.K, self.sensor = np.asarray(measured, np.float64), model, (poseKey, pointKey), K, body_P_sensor


class GenericProjectionFactorCal3DS2(GenericProjectionFactorCal3_S2):
    """GenericProjectionFactor<Pose3, Point3, Cal3DS2> (wrapped name of slam/slam.i's instantiation): same factor, K a Cal3DS2."""


class SmartProjectionParams:
    """slam/SmartFactorParams.h:42-66 + geometry/triangulation.h:558-600 (IMPLICIT_SCHUR is refused at upload)."""
    IGNORE_DEGENERACY, ZERO_ON_DEGENERACY, HANDLE_INFINITY = 0, 1, 2
    HESSIAN, IMPLICIT_SCHUR, JACOBIAN_Q, JACOBIAN_SVD = 0, 1, 2, 3

    def __init__(self, linearizationMode=0, degeneracyMode=0, retriangulationThreshold=1e-5):
        self.linearizationMode = linearizationMode
        self.degeneracyMode, self.retriangulationThreshold = degeneracyMode, retriangulationThreshold
        self.rankTolerance, self.landmarkDistanceThreshold, self.dynamicOutlierRejectionThreshold = 1.0, -1.0, -1.0
        self.enableEPI = False

    def setLinearizationMode(self, m): self.linearizationMode = m
    def setDegeneracyMode(self, m): self.degeneracyMode = m
    def setRetriangulationThreshold(self, t): self.retriangulationThreshold = t
    def setRankTolerance(self, t): self.rankTolerance = t
    def setEnableEPI(self, b): self.enableEPI = bool(b)
    def setLandmarkDistanceThreshold(self, t): self.landmarkDistanceThreshold = t
    def setDynamicOutlierRejectionThreshold(self, t): self.dynamicOutlierRejectionThreshold = t


class SmartProjectionFactorPinholeCameraCal3Bundler:
    """SmartProjectionFactor<PinholeCamera<Cal3Bundler>> (timing/timeSFMBALsmart.cpp:30-48): add(measured, cameraKey) per view."""

    def __init__(self, sharedNoiseModel, params: SmartProjectionParams | None = None):
        self.model, self.params, self.keys_, self.zs = sharedNoiseModel, params or SmartProjectionParams(), [], []

    def add(self, measured, key):
        self.zs.append(np.asarray(measured, np.float64)); self.keys_.append(key)


class BetweenFactorPose3:
    def __init__(self, key1, key2, measured: Pose3, model):
        self.keys_, self.z, self.model = (key1, key2), measured, model


class BetweenFactorPose2:
    def __init__(self, key1, key2, measured: Pose2, model):
        self.keys_, self.z, self.model = (key1, key2), measured, model


class _Prior:
    def __init__(self, key, prior, model): self.keys_, self.prior, self.model = (key,), prior, model


class PriorFactorPose3(_Prior): pass
class PriorFactorPose2(_Prior): pass
class PriorFactorPoint3(_Prior): pass
class PriorFactorPinholeCameraCal3Bundler(_Prior): pass


class Values:
    def __init__(self): self._d = {}

    def insert(self, key, value):
        if key in self._d:
            raise KeyError(f"ValuesKeyAlreadyExists: {key}")
        self._d[key] = value

    def update(self, key, value): self._d[key] = value
    def exists(self, key): return key in self._d
    def keys(self): return sorted(self._d)
    def size(self): return len(self._d)

    def at(self, key):
        if key not in self._d:
            raise KeyError(f"ValuesKeyDoesNotExist: {key}")
        return self._d[key]

    atPose3 = atPoint3 = atPinholeCameraCal3Bundler = at


class NonlinearFactorGraph:
    def __init__(self): self.factors = []

    def add(self, f): self.factors.append(f)
    push_back = add
    def size(self): return len(self.factors)
    def addPriorPose3(self, key, prior, model): self.add(PriorFactorPose3(key, prior, model))
    def addPriorPose2(self, key, prior, model): self.add(PriorFactorPose2(key, prior, model))
    def addPriorPoint3(self, key, prior, model): self.add(PriorFactorPoint3(key, prior, model))
    def addPriorPinholeCameraCal3Bundler(self, key, prior, model): self.add(PriorFactorPinholeCameraCal3Bundler(key, prior, model))

    def error(self, values: Values) -> float:
        """NonlinearFactorGraph::error on the GPU."""
        from .lib import DeviceGraph
        p, v0, _ = extract(self, values)
        dev = DeviceGraph(p)
        dev.set_values(v0)
        e = dev.error()
        dev.close()
        return e


def extract(graph: NonlinearFactorGraph, values: Values):
    """NonlinearFactorGraph + Values -> (Problem, packed values, keys in id order): the one pass the C++ shim does."""
    keys = values.keys()
    ids = {k: i for i, k in enumerate(keys)}
    vt, packed = [], []
    for k in keys:
        v = values.at(k)
        if isinstance(v, Pose3): vt.append(VAR_POSE3); packed.append(v.packed())
        elif isinstance(v, Pose2): vt.append(VAR_POSE2); packed.append(v.packed())
        elif isinstance(v, PinholeCameraCal3Bundler): vt.append(VAR_SFM_CAMERA); packed.append(v.packed())
        elif isinstance(v, np.ndarray) and v.size == 3: vt.append(VAR_POINT3); packed.append(v.astype(np.float64))
        else:
            raise ValueError(f"unsupported value type for key {k}")
    p = Problem(var_type=np.array(vt, np.int32))
    noise_ids = {}

    def nid(model, dim):
        if model.dim() != dim:
            raise ValueError("NoiseModelFactor: NoiseModel has wrong dimension")     # NonlinearFactor.cpp:97-104
        key = model.key()
        if key not in noise_ids:
            noise_ids[key] = p.add_noise(model.kind, model.dim(), model.params, model.robust)
        return noise_ids[key]

    def vid(k):
        if k not in ids:
            raise KeyError(f"ValuesKeyDoesNotExist: {k}")
        return ids[k]

    sfm, proj, btw = [], [], []
    calibs, sensors = {}, []
    for f in graph.factors:
        if isinstance(f, GeneralSFMFactorCal3Bundler):
            sfm.append((vid(f.keys_[0]), vid(f.keys_[1]), f.z, nid(f.model, 2)))
        elif isinstance(f, GenericProjectionFactorCal3_S2):
            ck = f.K.v.tobytes()
            if ck not in calibs:
                calibs[ck] = (len(calibs), f.K.v)
            si = -1
            if f.sensor is not None:
                si = len(sensors); sensors.append(f.sensor.packed())
            proj.append((vid(f.keys_[0]), vid(f.keys_[1]), f.z, nid(f.model, 2), calibs[ck][0], si))
        elif isinstance(f, SmartProjectionFactorPinholeCameraCal3Bundler):
            sp = f.params
            p.add_smart([vid(k) for k in f.keys_], np.concatenate(f.zs), nid(f.model, 2), sp.rankTolerance, sp.landmarkDistanceThreshold,
                        sp.dynamicOutlierRejectionThreshold, sp.retriangulationThreshold, sp.degeneracyMode, sp.linearizationMode, sp.enableEPI)
        elif isinstance(f, BetweenFactorPose3):
            btw.append((vid(f.keys_[0]), vid(f.keys_[1]), f.z.packed(), nid(f.model, 6)))
        elif isinstance(f, BetweenFactorPose2):   # same table: the measurement sits in the first 3 of the 12 doubles
            btw.append((vid(f.keys_[0]), vid(f.keys_[1]), np.concatenate([f.z.packed(), np.zeros(9)]), nid(f.model, 3)))
        elif isinstance(f, _Prior):
            v = vid(f.keys_[0]); t = vt[v]
            data = f.prior.packed() if hasattr(f.prior, "packed") else np.asarray(f.prior, np.float64)
            if data.size != STORAGE[t]:
                raise ValueError("prior type does not match the variable")
            p.add_prior(v, data, nid(f.model, TANGENT[t]))
        else:
            raise ValueError(f"factor type outside the GPU hot path: {type(f).__name__}")
    if sfm:
        p.sfm_cam = np.array([s[0] for s in sfm], np.int32); p.sfm_point = np.array([s[1] for s in sfm], np.int32)
        p.sfm_z = np.concatenate([s[2] for s in sfm]); p.sfm_noise = np.array([s[3] for s in sfm], np.int32)
    if proj:
        p.proj_pose = np.array([s[0] for s in proj], np.int32); p.proj_point = np.array([s[1] for s in proj], np.int32)
        p.proj_z = np.concatenate([s[2] for s in proj]); p.proj_noise = np.array([s[3] for s in proj], np.int32)
        p.proj_calib = np.array([s[4] for s in proj], np.int32); p.proj_sensor = np.array([s[5] for s in proj], np.int32)
        rows = [c[1] for c in sorted(calibs.values(), key=lambda c: c[0])]
        p.calib = np.concatenate([r[:5] for r in rows])
        if any(r.size == 9 for r in rows):      # Cal3DS2 entries: k1, k2, p1, p2 per calibration (zero rows for a Cal3_S2)
            p.calib_distortion = np.concatenate([r[5:] if r.size == 9 else np.zeros(4) for r in rows])
        p.sensor = np.concatenate(sensors) if sensors else np.zeros(0)
    if btw:
        p.between_v1 = np.array([s[0] for s in btw], np.int32); p.between_v2 = np.array([s[1] for s in btw], np.int32)
        p.between_z = np.concatenate([s[2] for s in btw]); p.between_noise = np.array([s[3] for s in btw], np.int32)
    return p, np.concatenate(packed) if packed else np.zeros(0), keys


class LevenbergMarquardtOptimizer:
    """Same surface as the wrapped gtsam.LevenbergMarquardtOptimizer (nonlinear/nonlinear.i:382-391)."""

    def __init__(self, graph: NonlinearFactorGraph, initialValues: Values, params: LevenbergMarquardtParams | None = None,
                 device: int = 0):
        self._problem, v0, self._keys = extract(graph, initialValues)
        self._types = [initialValues.at(k) for k in self._keys]
        # LevenbergMarquardtParams::setOrdering: the landmarks are always eliminated first on the device (the Schur ordering);
        # the order of the remaining variables is honoured (gtg_set_reduced_ordering), landmark keys in it are dropped
        reduced = None
        ordering = getattr(params, "ordering", None) if params is not None else None
        if ordering is not None:
            ids = {k: i for i, k in enumerate(self._keys)}
            reduced = [ids[k] for k in ordering if k in ids and self._problem.var_type[ids[k]] != 2]
            if len(reduced) != int((self._problem.var_type != 2).sum()):
                raise ValueError("LevenbergMarquardtParams.ordering must contain every non-landmark key exactly once")
        self._opt = DeviceLevenbergMarquardt(self._problem, v0, params, device=device, reduced_ordering=reduced)

    def optimize(self) -> Values:
        self._opt.optimize()
        return self.values()

    def iterate(self): self._opt.iterate()
    def error(self): return self._opt.error()
    def iterations(self): return self._opt.iterations()
    def lambda_(self): return self._opt.lambda_()
    def getInnerIterations(self): return self._opt.getInnerIterations()

    def values(self) -> Values:
        packed = self._opt.values_packed()
        off = self._problem.val_offsets()
        out = Values()
        for i, k in enumerate(self._keys):
            seg = packed[off[i]:off[i + 1]]
            proto = self._types[i]
            out.insert(k, Pose3.from_packed(seg) if isinstance(proto, Pose3) else Pose2.from_packed(seg) if isinstance(proto, Pose2)
                       else PinholeCameraCal3Bundler.from_packed(seg) if isinstance(proto, PinholeCameraCal3Bundler) else seg.copy())
        return out
