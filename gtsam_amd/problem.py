"""Structure-of-arrays factor-graph description handed across the C ABI (include/gtsam_amd.h:
``gtg_problem``).

This is what the extractor produces by walking a ``NonlinearFactorGraph`` once (reference:
``FactorGraph.h:92`` ``factors_``, ``Factor.h`` ``keys_``, ``Values.h:74-79``): dense variable ids,
one SoA table per supported factor type and a shared noise-model table.  Pure host bookkeeping --
no arithmetic of the hot path lives here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

VAR_POSE3, VAR_SFM_CAMERA, VAR_POINT3, VAR_POSE2 = 0, 1, 2, 3
FAC_GENERAL_SFM, FAC_PROJECTION, FAC_BETWEEN_POSE3, FAC_PRIOR = 0, 1, 2, 3
NOISE_UNIT, NOISE_ISOTROPIC, NOISE_DIAGONAL, NOISE_GAUSSIAN = 0, 1, 2, 3
ROBUST_NONE, ROBUST_FAIR, ROBUST_HUBER, ROBUST_CAUCHY, ROBUST_TUKEY, ROBUST_WELSCH, ROBUST_GEMANMCCLURE, ROBUST_DCS, ROBUST_L2WITHDEADZONE = range(9)

STORAGE = {VAR_POSE3: 12, VAR_SFM_CAMERA: 17, VAR_POINT3: 3, VAR_POSE2: 3}    # Pose2 = (x, y, theta)
TANGENT = {VAR_POSE3: 6, VAR_SFM_CAMERA: 9, VAR_POINT3: 3, VAR_POSE2: 3}


class gtg_problem(C.Structure):
    _i32p, _i64p, _f64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
    _fields_ = [
        ("n_vars", C.c_int32), ("var_type", _i32p),
        ("n_noise", C.c_int32), ("noise_kind", _i32p), ("noise_dim", _i32p), ("noise_off", _i64p),
        ("noise_data", _f64p), ("noise_robust", _i32p), ("noise_robust_param", _f64p),
        ("n_sfm", C.c_int64), ("sfm_cam", _i32p), ("sfm_point", _i32p), ("sfm_z", _f64p),
        ("sfm_noise", _i32p),
        ("n_proj", C.c_int64), ("proj_pose", _i32p), ("proj_point", _i32p), ("proj_z", _f64p),
        ("proj_noise", _i32p), ("proj_calib", _i32p), ("proj_sensor", _i32p),
        ("n_calib", C.c_int32), ("calib", _f64p), ("n_sensor", C.c_int32), ("sensor", _f64p),
        ("n_between", C.c_int64), ("between_v1", _i32p), ("between_v2", _i32p),
        ("between_z", _f64p), ("between_noise", _i32p),
        ("n_prior", C.c_int64), ("prior_var", _i32p), ("prior_off", _i64p), ("prior_data", _f64p),
        ("prior_noise", _i32p),
        ("calib_distortion", _f64p),
        ("n_smart", C.c_int64), ("smart_ptr", _i64p), ("smart_cam", _i32p), ("smart_z", _f64p), ("smart_noise", _i32p),
        ("smart_params", _f64p),
    ]


def _a(x, dt):
    return np.ascontiguousarray(np.asarray(x, dtype=dt).reshape(-1))


@dataclass
class Problem:
    """Host-side SoA factor graph.  All arrays are numpy; see include/gtsam_amd.h for meaning."""
    var_type: np.ndarray
    noise_kind: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    noise_dim: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    noise_off: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    noise_data: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    noise_robust: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    noise_robust_param: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    sfm_cam: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    sfm_point: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    sfm_z: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    sfm_noise: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    proj_pose: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    proj_point: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    proj_z: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    proj_noise: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    proj_calib: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    proj_sensor: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    calib: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    calib_distortion: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))   # [n_calib*4] k1,k2,p1,p2 (Cal3DS2) or empty
    # SmartProjectionFactor<PinholeCamera<Cal3Bundler>>: one track per factor (include/gtsam_amd.h)
    smart_ptr: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int64))
    smart_cam: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    smart_z: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    smart_noise: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    smart_params: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    sensor: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    between_v1: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    between_v2: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    between_z: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    between_noise: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    prior_var: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    prior_off: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    prior_data: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    prior_noise: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))

    def __post_init__(self):
        for name, dt in (("var_type", np.int32), ("noise_kind", np.int32), ("noise_dim", np.int32),
                         ("noise_off", np.int64), ("noise_data", np.float64),
                         ("noise_robust", np.int32), ("noise_robust_param", np.float64),
                         ("sfm_cam", np.int32), ("sfm_point", np.int32), ("sfm_z", np.float64),
                         ("sfm_noise", np.int32), ("proj_pose", np.int32), ("proj_point", np.int32),
                         ("proj_z", np.float64), ("proj_noise", np.int32), ("proj_calib", np.int32),
                         ("proj_sensor", np.int32), ("calib", np.float64), ("sensor", np.float64),
                         ("between_v1", np.int32), ("between_v2", np.int32),
                         ("between_z", np.float64), ("between_noise", np.int32),
                         ("prior_var", np.int32), ("prior_off", np.int64),
                         ("prior_data", np.float64), ("prior_noise", np.int32)):
            setattr(self, name, _a(getattr(self, name), dt))

    # ---- sizes -------------------------------------------------------------------------------
    @property
    def n_vars(self): return int(self.var_type.size)
    @property
    def n_sfm(self): return int(self.sfm_cam.size)
    @property
    def n_proj(self): return int(self.proj_pose.size)
    @property
    def n_between(self): return int(self.between_v1.size)
    @property
    def n_prior(self): return int(self.prior_var.size)
    @property
    def n_smart(self): return int(self.smart_noise.size)

    def val_offsets(self):
        s = np.array([STORAGE[int(t)] for t in self.var_type], np.int64)
        return np.concatenate([[0], np.cumsum(s)]).astype(np.int64)

    def dim_offsets(self):
        s = np.array([TANGENT[int(t)] for t in self.var_type], np.int64)
        return np.concatenate([[0], np.cumsum(s)]).astype(np.int64)

    # ---- noise table -------------------------------------------------------------------------
    def add_noise(self, kind: int, dim: int, data=(), robust=(ROBUST_NONE, 0.0)) -> int:
        """Append a noise model; ``data`` = () | [sigma] | sigmas[dim] | R[dim*dim] row-major;
        ``robust`` = (ROBUST_*, parameter) wraps it in noiseModel::Robust."""
        data = _a(data, np.float64)
        want = {NOISE_UNIT: 0, NOISE_ISOTROPIC: 1, NOISE_DIAGONAL: dim, NOISE_GAUSSIAN: dim * dim}[kind]
        if data.size != want:
            raise ValueError(f"noise kind {kind} dim {dim}: expected {want} parameters, got {data.size}")
        idx = int(self.noise_kind.size)
        self.noise_off = np.append(self.noise_off, np.int64(self.noise_data.size))
        self.noise_kind = np.append(self.noise_kind, np.int32(kind))
        self.noise_dim = np.append(self.noise_dim, np.int32(dim))
        self.noise_data = np.concatenate([self.noise_data, data])
        if self.noise_robust.size < idx:      # tables created before robust support / loaded without it
            self.noise_robust = np.concatenate([self.noise_robust, np.zeros(idx - self.noise_robust.size, np.int32)])
            self.noise_robust_param = np.concatenate([self.noise_robust_param, np.zeros(idx - self.noise_robust_param.size)])
        self.noise_robust = np.append(self.noise_robust, np.int32(robust[0]))
        self.noise_robust_param = np.append(self.noise_robust_param, np.float64(robust[1]))
        return idx

    # ---- ctypes view (keeps the arrays alive through the returned object) ---------------------
    def to_ctypes(self) -> gtg_problem:
        p = gtg_problem()
        keep = []

        def ptr(arr, ct):
            keep.append(arr)
            return arr.ctypes.data_as(C.POINTER(ct)) if arr.size else C.cast(None, C.POINTER(ct))

        p.n_vars = self.n_vars
        p.var_type = ptr(self.var_type, C.c_int32)
        p.n_noise = int(self.noise_kind.size)
        p.noise_kind = ptr(self.noise_kind, C.c_int32)
        p.noise_dim = ptr(self.noise_dim, C.c_int32)
        p.noise_off = ptr(self.noise_off, C.c_int64)
        p.noise_data = ptr(self.noise_data, C.c_double)
        if self.noise_robust.size != self.noise_kind.size:
            self.noise_robust = np.zeros(self.noise_kind.size, np.int32)
            self.noise_robust_param = np.zeros(self.noise_kind.size, np.float64)
        p.noise_robust = ptr(self.noise_robust, C.c_int32)
        p.noise_robust_param = ptr(self.noise_robust_param, C.c_double)
        p.n_sfm = self.n_sfm
        p.sfm_cam = ptr(self.sfm_cam, C.c_int32)
        p.sfm_point = ptr(self.sfm_point, C.c_int32)
        p.sfm_z = ptr(self.sfm_z, C.c_double)
        p.sfm_noise = ptr(self.sfm_noise, C.c_int32)
        p.n_proj = self.n_proj
        p.proj_pose = ptr(self.proj_pose, C.c_int32)
        p.proj_point = ptr(self.proj_point, C.c_int32)
        p.proj_z = ptr(self.proj_z, C.c_double)
        p.proj_noise = ptr(self.proj_noise, C.c_int32)
        p.proj_calib = ptr(self.proj_calib, C.c_int32)
        if self.proj_sensor.size == 0 and self.n_proj:
            self.proj_sensor = np.full(self.n_proj, -1, np.int32)
        p.proj_sensor = ptr(self.proj_sensor, C.c_int32)
        p.n_calib = int(self.calib.size // 5)
        p.calib = ptr(self.calib, C.c_double)
        if self.calib_distortion.size not in (0, 4 * p.n_calib):
            raise ValueError("calib_distortion must be empty or hold k1, k2, p1, p2 for every calibration")
        p.calib_distortion = ptr(self.calib_distortion, C.c_double)
        p.n_sensor = int(self.sensor.size // 12)
        p.sensor = ptr(self.sensor, C.c_double)
        p.n_between = self.n_between
        p.between_v1 = ptr(self.between_v1, C.c_int32)
        p.between_v2 = ptr(self.between_v2, C.c_int32)
        p.between_z = ptr(self.between_z, C.c_double)
        p.between_noise = ptr(self.between_noise, C.c_int32)
        p.n_prior = self.n_prior
        p.prior_var = ptr(self.prior_var, C.c_int32)
        p.prior_off = ptr(self.prior_off, C.c_int64)
        p.prior_data = ptr(self.prior_data, C.c_double)
        p.prior_noise = ptr(self.prior_noise, C.c_int32)
        p.n_smart = self.n_smart
        if self.n_smart:
            if self.smart_ptr.size != self.n_smart + 1 or self.smart_params.size != 8 * self.n_smart or \
               self.smart_cam.size != int(self.smart_ptr[-1]) or self.smart_z.size != 2 * self.smart_cam.size:
                raise ValueError("inconsistent smart factor tables")
        p.smart_ptr = ptr(np.ascontiguousarray(self.smart_ptr, np.int64), C.c_int64) if self.n_smart else C.cast(None, C.POINTER(C.c_int64))
        p.smart_cam = ptr(self.smart_cam, C.c_int32)
        p.smart_z = ptr(self.smart_z, C.c_double)
        p.smart_noise = ptr(self.smart_noise, C.c_int32)
        p.smart_params = ptr(self.smart_params, C.c_double)
        p._keep = keep
        return p

    # ---- smart factors -----------------------------------------------------------------------
    def add_smart(self, cams, zs, noise_idx: int, rank_tolerance=1.0, landmark_distance_threshold=-1.0,
                  dynamic_outlier_rejection_threshold=-1.0, retriangulation_threshold=1e-5, degeneracy_mode=0,
                  linearization_mode=0, enable_epi=False):
        """One SmartProjectionFactor<PinholeCamera<Cal3Bundler>>: camera variable ids + their pixel measurements; defaults =
        SmartProjectionParams() / TriangulationParameters() (slam/SmartFactorParams.h:58-66, geometry/triangulation.h:583-600).
        degeneracy_mode 0 IGNORE_DEGENERACY, 1 ZERO_ON_DEGENERACY, 2 HANDLE_INFINITY; linearization_mode 0 HESSIAN, 2 JACOBIAN_Q,
        3 JACOBIAN_SVD (1 = IMPLICIT_SCHUR cannot be eliminated by the reference's direct solvers and is refused); enable_epi =
        TriangulationParameters::enableEPI (the DLT point refined by the reference's LM on TriangulationFactors)."""
        cams = _a(cams, np.int32); zs = _a(zs, np.float64)
        if zs.size != 2 * cams.size or cams.size < 1:
            raise ValueError("a smart factor needs one 2-vector per camera")
        self.smart_cam = np.concatenate([self.smart_cam, cams]); self.smart_z = np.concatenate([self.smart_z, zs])
        self.smart_ptr = np.concatenate([np.asarray(self.smart_ptr, np.int64), [int(self.smart_ptr[-1]) + cams.size]]).astype(np.int64)
        self.smart_noise = np.concatenate([self.smart_noise, np.array([noise_idx], np.int32)])
        self.smart_params = np.concatenate([self.smart_params, [rank_tolerance, landmark_distance_threshold,
                                                                dynamic_outlier_rejection_threshold, retriangulation_threshold,
                                                                float(degeneracy_mode), float(linearization_mode), 1.0 if enable_epi else 0.0, 0.0]])

    # ---- priors ------------------------------------------------------------------------------
    def add_prior(self, var: int, value, noise_idx: int):
        value = _a(value, np.float64)
        if value.size != STORAGE[int(self.var_type[var])]:
            raise ValueError("prior value has wrong storage size for the variable type")
        self.prior_off = np.append(self.prior_off, np.int64(self.prior_data.size))
        self.prior_data = np.concatenate([self.prior_data, value])
        self.prior_var = np.append(self.prior_var, np.int32(var))
        self.prior_noise = np.append(self.prior_noise, np.int32(noise_idx))


def bal_problem(cams17, pts3, obs_cam, obs_pt, obs_z, noise=(NOISE_UNIT, ())):
    """BAL bundle-adjustment graph exactly as examples/SFMExample_bal.cpp:52-76 /
    timing/timeSFMBAL.cpp:33-55 build it: one GeneralSFMFactor<SfmCamera,Point3> per observation,
    cameras are variables 0..nC-1 (symbol C(i)), points nC..nC+nP-1 (symbol P(j)).
    Returns (Problem, packed initial values)."""
    cams17 = np.asarray(cams17, np.float64).reshape(-1, 17)
    pts3 = np.asarray(pts3, np.float64).reshape(-1, 3)
    nC, nP = cams17.shape[0], pts3.shape[0]
    vt = np.concatenate([np.full(nC, VAR_SFM_CAMERA, np.int32), np.full(nP, VAR_POINT3, np.int32)])
    p = Problem(var_type=vt)
    ni = p.add_noise(noise[0], 2, noise[1])
    p.sfm_cam = _a(obs_cam, np.int32)
    p.sfm_point = (_a(obs_pt, np.int32) + nC).astype(np.int32)
    p.sfm_z = _a(obs_z, np.float64)
    p.sfm_noise = np.full(p.sfm_cam.size, ni, np.int32)
    values = np.concatenate([cams17.reshape(-1), pts3.reshape(-1)])
    return p, values


def smart_bal_problem(cams17, obs_cam, obs_pt, obs_z, noise=(NOISE_UNIT, ()), min_observations=1, **smart_params):
    """The BAL graph as timing/timeSFMBALsmart.cpp:33-58 builds it: one SmartProjectionFactor<PinholeCamera<Cal3Bundler>> per track,
    the cameras are the only variables (symbol C(i)).  Returns (Problem, packed initial values = the cameras)."""
    cams17 = np.asarray(cams17, np.float64).reshape(-1, 17)
    p = Problem(var_type=np.full(cams17.shape[0], VAR_SFM_CAMERA, np.int32))
    ni = p.add_noise(noise[0], 2, noise[1])
    obs_cam = _a(obs_cam, np.int32); obs_pt = _a(obs_pt, np.int32); obs_z = _a(obs_z, np.float64).reshape(-1, 2)
    order = np.argsort(obs_pt, kind="stable")
    starts = np.flatnonzero(np.r_[True, np.diff(obs_pt[order]) != 0, True])
    for a, b in zip(starts[:-1], starts[1:]):
        if b - a >= min_observations:
            idx = order[a:b]
            p.add_smart(obs_cam[idx], obs_z[idx], ni, **smart_params)
    return p, cams17.reshape(-1).copy()


def pose_graph_problem(n_poses, v1, v2, z12, noise_kind, noise_params):
    """Pose3 pose graph: BetweenFactor<Pose3> per edge (slam/dataset.cpp:838-859).  noise_kind[k],
    noise_params[k] (36 doubles: sigma | sigmas[6] | R 6x6 row-major) per edge; identical models are
    shared in the table."""
    p = Problem(var_type=np.full(n_poses, VAR_POSE3, np.int32))
    p.between_v1 = _a(v1, np.int32)
    p.between_v2 = _a(v2, np.int32)
    p.between_z = _a(z12, np.float64)
    noise_params = np.asarray(noise_params, np.float64).reshape(-1, 36)
    table = {}
    idx = np.zeros(p.between_v1.size, np.int32)
    for k in range(p.between_v1.size):
        kind = int(noise_kind[k])
        n = {NOISE_UNIT: 0, NOISE_ISOTROPIC: 1, NOISE_DIAGONAL: 6, NOISE_GAUSSIAN: 36}[kind]
        key = (kind, noise_params[k, :n].tobytes())
        if key not in table:
            table[key] = p.add_noise(kind, 6, noise_params[k, :n])
        idx[k] = table[key]
    p.between_noise = idx
    return p


def pose2_graph_problem(n_poses, v1, v2, z3, noise_kind, noise_params):
    """Pose2 pose graph: BetweenFactor<Pose2> per edge (slam/dataset.cpp load2D).  z3 = (x, y, theta) per edge;
    noise_kind[k], noise_params[k] (9 doubles: sigma | sigmas[3] | R 3x3 row-major).  On the C ABI a Pose2 between factor
    uses the BetweenFactor table with its measurement in the first 3 of the 12 doubles (the factor's type follows
    from its variables' type)."""
    p = Problem(var_type=np.full(n_poses, VAR_POSE2, np.int32))
    p.between_v1 = _a(v1, np.int32)
    p.between_v2 = _a(v2, np.int32)
    z = np.zeros((p.between_v1.size, 12))
    z[:, :3] = np.asarray(z3, np.float64).reshape(-1, 3)
    p.between_z = _a(z, np.float64)
    noise_params = np.asarray(noise_params, np.float64).reshape(-1, 9)
    table = {}
    idx = np.zeros(p.between_v1.size, np.int32)
    for k in range(p.between_v1.size):
        kind = int(noise_kind[k])
        n = {NOISE_UNIT: 0, NOISE_ISOTROPIC: 1, NOISE_DIAGONAL: 3, NOISE_GAUSSIAN: 9}[kind]
        key = (kind, noise_params[k, :n].tobytes())
        if key not in table:
            table[key] = p.add_noise(kind, 3, noise_params[k, :n])
        idx[k] = table[key]
    p.between_noise = idx
    return p
