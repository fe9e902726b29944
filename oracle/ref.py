"""oracle/ref.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_ref/libgtsam_ref_harness.so = the REAL reference (borglab/gtsam) built
from /root/reference by oracle/Makefile.  Only tests/, tests/golden/make_golden.py,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
(gtsam_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libgtsam_ref_harness.so")
_lib = None


def available() -> bool:
    return os.path.exists(_LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref not built (run `make -C oracle ref` where /root/reference exists)")
        _lib = C.CDLL(_LIB_PATH)
        _lib.ref_graph_create.restype = C.c_void_p
        _lib.ref_graph_error.restype = C.c_double
        _lib.ref_graph_values_size.restype = C.c_int64
        _lib.ref_graph_tangent_size.restype = C.c_int64
        _lib.ref_cholesky_partial.restype = C.c_bool
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ref_lm_params(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("relative_error_tol", C.c_double),
                ("absolute_error_tol", C.c_double), ("error_tol", C.c_double),
                ("lambda_initial", C.c_double), ("lambda_factor", C.c_double),
                ("lambda_upper_bound", C.c_double), ("lambda_lower_bound", C.c_double),
                ("min_model_fidelity", C.c_double), ("diagonal_damping", C.c_int32),
                ("use_fixed_lambda_factor", C.c_int32), ("min_diagonal", C.c_double),
                ("max_diagonal", C.c_double), ("ordering_kind", C.c_int32)]


def lm_params_struct(params, ordering_kind=0) -> ref_lm_params:
    """params: any object with the LevenbergMarquardtParams field names (gtsam_amd.LevenbergMarquardtParams)."""
    return ref_lm_params(int(params.maxIterations), params.relativeErrorTol, params.absoluteErrorTol,
                         params.errorTol, params.lambdaInitial, params.lambdaFactor,
                         params.lambdaUpperBound, params.lambdaLowerBound, params.minModelFidelity,
                         int(params.diagonalDamping), int(params.useFixedLambdaFactor),
                         params.minDiagonal, params.maxDiagonal, int(ordering_kind))


class RefGraph:
    """gtsam::NonlinearFactorGraph built by the reference's own classes from a Problem."""

    JAC_ROW = {0: 2 * 9 + 2 * 3 + 2, 1: 2 * 6 + 2 * 3 + 2, 2: 36 + 36 + 6, 3: 90}

    def __init__(self, problem):
        self.problem = problem
        self._cp = problem.to_ctypes()
        self.h = C.c_void_p(lib().ref_graph_create(C.byref(self._cp)))
        self.val_size = lib().ref_graph_values_size(self.h)
        self.dim_size = lib().ref_graph_tangent_size(self.h)

    def __del__(self):
        try:
            lib().ref_graph_destroy(self.h)
        except Exception:
            pass

    def error(self, values):
        v = np.ascontiguousarray(values, np.float64)
        return float(lib().ref_graph_error(self.h, _p(v)))

    def jacobians(self, values, ftype):
        n = {0: self.problem.n_sfm, 1: self.problem.n_proj, 2: self.problem.n_between, 3: self.problem.n_prior}[ftype]
        out = np.zeros((n, self.JAC_ROW[ftype]))
        v = np.ascontiguousarray(values, np.float64)
        rc = lib().ref_graph_jacobians(self.h, _p(v), C.c_int(ftype), _p(out), C.c_int64(out.size))
        if rc:
            raise RuntimeError(f"ref_graph_jacobians rc={rc}")
        return out

    def hessian(self, values):
        v = np.ascontiguousarray(values, np.float64)
        n = self.dim_size
        H = np.zeros((n, n)); g = np.zeros(n)
        lib().ref_graph_hessian(self.h, _p(v), _p(H), _p(g))
        return H, g

    def hessian_diagonal(self, values):
        v = np.ascontiguousarray(values, np.float64)
        d = np.zeros(self.dim_size)
        rc = lib().ref_graph_hessian_diagonal(self.h, _p(v), _p(d))
        if rc == 3:
            raise RuntimeError("CheiralityException")
        return d

    def solve(self, values, lam, diagonal_damping=False, min_diag=1e-6, max_diag=1e32, ordering_kind=0):
        """Returns (status, delta, [linear.error(0), linear.error(delta)])."""
        v = np.ascontiguousarray(values, np.float64)
        d = np.zeros(self.dim_size); le = np.zeros(2)
        rc = lib().ref_graph_solve(self.h, _p(v), C.c_double(lam), C.c_int(int(diagonal_damping)),
                                   C.c_double(min_diag), C.c_double(max_diag), C.c_int(ordering_kind),
                                   _p(d), _p(le))
        return rc, d, le

    def solve_pcg(self, values, lam, diagonal_damping=False, min_diagonal=1e-6, max_diagonal=1e32, max_iterations=500,
                  epsilon_rel=1e-3, epsilon_abs=1e-3):
        """The reference's PCGSolver + BlockJacobi on the full damped system -> delta."""
        v = np.ascontiguousarray(values, np.float64)
        d = np.zeros(self.dim_size)
        lib().ref_graph_solve_pcg(self.h, _p(v), C.c_double(lam), C.c_int(int(diagonal_damping)), C.c_double(min_diagonal),
                                  C.c_double(max_diagonal), C.c_int(max_iterations), C.c_double(epsilon_rel),
                                  C.c_double(epsilon_abs), _p(d))
        return d

    def retract(self, values, delta):
        v = np.ascontiguousarray(values, np.float64)
        d = np.ascontiguousarray(delta, np.float64)
        out = np.zeros(self.val_size)
        lib().ref_graph_retract(self.h, _p(v), _p(d), _p(out))
        return out

    def lm_logfile(self, values0, params, ordering_kind=0):
        """The reference's optimize() with params.logFile set -> rows (inner, seconds, error, lambda, outer) of the CSV
        written by LevenbergMarquardtOptimizer::writeLogFile."""
        import os, tempfile
        v = np.ascontiguousarray(values0, np.float64)
        rp = lm_params_struct(params, ordering_kind)
        fd, path = tempfile.mkstemp(suffix=".csv"); os.close(fd); os.unlink(path)
        try:
            lib().ref_graph_lm_logfile(self.h, _p(v), C.byref(rp), path.encode(), None)
            return np.loadtxt(path, delimiter=",", ndmin=2)
        finally:
            if os.path.exists(path):
                os.unlink(path)

    def lm(self, values0, params, ordering_kind=0, max_trace=1000):
        """Reference LM.  Returns dict(values, trace[n,4]=(inner, error, lambda, seconds), iterations, seconds)."""
        v = np.ascontiguousarray(values0, np.float64)
        out = np.zeros(self.val_size)
        trace = np.zeros((max_trace, 4)); nt = C.c_int(0); secs = C.c_double(0)
        rp = lm_params_struct(params, ordering_kind)
        it = lib().ref_graph_lm(self.h, _p(v), C.byref(rp), _p(out), C.c_int(max_trace), _p(trace),
                                C.byref(nt), C.byref(secs))
        return dict(values=out, trace=trace[:nt.value].copy(), iterations=int(it), seconds=secs.value)

    def iteration_phases(self, values, lam, diagonal_damping, ordering_kind, with_results=False):
        """One LM iteration made of the reference's own calls, timed per phase (ms[8]); with_results: also (error, linear error
        at 0 and at delta, trial error, |delta|_2, |delta|_inf) of that lambda try."""
        v = np.ascontiguousarray(values, np.float64)
        ms = np.zeros(8); res = np.zeros(6)
        rc = lib().ref_graph_iteration_phases2(self.h, _p(v), C.c_double(lam), C.c_int(int(diagonal_damping)),
                                               C.c_int(ordering_kind), _p(ms), _p(res))
        return (rc, ms, res) if with_results else (rc, ms)


def _iteration_mt(self, values, lam, diagonal_damping, n_threads):
    """ref_graph_iteration_mt: the same iteration with linearize and the landmark eliminations split over n_threads threads in the
    harness -> (status, ms[5] = linearize, eliminate landmarks, eliminate + solve the rest, back-substitute, total, res[6])."""
    v = np.ascontiguousarray(values, np.float64)
    ms = np.zeros(5); res = np.zeros(6)
    rc = lib().ref_graph_iteration_mt(self.h, _p(v), C.c_double(lam), C.c_int(int(diagonal_damping)), C.c_int(int(n_threads)), _p(ms), _p(res))
    return rc, ms, res


RefGraph.iteration_mt = _iteration_mt


def load_bal(path):
    """SfmData::FromBalFile (sfm/SfmData.cpp:189-246) -> (cams17, pts3, obs_cam, obs_pt, obs_z)."""
    nc, npt, nobs = C.c_int64(), C.c_int64(), C.c_int64()
    lib().ref_load_bal(path.encode(), C.byref(nc), C.byref(npt), C.byref(nobs))
    cams = np.zeros((nc.value, 17)); pts = np.zeros((npt.value, 3))
    oc = np.zeros(nobs.value, np.int32); op = np.zeros(nobs.value, np.int32); oz = np.zeros((nobs.value, 2))
    lib().ref_bal_fill(_p(cams), _p(pts), _p(oc), _p(op), _p(oz))
    return cams, pts, oc, op, oz


def rewrite_bal(path_in, path_out):
    """SfmData::FromBalFile(path_in) -> writeBAL(path_out) (sfm/SfmData.cpp:249-327), the reference's own writer."""
    nc, npt, nobs = C.c_int64(), C.c_int64(), C.c_int64()
    lib().ref_load_bal(path_in.encode(), C.byref(nc), C.byref(npt), C.byref(nobs))
    return lib().ref_write_bal(path_out.encode())


def rewrite_g2o(path_in, path_out, is3d):
    """readG2o / load2D(path_in) -> writeG2o(graph, initial, path_out) (slam/dataset.cpp:636-735)."""
    nb, nv = C.c_int64(), C.c_int64()
    if is3d:
        lib().ref_load_g2o3d(path_in.encode(), C.byref(nb), C.byref(nv)); lib().ref_write_g2o3d(path_out.encode())
    else:
        lib().ref_load_2d(path_in.encode(), C.byref(nb), C.byref(nv)); lib().ref_write_g2o2d(path_out.encode())


def load_g2o3d(path):
    """readG2o(path, is3D=true) (slam/dataset.cpp:621-633) -> dict of arrays."""
    nb, nv = C.c_int64(), C.c_int64()
    lib().ref_load_g2o3d(path.encode(), C.byref(nb), C.byref(nv))
    v1 = np.zeros(nb.value, np.int64); v2 = np.zeros(nb.value, np.int64)
    z = np.zeros((nb.value, 12)); nk = np.zeros(nb.value, np.int32); nd = np.zeros((nb.value, 36))
    vk = np.zeros(nv.value, np.int64); vp = np.zeros((nv.value, 12))
    lib().ref_g2o3d_fill(_p(v1), _p(v2), _p(z), _p(nk), _p(nd), _p(vk), _p(vp))
    return dict(v1=v1, v2=v2, z=z, noise_kind=nk, noise=nd, vertex_keys=vk, vertex_poses=vp)


def load_2d(path):
    """load2D(path) (slam/dataset.cpp:208-330) -> dict of arrays (Pose2 graph)."""
    nb, nv = C.c_int64(), C.c_int64()
    lib().ref_load_2d(path.encode(), C.byref(nb), C.byref(nv))
    v1 = np.zeros(nb.value, np.int64); v2 = np.zeros(nb.value, np.int64)
    z = np.zeros((nb.value, 3)); nk = np.zeros(nb.value, np.int32); nd = np.zeros((nb.value, 9))
    vk = np.zeros(nv.value, np.int64); vp = np.zeros((nv.value, 3))
    lib().ref_2d_fill(_p(v1), _p(v2), _p(z), _p(nk), _p(nd), _p(vk), _p(vp))
    return dict(v1=v1, v2=v2, z=z, noise_kind=nk, noise=nd, vertex_keys=vk, vertex_poses=vp)


def cholesky_partial(ABC, n_frontal):
    A = np.ascontiguousarray(ABC, np.float64).copy()
    ok = lib().ref_cholesky_partial(_p(A), C.c_int(A.shape[0]), C.c_int(n_frontal))
    return bool(ok), A
