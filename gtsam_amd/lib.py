"""ctypes binding of the C ABI (include/gtsam_amd.h) -> gtsam_amd/lib/libgtsam_amd.so.

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, every entry point
raises.  torch is only used by callers for multi-GPU plumbing (torch.distributed), never here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .problem import Problem, gtg_problem

_HERE = os.path.dirname(os.path.abspath(__file__))
# GTSAM_AMD_LIB: another build of the SAME library (the fenced A/B build of the dataflow protocol, a sanitizer build); no fallback
LIB_PATH = os.environ.get("GTSAM_AMD_LIB") or os.path.join(_HERE, "lib", "libgtsam_amd.so")

GTG_OK, GTG_INDETERMINATE = 0, 1
PHASES = ["linearize", "assemble", "point_eliminate", "schur", "cholesky", "solve", "linear_error",
          "retract", "error"]

# every symbol include/gtsam_amd.h declares (tests check the .so exports all of them)
SYMBOLS = ["gtg_create", "gtg_destroy", "gtg_prewarm", "gtg_last_error", "gtg_version", "gtg_upload_problem",
           "gtg_set_reduced_ordering", "gtg_values_size", "gtg_tangent_size", "gtg_set_values",
           "gtg_get_values", "gtg_get_trial_values", "gtg_error", "gtg_linearize", "gtg_try_lambda", "gtg_try_lambda_pcg",
           "gtg_accept", "gtg_get_delta", "gtg_get_gradient", "gtg_get_hessian_diagonal",
           "gtg_get_jacobians", "gtg_reduced_dim", "gtg_get_reduced_matrix", "gtg_set_allreduce",
           "gtg_enable_timing", "gtg_get_phase_ms", "gtg_reset_timing", "gtg_phase_name",
           "gtg_cholesky_flops", "gtg_cholesky_flops_block_level", "gtg_cholesky_flops_executed", "gtg_linearize_bytes", "gtg_dense_cholesky_host", "gtg_structure_hash",
           "gtg_debug_plan_sizes", "gtg_debug_plan_lists", "gtg_debug_df_plan", "gtg_debug_df_device_tables", "gtg_debug_df_chains", "gtg_debug_reduced_order", "gtg_debug_df_ctrl", "gtg_debug_df_poll_stats", "gtg_debug_df_trace", "gtg_release_cached_memory", "gtg_cached_memory_bytes", "gtg_values_device_ptr", "gtg_values_changed",
           "gtg_io_last_error", "gtg_io_bal_sizes", "gtg_io_read_bal", "gtg_io_write_bal",
           "gtg_io_g2o_sizes", "gtg_io_read_g2o", "gtg_io_write_g2o",
           "gtg_debug_scan", "gtg_debug_sort_pairs", "gtg_debug_runs"]

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)

_lib = None


class GtsamAmdError(RuntimeError):
    pass


def load():
    """Load libgtsam_amd.so; raises loudly when it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GtsamAmdError(f"{LIB_PATH} not found: build the HIP extension first "
                            "(python -c 'import __graft_entry__ as g; g.build()' or make -C gtsam_amd/csrc)")
    # torch first: its wheel bundles its own libamdhip64.so.7 / libhsa-runtime64; whichever HIP runtime is mapped
    # first serves the whole process (same soname), and torch cannot see the GPU through /opt/rocm's copy.  The
    # library itself has no torch dependency -- a C/C++ host (gtsam_amd/host) links /opt/rocm's runtime directly.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.gtg_last_error.restype = C.c_char_p
    lib.gtg_version.restype = C.c_char_p
    lib.gtg_phase_name.restype = C.c_char_p
    for name in ("gtg_values_size", "gtg_tangent_size", "gtg_reduced_dim", "gtg_structure_hash"):
        getattr(lib, name).restype = C.c_int64
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.gtg_cholesky_flops.restype = C.c_double
    lib.gtg_cholesky_flops.argtypes = [C.c_void_p]
    lib.gtg_cholesky_flops_block_level.restype = C.c_double
    lib.gtg_cholesky_flops_executed.restype = C.c_double
    lib.gtg_cholesky_flops_executed.argtypes = [C.c_void_p]
    lib.gtg_cholesky_flops_block_level.argtypes = [C.c_void_p]
    lib.gtg_linearize_bytes.restype = C.c_double
    lib.gtg_linearize_bytes.argtypes = [C.c_void_p]
    lib.gtg_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.gtg_destroy.argtypes = [C.c_void_p]
    lib.gtg_upload_problem.argtypes = [C.c_void_p, C.POINTER(gtg_problem), C.c_int, C.c_int]
    lib.gtg_set_reduced_ordering.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    for name in ("gtg_set_values", "gtg_get_values", "gtg_get_trial_values", "gtg_get_delta",
                 "gtg_get_gradient", "gtg_get_hessian_diagonal", "gtg_get_reduced_matrix"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.gtg_get_jacobians.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    lib.gtg_error.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.gtg_linearize.argtypes = [C.c_void_p]
    lib.gtg_try_lambda.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p]
    lib.gtg_try_lambda_pcg.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_int32)]
    lib.gtg_accept.argtypes = [C.c_void_p]
    lib.gtg_values_device_ptr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gtg_values_changed.argtypes = [C.c_void_p]
    lib.gtg_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
    lib.gtg_enable_timing.argtypes = [C.c_void_p, C.c_int]
    lib.gtg_reset_timing.argtypes = [C.c_void_p]
    lib.gtg_get_phase_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.gtg_dense_cholesky_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.gtg_debug_plan_sizes.argtypes = [C.c_void_p, C.c_void_p]
    lib.gtg_debug_plan_lists.argtypes = [C.c_void_p] + [C.c_void_p] * 9
    lib.gtg_debug_df_plan.argtypes = [C.c_void_p] * 4
    lib.gtg_debug_df_device_tables.argtypes = [C.c_void_p] * 5
    lib.gtg_debug_df_chains.argtypes = [C.c_void_p] * 5
    lib.gtg_debug_reduced_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.gtg_debug_df_ctrl.argtypes = [C.c_void_p] * 2
    lib.gtg_debug_df_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.gtg_debug_df_poll_stats.argtypes = [C.c_void_p, C.c_void_p]
    lib.gtg_cached_memory_bytes.restype = C.c_int64
    lib.gtg_io_last_error.restype = C.c_char_p
    lib.gtg_io_bal_sizes.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.gtg_io_read_bal.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64] + [C.c_void_p] * 5
    lib.gtg_io_write_bal.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64] + [C.c_void_p] * 5
    lib.gtg_debug_scan.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    lib.gtg_debug_sort_pairs.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    lib.gtg_debug_runs.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gtg_io_g2o_sizes.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.gtg_io_read_g2o.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int64, C.c_int64] + [C.c_void_p] * 7
    lib.gtg_io_write_g2o.argtypes = [C.c_char_p, C.c_int, C.c_int64] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
    _lib = lib
    return lib


def _check(rc, what):
    if rc < 0:
        raise GtsamAmdError(f"{what} failed (rc={rc}): {load().gtg_last_error().decode()}")
    return rc


class DeviceGraph:
    """One factor graph resident on one MI355X (one handle of the C ABI)."""

    JAC_ROW = {0: 26, 1: 20, 2: 78, 3: 90}

    def __init__(self, problem: Problem, device: int = 0, shard: int = 0, n_shards: int = 1,
                 reduced_ordering=None, allreduce=None):
        self.lib = load()
        self.problem = problem
        self.h = C.c_void_p()
        _check(self.lib.gtg_create(C.byref(self.h), device), "gtg_create")
        self._cb = None
        if allreduce is not None:
            self.set_allreduce(allreduce)
        if reduced_ordering is not None:
            o = np.ascontiguousarray(reduced_ordering, np.int32)
            _check(self.lib.gtg_set_reduced_ordering(self.h, o.ctypes.data, o.size), "gtg_set_reduced_ordering")
        self._cp = problem.to_ctypes()
        _check(self.lib.gtg_upload_problem(self.h, C.byref(self._cp), shard, n_shards), "gtg_upload_problem")
        self.val_size = self.lib.gtg_values_size(self.h)
        self.dim_size = self.lib.gtg_tangent_size(self.h)
        self.reduced_dim = self.lib.gtg_reduced_dim(self.h)
        self.reduced_vars = int((np.asarray(problem.var_type) != 2).sum())        # everything but the POINT3 variables

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.gtg_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_allreduce(self, fn):
        """fn(device_ptr:int, n_doubles:int, stream:int) -> None ; sums the buffer across shards in place."""
        def _cb(ptr, n, stream, user):
            try:
                fn(ptr, n, stream)
                return 0
            except Exception as e:  # noqa: BLE001 - must not propagate through C
                print("allreduce callback failed:", e)
                return 1
        self._cb = ALLREDUCE_FN(_cb)
        _check(self.lib.gtg_set_allreduce(self.h, self._cb, None), "gtg_set_allreduce")

    # values ----------------------------------------------------------------------------------------
    def set_values(self, v):
        v = np.ascontiguousarray(v, np.float64)
        _check(self.lib.gtg_set_values(self.h, v.ctypes.data, v.size), "gtg_set_values")

    def _get(self, fn, n):
        out = np.empty(n, np.float64)
        _check(fn(self.h, out.ctypes.data, out.size), fn.__name__)
        return out

    def values(self): return self._get(self.lib.gtg_get_values, self.val_size)
    def trial_values(self): return self._get(self.lib.gtg_get_trial_values, self.val_size)
    def delta(self): return self._get(self.lib.gtg_get_delta, self.dim_size)
    def gradient(self): return self._get(self.lib.gtg_get_gradient, self.dim_size)
    def hessian_diagonal(self): return self._get(self.lib.gtg_get_hessian_diagonal, self.dim_size)

    def reduced_matrix(self):
        n = self.reduced_dim
        out = np.empty((n, n), np.float64)
        _check(self.lib.gtg_get_reduced_matrix(self.h, out.ctypes.data, out.size), "gtg_get_reduced_matrix")
        return out

    def jacobians(self, ftype):
        n = {0: self.problem.n_sfm, 1: self.problem.n_proj, 2: self.problem.n_between, 3: self.problem.n_prior}[ftype]
        out = np.empty((n, self.JAC_ROW[ftype]), np.float64)
        _check(self.lib.gtg_get_jacobians(self.h, ftype, out.ctypes.data, out.size), "gtg_get_jacobians")
        return out

    # the hot path ------------------------------------------------------------------------------------
    def error(self):
        e = C.c_double()
        _check(self.lib.gtg_error(self.h, C.byref(e)), "gtg_error")
        return e.value

    def linearize(self):
        _check(self.lib.gtg_linearize(self.h), "gtg_linearize")

    def try_lambda(self, lam, diagonal_damping=False, min_diagonal=1e-6, max_diagonal=1e32):
        """-> (status, [linear.error(0), linear.error(delta), graph.error(trial), |delta|])."""
        out = np.zeros(4)
        rc = _check(self.lib.gtg_try_lambda(self.h, lam, int(diagonal_damping), min_diagonal, max_diagonal,
                                            out.ctypes.data), "gtg_try_lambda")
        return rc, out

    def try_lambda_pcg(self, lam, diagonal_damping=False, min_diagonal=1e-6, max_diagonal=1e32, max_iterations=500,
                       min_iterations=1, epsilon_rel=1e-3, epsilon_abs=1e-3):
        """gtg_try_lambda_pcg -> (status, out[4] as try_lambda, CG iterations)."""
        out = np.zeros(4)
        cg = np.array([max_iterations, min_iterations, epsilon_rel, epsilon_abs], np.float64)
        its = C.c_int32(0)
        rc = _check(self.lib.gtg_try_lambda_pcg(self.h, lam, int(diagonal_damping), min_diagonal, max_diagonal,
                                                cg.ctypes.data, out.ctypes.data, C.byref(its)), "gtg_try_lambda_pcg")
        return rc, out, its.value

    def accept(self):
        _check(self.lib.gtg_accept(self.h), "gtg_accept")

    def values_device_ptr(self, which=0):
        """-> (device address, doubles, stream) of the current (0) / trial (1) values: gtg_values_device_ptr."""
        p = C.c_void_p(); n = C.c_int64(); st = C.c_void_p()
        _check(self.lib.gtg_values_device_ptr(self.h, int(which), C.byref(p), C.byref(n), C.byref(st)), "gtg_values_device_ptr")
        return int(p.value or 0), int(n.value), int(st.value or 0)

    def values_changed(self):
        _check(self.lib.gtg_values_changed(self.h), "gtg_values_changed")

    # measurement ---------------------------------------------------------------------------------------
    def enable_timing(self, on=True): self.lib.gtg_enable_timing(self.h, int(on))
    def reset_timing(self): self.lib.gtg_reset_timing(self.h)

    def phase_ms(self):
        ms = np.zeros(len(PHASES)); calls = np.zeros(len(PHASES), np.int64)
        self.lib.gtg_get_phase_ms(self.h, ms.ctypes.data, calls.ctypes.data, len(PHASES))
        return {p: (float(ms[i]), int(calls[i])) for i, p in enumerate(PHASES)}

    def cholesky_flops(self): return self.lib.gtg_cholesky_flops(self.h)
    def cholesky_flops_block_level(self): return self.lib.gtg_cholesky_flops_block_level(self.h)
    def cholesky_flops_executed(self): return self.lib.gtg_cholesky_flops_executed(self.h)
    def structure_hash(self): return int(self.lib.gtg_structure_hash(self.h))
    def linearize_bytes(self): return self.lib.gtg_linearize_bytes(self.h)

    def cholesky_plan(self):
        """Test hook: the tile schedule (dict of numpy index arrays), see gtg_debug_plan_lists."""
        sz = np.zeros(8, np.int64)
        _check(self.lib.gtg_debug_plan_sizes(self.h, sz.ctypes.data), "gtg_debug_plan_sizes")
        nt, nrows, npairs_e, nbcols, nstored, nexch, np_, nparts = (int(x) for x in sz)
        d = dict(nt=nt, rows=np.zeros(nrows, np.int32), pairs=np.zeros(npairs_e, np.int32), bcols=np.zeros(nbcols, np.int32),
                 stored=np.zeros(2 * nstored, np.int32), exch=np.zeros(2 * nexch, np.int32), per_tile=np.zeros((nt, 4), np.int64),
                 per_pair=np.zeros((np_, 8), np.int64), pair_part=np.zeros(np_ if nparts else 0, np.int32),
                 part_parent=np.zeros(nparts, np.int32))
        _check(self.lib.gtg_debug_plan_lists(self.h, *(d[k].ctypes.data for k in ("rows", "pairs", "bcols", "stored", "exch", "per_tile",
                                                                                  "per_pair", "pair_part", "part_parent"))), "gtg_debug_plan_lists")
        d["stored"] = d["stored"].reshape(-1, 2); d["exch"] = d["exch"].reshape(-1, 2)
        return d

    def reduced_order(self):
        """Test hook: the variable id at every position of the reduced system's elimination order (gtg_debug_reduced_order)."""
        n = int(self.reduced_vars)
        out = np.zeros(n, np.int32)
        _check(self.lib.gtg_debug_reduced_order(self.h, out.ctypes.data, n), "gtg_debug_reduced_order")
        return out

    def df_plan(self):
        """Test hook: the dataflow schedule -- dict(nt, active, tasks[n][6] = I, J, koff, kcnt, piece, pieces, klist), see gtg_debug_df_plan."""
        sz = np.zeros(4, np.int64)
        _check(self.lib.gtg_debug_df_plan(self.h, sz.ctypes.data, None, None), "gtg_debug_df_plan")
        tasks = np.zeros((int(sz[1]), 6), np.int32); klist = np.zeros(int(sz[2]), np.int32)
        _check(self.lib.gtg_debug_df_plan(self.h, sz.ctypes.data, tasks.ctypes.data, klist.ctypes.data), "gtg_debug_df_plan")
        cs = np.zeros(3, np.int64)
        _check(self.lib.gtg_debug_df_chains(self.h, cs.ctypes.data, None, None, None), "gtg_debug_df_chains")
        coff = np.zeros(int(cs[0]) + 1, np.int32); ctiles = np.zeros(int(cs[1]), np.int32); seq = np.zeros(int(cs[2]), np.int32)
        _check(self.lib.gtg_debug_df_chains(self.h, cs.ctypes.data, coff.ctypes.data, ctiles.ctypes.data, seq.ctypes.data), "gtg_debug_df_chains")
        return dict(nt=int(sz[0]), active=bool(sz[3]), tasks=tasks, klist=klist, chain_off=coff, chain_tiles=ctiles, seq=seq)

    def df_device_tables(self):
        """Test hook: the dataflow schedule as the kernels read it, copied back from the device (gtg_debug_df_device_tables):
        (tasks[n][12], steps[m][6], chain words)."""
        sz = np.zeros(3, np.int64)
        _check(self.lib.gtg_debug_df_device_tables(self.h, sz.ctypes.data, None, None, None), "gtg_debug_df_device_tables")
        tasks = np.zeros(int(sz[0]), np.int32); steps = np.zeros(int(sz[1]), np.int32); chain = np.zeros(int(sz[2]), np.int32)
        _check(self.lib.gtg_debug_df_device_tables(self.h, sz.ctypes.data, tasks.ctypes.data, steps.ctypes.data, chain.ctypes.data), "gtg_debug_df_device_tables")
        return tasks.reshape(-1, 12), steps.reshape(-1, 6), chain

    def df_trace(self):
        """GTG_DF_TRACE=1: (tasks[n][8] stamps, chain[nt][2] stamps) of the last factorisation, 100 MHz ticks."""
        pl = self.df_plan()
        nt8, nt = 8 * pl["tasks"].shape[0], pl["nt"]
        n = nt8 + 66 * nt
        out = np.zeros(n, np.int64)
        _check(self.lib.gtg_debug_df_trace(self.h, out.ctypes.data, n), "gtg_debug_df_trace")
        self.potrf_stamps = out[nt8 + 2 * nt:].reshape(-1, 64)     # per diagonal tile: the 15 stamps of chol_device.h::potrf_body
        return out[:nt8].reshape(-1, 8), out[nt8:nt8 + 2 * nt].reshape(-1, 2)

    def df_poll_stats(self):
        """(long waits, ended on the RMW poll, of those still stale for the sc1 load, ended on the shadow word, of those still stale)"""
        out = (C.c_int64 * 5)()
        _check(self.lib.gtg_debug_df_poll_stats(self.h, out), "gtg_debug_df_poll_stats")
        return [int(x) for x in out]

    def df_ctrl(self):
        out = np.zeros(16, np.int32)
        _check(self.lib.gtg_debug_df_ctrl(self.h, out.ctypes.data), "gtg_debug_df_ctrl")
        return out

    def dense_cholesky(self, A, rhs=None):
        """Unit-test hook: (status, L (lower), x) of the device Cholesky + solve on a host matrix."""
        A = np.ascontiguousarray(A, np.float64).copy()
        x = None if rhs is None else np.ascontiguousarray(rhs, np.float64).copy()
        rc = _check(self.lib.gtg_dense_cholesky_host(self.h, A.ctypes.data, A.shape[0],
                                                     None if x is None else x.ctypes.data), "gtg_dense_cholesky_host")
        return rc, np.tril(A), x
