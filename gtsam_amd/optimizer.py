"""LevenbergMarquardtOptimizer -- host mirror of the reference's optimizer around the device path.

The control flow restates nonlinear/LevenbergMarquardtOptimizer.cpp:121-308 (iterate / tryLambda),
nonlinear/internal/LevenbergMarquardtState.h:70-94 (lambda policy) and
nonlinear/NonlinearOptimizer.cpp:62-117,182-231 (defaultOptimize / checkConvergence); every O(n)
operation is a call through the C ABI into HIP (gtsam_amd/lib.py).  Same method names as the
reference's wrapper (nonlinear/nonlinear.i:382-391): optimize(), iterate(), error(), iterations(),
lambda_(), values(), getInnerIterations().
"""
from __future__ import annotations

import math
import time

import numpy as np

from .lib import GTG_INDETERMINATE, DeviceGraph
from .params import LevenbergMarquardtParams
from .problem import Problem

_EPS = float(np.finfo(np.float64).eps)


# verbosity levels, NonlinearOptimizerParams.h:38-40 and LevenbergMarquardtParams.h:38-40
_VERBOSITY = {"SILENT": 0, "TERMINATION": 1, "ERROR": 2, "VALUES": 3, "DELTA": 4, "LINEAR": 5}
_VERBOSITY_LM = {"SILENT": 0, "SUMMARY": 1, "TERMINATION": 2, "LAMBDA": 3, "TRYLAMBDA": 4, "TRYCONFIG": 5, "DAMPED": 6, "TRYDELTA": 7}


def _level(table, v):
    return v if isinstance(v, int) else table[str(v).upper()]


def check_convergence(relative_error_tol, absolute_error_tol, error_tol, current_error, new_error, verbosity="SILENT"):
    """checkConvergence, nonlinear/NonlinearOptimizer.cpp:182-231 (same messages at the same verbosity levels)."""
    verb = _level(_VERBOSITY, verbosity)
    if verb >= _VERBOSITY["ERROR"]:
        print(f"errorThreshold: {new_error:g} {'<' if new_error <= error_tol else '>'} {error_tol:g}")
    if new_error <= error_tol:
        return True
    absolute_decrease = current_error - new_error
    if verb >= _VERBOSITY["ERROR"]:
        print(f"absoluteDecrease: {absolute_decrease:.12g} {'<' if absolute_decrease <= absolute_error_tol else '>='} {absolute_error_tol:g}")
    with np.errstate(divide="ignore", invalid="ignore"):   # plain IEEE division, as the reference: x/0 = +-inf, 0/0 = nan
        relative_decrease = float(np.float64(absolute_decrease) / np.float64(current_error))
    if verb >= _VERBOSITY["ERROR"]:
        print(f"relativeDecrease: {relative_decrease:.12g} {'<' if relative_decrease <= relative_error_tol else '>='} {relative_error_tol:g}")
    converged = bool((relative_error_tol and relative_decrease <= relative_error_tol) or
                     absolute_decrease <= absolute_error_tol)
    if verb >= _VERBOSITY["TERMINATION"] and converged:
        print("converged" if absolute_decrease >= 0.0 else "Warning:  stopping nonlinear iterations because error increased")
        print(f"errorThreshold: {new_error:g} <? {error_tol:g}")
        print(f"absoluteDecrease: {absolute_decrease:.12g} <? {absolute_error_tol:g}")
        print(f"relativeDecrease: {relative_decrease:.12g} <? {relative_error_tol:g}")
    return converged


class DeviceLevenbergMarquardt:
    """LM over a packed Problem (the layer the Values/NonlinearFactorGraph mirror sits on)."""

    def __init__(self, problem: Problem, values0, params: LevenbergMarquardtParams | None = None,
                 device: int = 0, shard: int = 0, n_shards: int = 1, allreduce=None,
                 reduced_ordering=None):
        self.params = params if params is not None else LevenbergMarquardtParams()
        self.dev = DeviceGraph(problem, device, shard, n_shards, reduced_ordering, allreduce)
        self.dev.set_values(values0)
        self._n_values = int(problem.n_vars)
        self._t0 = time.perf_counter()
        # State(initialValues, graph.error(initialValues), lambdaInitial, lambdaFactor), LM.cpp:47-53
        self._error = self.dev.error()
        self._lambda = float(self.params.lambdaInitial)
        self._factor = float(self.params.lambdaFactor)
        self._iterations = 0
        self._inner = 0
        self.trace = [(0, self._error, self._lambda, 0.0)]  # (inner, error, lambda, seconds): logFile schema LM.cpp:104-118

    # accessors (NonlinearOptimizer.h:105-121, LM.h:81-90)
    def error(self): return self._error
    def iterations(self): return self._iterations
    def lambda_(self): return self._lambda
    def getInnerIterations(self): return self._inner
    def values_packed(self): return self.dev.values()

    def _write_log_file(self, current_error):
        """writeLogFile, LM.cpp:101-118: inner iterations, seconds since construction, error, lambda, outer iterations."""
        if self.params.logFile:
            with open(self.params.logFile, "a") as os_:
                os_.write(f"{self._inner},{time.perf_counter() - self._t0:g},{current_error:g},{self._lambda:g},{self._iterations}\n")

    def _try_lambda(self):
        """One tryLambda (LM.cpp:121-270); returns True when the lambda search of this iteration ends."""
        p = self.params
        vlm = _level(_VERBOSITY_LM, p.verbosityLM)
        verbose = vlm >= _VERBOSITY_LM["TRYLAMBDA"]
        t_start = time.perf_counter()
        if verbose:
            print(f"trying lambda = {self._lambda:g}")
        if vlm >= _VERBOSITY_LM["DAMPED"]:
            print(f"building damped system with lambda {self._lambda:g}")
        if p.linearSolverType == "Iterative":            # NonlinearOptimizer::solve, Iterative branch (NonlinearOptimizer.cpp:154-172)
            it = p.iterativeParams
            rc, out, self.last_cg_iterations = self.dev.try_lambda_pcg(
                self._lambda, p.diagonalDamping, p.minDiagonal, p.maxDiagonal, it.maxIterations, it.minIterations,
                it.epsilon_rel, it.epsilon_abs)
        else:
            rc, out = self.dev.try_lambda(self._lambda, p.diagonalDamping, p.minDiagonal, p.maxDiagonal)
        step_is_successful = False
        stop_searching_lambda = False
        model_fidelity = 0.0
        new_error = math.inf
        cost_change = 0.0
        if rc != GTG_INDETERMINATE:                      # systemSolvedSuccessfully
            old_lin, new_lin, trial_error = out[0], out[1], out[2]
            if verbose:
                print(f"linear delta norm = {out[3]:g}")
            linearized_cost_change = old_lin - new_lin
            if verbose:
                print(f"newlinearizedError = {new_lin:g}  linearizedCostChange = {linearized_cost_change:g}")
            if linearized_cost_change >= 0:
                new_error = trial_error
                if verbose:
                    print("calculating error:")
                    print(f"old error ({self._error:g}) new (tentative) error ({new_error:g})")
                cost_change = self._error - new_error
                if linearized_cost_change > _EPS * old_lin:
                    model_fidelity = cost_change / linearized_cost_change
                    step_is_successful = model_fidelity > p.minModelFidelity
                    if verbose:
                        print(f"modelFidelity: {model_fidelity:g}")
                min_absolute_tolerance = p.relativeErrorTol * self._error
                if abs(cost_change) < min_absolute_tolerance:
                    if verbose:
                        print(f"abs(costChange)={abs(cost_change):g}  minAbsoluteTolerance={min_absolute_tolerance:g}"
                              f" (relativeErrorTol={p.relativeErrorTol:g})")
                    stop_searching_lambda = True
        if vlm == _VERBOSITY_LM["SUMMARY"]:              # LM.cpp:223-242
            if self._iterations == 0:
                print("iter      cost      cost_change    lambda  success iter_time")
            print(f"{self._iterations:4d} {new_error:12g} {cost_change:12.2g} {self._lambda:10.2g} "
                  f"{int(rc != GTG_INDETERMINATE):6d} {time.perf_counter() - t_start:10.2g}")
        if step_is_successful:
            # decreaseLambda, LMState.h:81-94
            if p.useFixedLambdaFactor:
                self._lambda /= self._factor
            else:
                self._lambda *= max(1.0 / 3.0, 1.0 - (2.0 * model_fidelity - 1.0) ** 3)
                self._factor *= 2.0
            self._lambda = max(p.lambdaLowerBound, self._lambda)
            self.dev.accept()
            self._error = new_error
            self._iterations += 1
            self._inner += 1
            return True
        if not stop_searching_lambda:
            # increaseLambda, LMState.h:70-76
            if verbose:
                print("increasing lambda")
            self._lambda *= self._factor
            self._inner += 1
            if not p.useFixedLambdaFactor:
                self._factor *= 2.0
            if self._lambda >= p.lambdaUpperBound:        # give up (LM.cpp:256-261)
                if _level(_VERBOSITY, p.verbosity) >= _VERBOSITY["TERMINATION"] or vlm == _VERBOSITY_LM["SUMMARY"]:
                    print("Warning:  Levenberg-Marquardt giving up because cannot decrease error with maximum lambda")
                return True
            return False
        if verbose:
            print("Levenberg-Marquardt: stopping as relative cost reduction is small")
        return True

    def iterate(self):
        """LevenbergMarquardtOptimizer::iterate (LM.cpp:273-308): linearize once, then try lambdas."""
        if _level(_VERBOSITY_LM, self.params.verbosityLM) >= _VERBOSITY_LM["DAMPED"]:
            print("linearizing = ")
        self.dev.linearize()
        if self._inner == 0:                              # write initial error (LM.cpp:283-290)
            self._write_log_file(self._error)
            if _level(_VERBOSITY_LM, self.params.verbosityLM) == _VERBOSITY_LM["SUMMARY"]:
                print(f"Initial error: {self._error:g}, values: {self._n_values}")
        while not self._try_lambda():                     # the reference logs after every try that keeps searching
            self._write_log_file(self._error)
        self.trace.append((self._inner, self._error, self._lambda, time.perf_counter() - self._t0))

    def optimize(self):
        """NonlinearOptimizer::defaultOptimize (NonlinearOptimizer.cpp:62-117). Returns packed values."""
        p = self.params
        verb = _level(_VERBOSITY, p.verbosity)
        current_error = self._error
        if current_error <= p.errorTol:
            if verb >= _VERBOSITY["ERROR"]:
                print(f"Exiting, as error = {current_error:g} < {p.errorTol:g}")
            return self.dev.values()
        if verb >= _VERBOSITY["ERROR"]:
            print(f"Initial error: {current_error:g}")
        if self._iterations >= p.maxIterations:
            if verb >= _VERBOSITY["TERMINATION"]:
                print(f"iterations: {self._iterations} >? {p.maxIterations}")
            return self.dev.values()
        new_error = current_error
        while True:
            current_error = new_error
            self.iterate()
            new_error = self._error
            if p.iterationHook:
                p.iterationHook(self._iterations, current_error, new_error)
            if verb >= _VERBOSITY["ERROR"]:
                print(f"newError: {new_error:g}")
            if not (self._iterations < p.maxIterations and
                    not check_convergence(p.relativeErrorTol, p.absoluteErrorTol, p.errorTol,
                                          current_error, new_error, p.verbosity) and math.isfinite(current_error)):
                break
        if verb >= _VERBOSITY["TERMINATION"]:
            print(f"iterations: {self._iterations} >? {p.maxIterations}")
            if self._iterations >= p.maxIterations:
                print("Terminating because reached maximum iterations")
        return self.dev.values()
