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


def check_convergence(relative_error_tol, absolute_error_tol, error_tol, current_error, new_error):
    """checkConvergence, nonlinear/NonlinearOptimizer.cpp:182-231."""
    if new_error <= error_tol:
        return True
    absolute_decrease = current_error - new_error
    relative_decrease = absolute_decrease / current_error if current_error != 0 else math.inf
    return bool((relative_error_tol and relative_decrease <= relative_error_tol) or
                absolute_decrease <= absolute_error_tol)


class DeviceLevenbergMarquardt:
    """LM over a packed Problem (the layer the Values/NonlinearFactorGraph mirror sits on)."""

    def __init__(self, problem: Problem, values0, params: LevenbergMarquardtParams | None = None,
                 device: int = 0, shard: int = 0, n_shards: int = 1, allreduce=None,
                 reduced_ordering=None):
        self.params = params if params is not None else LevenbergMarquardtParams()
        self.dev = DeviceGraph(problem, device, shard, n_shards, reduced_ordering, allreduce)
        self.dev.set_values(values0)
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

    def _try_lambda(self):
        """One tryLambda (LM.cpp:121-270); returns True when the lambda search of this iteration ends."""
        p = self.params
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
        if rc != GTG_INDETERMINATE:                      # systemSolvedSuccessfully
            old_lin, new_lin, trial_error = out[0], out[1], out[2]
            linearized_cost_change = old_lin - new_lin
            if linearized_cost_change >= 0:
                new_error = trial_error
                cost_change = self._error - new_error
                if linearized_cost_change > _EPS * old_lin:
                    model_fidelity = cost_change / linearized_cost_change
                    step_is_successful = model_fidelity > p.minModelFidelity
                if abs(cost_change) < p.relativeErrorTol * self._error:
                    stop_searching_lambda = True
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
            self._lambda *= self._factor
            self._inner += 1
            if not p.useFixedLambdaFactor:
                self._factor *= 2.0
            return self._lambda >= p.lambdaUpperBound     # give up (LM.cpp:256-261)
        return True

    def iterate(self):
        """LevenbergMarquardtOptimizer::iterate (LM.cpp:273-308): linearize once, then try lambdas."""
        self.dev.linearize()
        while not self._try_lambda():
            pass
        self.trace.append((self._inner, self._error, self._lambda, time.perf_counter() - self._t0))

    def optimize(self):
        """NonlinearOptimizer::defaultOptimize (NonlinearOptimizer.cpp:62-117). Returns packed values."""
        p = self.params
        current_error = self._error
        if current_error <= p.errorTol or self._iterations >= p.maxIterations:
            return self.dev.values()
        new_error = current_error
        while True:
            current_error = new_error
            self.iterate()
            new_error = self._error
            if p.iterationHook:
                p.iterationHook(self._iterations, current_error, new_error)
            if not (self._iterations < p.maxIterations and
                    not check_convergence(p.relativeErrorTol, p.absoluteErrorTol, p.errorTol,
                                          current_error, new_error) and math.isfinite(current_error)):
                break
        return self.dev.values()
