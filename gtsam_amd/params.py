"""LevenbergMarquardtParams -- host mirror of the reference's parameter struct.

Field names, defaults and the two presets follow nonlinear/LevenbergMarquardtParams.h:35-157 and
nonlinear/NonlinearOptimizerParams.h:35-182 (the Python wrapper exposes the same names through
nonlinear/nonlinear.i).
"""
from __future__ import annotations


class PCGSolverParameters:
    """linear/PCGSolver.h + ConjugateGradientSolver.h:34-50 defaults; the preconditioner of the GPU path is block Jacobi
    (PreconditionerParameters BLOCK_JACOBI) on the implicit Schur complement."""

    def __init__(self, maxIterations=500, minIterations=1, epsilon_rel=1e-3, epsilon_abs=1e-3):
        self.maxIterations, self.minIterations = int(maxIterations), int(minIterations)
        self.epsilon_rel, self.epsilon_abs = float(epsilon_rel), float(epsilon_abs)
        self.preconditioner = "BLOCK_JACOBI"


class LevenbergMarquardtParams:
    def __init__(self):
        # NonlinearOptimizerParams.h:43-48
        self.maxIterations = 100
        self.relativeErrorTol = 1e-5
        self.absoluteErrorTol = 1e-5
        self.errorTol = 0.0
        self.verbosity = "SILENT"
        self.orderingType = "COLAMD"
        self.ordering = None
        self.iterationHook = None
        self.linearSolverType = "MULTIFRONTAL_CHOLESKY"   # or "Iterative" (NonlinearOptimizerParams.h:92-101) with iterativeParams
        self.iterativeParams = PCGSolverParameters()
        # LevenbergMarquardtParams.h:62-66
        self.verbosityLM = "SILENT"
        self.diagonalDamping = False
        self.minDiagonal = 1e-6
        self.maxDiagonal = 1e32
        self.logFile = ""
        self.SetLegacyDefaults(self)

    @staticmethod
    def SetLegacyDefaults(p):
        """LevenbergMarquardtParams.h:69-82."""
        p.maxIterations = 100
        p.relativeErrorTol = 1e-5
        p.absoluteErrorTol = 1e-5
        p.lambdaInitial = 1e-5
        p.lambdaFactor = 10.0
        p.lambdaUpperBound = 1e5
        p.lambdaLowerBound = 0.0
        p.minModelFidelity = 1e-3
        p.diagonalDamping = False
        p.useFixedLambdaFactor = True

    @staticmethod
    def SetCeresDefaults(p):
        """LevenbergMarquardtParams.h:85-98."""
        p.maxIterations = 50
        p.absoluteErrorTol = 0
        p.relativeErrorTol = 1e-6
        p.lambdaUpperBound = 1e32
        p.lambdaLowerBound = 1e-16
        p.lambdaInitial = 1e-04
        p.lambdaFactor = 2.0
        p.minModelFidelity = 1e-3
        p.diagonalDamping = True
        p.useFixedLambdaFactor = False

    @classmethod
    def LegacyDefaults(cls):
        p = cls(); cls.SetLegacyDefaults(p); return p

    @classmethod
    def CeresDefaults(cls):
        p = cls(); cls.SetCeresDefaults(p); return p

    # wrapper-style setters/getters (LevenbergMarquardtParams.h:127-146, NonlinearOptimizerParams.h:50-60)
    def setMaxIterations(self, v): self.maxIterations = int(v)
    def setRelativeErrorTol(self, v): self.relativeErrorTol = float(v)
    def setAbsoluteErrorTol(self, v): self.absoluteErrorTol = float(v)
    def setErrorTol(self, v): self.errorTol = float(v)
    def setVerbosity(self, s): self.verbosity = s
    def setVerbosityLM(self, s): self.verbosityLM = s
    def setDiagonalDamping(self, f): self.diagonalDamping = bool(f)
    def setlambdaFactor(self, v): self.lambdaFactor = float(v)
    def setlambdaInitial(self, v): self.lambdaInitial = float(v)
    def setlambdaLowerBound(self, v): self.lambdaLowerBound = float(v)
    def setlambdaUpperBound(self, v): self.lambdaUpperBound = float(v)
    def setUseFixedLambdaFactor(self, f): self.useFixedLambdaFactor = bool(f)
    def setLogFile(self, s): self.logFile = s
    def setOrdering(self, o): self.ordering = o
    def getMaxIterations(self): return self.maxIterations
    def getRelativeErrorTol(self): return self.relativeErrorTol
    def getAbsoluteErrorTol(self): return self.absoluteErrorTol
    def getErrorTol(self): return self.errorTol
    def getDiagonalDamping(self): return self.diagonalDamping
    def getlambdaFactor(self): return self.lambdaFactor
    def getlambdaInitial(self): return self.lambdaInitial
    def getlambdaLowerBound(self): return self.lambdaLowerBound
    def getlambdaUpperBound(self): return self.lambdaUpperBound
    def getUseFixedLambdaFactor(self): return self.useFixedLambdaFactor
