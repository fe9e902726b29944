"""gtsam_amd -- MI355X-native Levenberg-Marquardt inner loop behind GTSAM's API.

Only what the hot path needs: csrc/ (HIP kernels + the C ABI of include/gtsam_amd.h), lib.py (ctypes
binding, no CPU fallback), optimizer.py / params.py / problem.py (host-side mirror of the reference's
optimizer interface), datasets.py (synthetic stand-ins for absent datasets).
"""
from .params import LevenbergMarquardtParams  # noqa: F401
from .problem import Problem, bal_problem, pose_graph_problem  # noqa: F401
