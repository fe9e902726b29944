"""CPU: the replay logic of the speculative lambda search (gtsam_amd/speculative.py) on a stand-in device.

The device is replaced by a few lines of numpy (a small nonlinear least-squares problem with the C ABI's try_lambda contract:
linear.error(0), linear.error(delta), error(trial), |delta|; GTG_INDETERMINATE for a singular damped system), the exchanges are
real (torch.distributed, gloo, 2 / 3 / 5 processes).  Checked: the speculative optimizer's trace (inner iterations, errors, lambdas)
and final values equal the sequential optimizer's exactly -- accepted steps, rejections, the doubled factor of the Ceres policy,
the fixed factor of the legacy policy, stopping on a small relative cost change, giving up at lambdaUpperBound."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
dist.init_process_group("gloo")
import gtsam_amd.optimizer as OPT
from gtsam_amd.params import LevenbergMarquardtParams as LMP


class FakeProblem:
    n_vars = 4; n_smart = 0
    def __init__(self, kind): self.kind = kind


def residual(kind, x):
    if kind == "rosenbrock":      # two coupled Rosenbrock valleys
        return np.array([10 * (x[1] - x[0] ** 2), 1 - x[0], 10 * (x[3] - x[2] ** 2), 1 - x[2], 0.1 * (x[0] - x[2])])
    if kind == "powell":          # Powell's singular function
        return np.array([x[0] + 10 * x[1], np.sqrt(5) * (x[2] - x[3]), (x[1] - 2 * x[2]) ** 2, np.sqrt(10) * (x[0] - x[3]) ** 2])
    return np.array([np.exp(x[0]) - 2.0, x[1] * x[1] - 1e-3 * x[1] + 4.0, x[2] - x[3], 1e-8 * x[3]])   # "flat": a residual that cannot reach zero


def jacobian(kind, x, h=1e-7):
    r0 = residual(kind, x); J = np.zeros((r0.size, x.size))
    for i in range(x.size):
        d = np.zeros_like(x); d[i] = h
        J[:, i] = (residual(kind, x + d) - residual(kind, x - d)) / (2 * h)
    return J


class FakeDevice:
    """The calls DeviceLevenbergMarquardt makes on gtsam_amd.lib.DeviceGraph."""
    def __init__(self, problem, device=0, shard=0, n_shards=1, reduced_ordering=None, allreduce=None):
        self.kind = problem.kind; self.val_size = 4; self.x = np.zeros(4); self.trial = None
    def set_values(self, v): self.x = np.array(v, float)
    def values(self): return self.x.copy()
    def error(self): r = residual(self.kind, self.x); return 0.5 * float(r @ r)
    def linearize(self): self.J = jacobian(self.kind, self.x); self.r = residual(self.kind, self.x)
    def try_lambda(self, lam, diag=False, dmin=1e-6, dmax=1e32):
        H = self.J.T @ self.J
        D = np.clip(np.diag(H), dmin, dmax) if diag else np.ones(4)
        A = H + lam * np.diag(D)
        if not np.all(np.linalg.eigvalsh(A) > 1e-300):
            return 1, np.zeros(4)
        d = np.linalg.solve(A, -self.J.T @ self.r)
        l0 = 0.5 * float(self.r @ self.r); rl = self.r + self.J @ d; l1 = 0.5 * float(rl @ rl)
        self.trial = self.x + d
        rt = residual(self.kind, self.trial)
        return 0, np.array([l0, l1, 0.5 * float(rt @ rt) if l0 - l1 >= 0 else np.inf, float(np.linalg.norm(d))])
    def accept(self): self.x = self.trial.copy()
    def close(self): pass


OPT.DeviceGraph = FakeDevice
from gtsam_amd.speculative import SpeculativeLevenbergMarquardt, TorchComm
starts = {"rosenbrock": [-1.2, 1.0, -0.5, 2.0], "powell": [3.0, -1.0, 0.0, 1.0], "flat": [3.0, 2.0, 1.0, 5.0]}
res = {}
for kind in ("rosenbrock", "powell", "flat"):
    for pname in ("legacy", "ceres", "tight"):
        prm = LMP() if pname != "ceres" else LMP.CeresDefaults()
        if pname == "tight":
            prm.lambdaUpperBound = 1e3; prm.lambdaInitial = 1e-9
        prm.setMaxIterations(60)
        a = OPT.DeviceLevenbergMarquardt(FakeProblem(kind), starts[kind], prm); a.optimize()
        b = SpeculativeLevenbergMarquardt(FakeProblem(kind), starts[kind], prm, comm=TorchComm()); b.optimize()
        ta = np.array(a.trace)[:, :3]; tb = np.array(b.trace)[:, :3]
        res[kind + "/" + pname] = dict(same=bool(ta.shape == tb.shape and np.array_equal(ta, tb) and np.array_equal(a.values_packed(), b.values_packed())
                                                 and a.getInnerIterations() == b.getInnerIterations() and a.lambda_() == b.lambda_()),
                                       inner=int(a.getInnerIterations()), iterations=int(a.iterations()), rounds=int(b.speculated))
print("RESULT " + json.dumps(res))
'''


@pytest.mark.parametrize("world", [2, 3, 5])
def test_speculative_replay_equals_the_sequential_search(world):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    recs = [json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]) for so, _ in outs]
    rejections = 0
    for rec in recs:
        for k, v in rec.items():
            assert v["same"], (k, v)
    for k, v in recs[0].items():
        rejections += v["inner"] - v["iterations"]
        assert v["rounds"] <= v["inner"]
    assert rejections > 0          # the problems do exercise rejected tries


# ---- a speculated try that raises: the failure travels through the gather as a status code; every rank still takes part
# in the collectives; the error is raised -- on ALL ranks -- only when the in-order replay reaches that try
_CHILD_ERR = _CHILD.split("OPT.DeviceGraph = FakeDevice")[0] + r'''
import os
rank = int(os.environ["RANK"])
mode = os.environ["FAIL_MODE"]
_residual = residual
def residual(kind, x):      # + a linear problem: every first try of an iteration is accepted, the second lambda is never reached
    if kind == "linear":
        return np.array([x[0] - 1.0, 2.0 * x[1] + 3.0, x[2] - x[3], x[3] - 0.5, x[0] + x[1]])
    return _residual(kind, x)


class FailingDevice(FakeDevice):
    """replica 1's device raises on every try ("unreached": only lambdas the sequential search never gets to, i.e. its speculated
    second lambda of an iteration whose first try is accepted; "reached": from the first rejected try on)"""
    tries = 0
    def try_lambda(self, lam, *a, **k):
        if rank == 1 and mode == "always":
            raise RuntimeError("injected: the step was not computed")
        return super().try_lambda(lam, *a, **k)


OPT.DeviceGraph = FailingDevice
from gtsam_amd.speculative import SpeculativeLevenbergMarquardt, TorchComm
prm = LMP.CeresDefaults(); prm.setMaxIterations(60)
kind = os.environ["FAIL_KIND"]
start = {"rosenbrock": [-1.2, 1.0, -0.5, 2.0], "linear": [0.5, 0.1, 0.2, 0.2]}[kind]
pk = kind
OPT.DeviceGraph = FakeDevice
a = OPT.DeviceLevenbergMarquardt(FakeProblem(pk), start, prm); a.optimize()
OPT.DeviceGraph = FailingDevice
out = {"rejections": int(a.getInnerIterations() - a.iterations())}
try:
    b = SpeculativeLevenbergMarquardt(FakeProblem(pk), start, prm, comm=TorchComm()); b.optimize()
    out["raised"] = None
    out["same"] = bool(np.array_equal(np.array(a.trace)[:, :3], np.array(b.trace)[:, :3]) and np.array_equal(a.values_packed(), b.values_packed()))
except RuntimeError as e:
    out["raised"] = str(e)
print("RESULT " + json.dumps(out))
'''


@pytest.mark.parametrize("kind", ["linear", "rosenbrock"])
def test_a_failing_speculated_try_is_an_error_only_where_the_trajectory_reaches_it(kind):
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FAIL_MODE="always", FAIL_KIND=kind)
        procs.append(subprocess.Popen([sys.executable, "-c", _CHILD_ERR % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]          # (a rank left alone in a collective would run into this bound)
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    recs = [json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]) for so, _ in outs]
    assert (recs[0]["rejections"] == 0) == (kind == "linear")
    if recs[0]["rejections"] == 0:
        # replica 1 only ever speculates on lambdas behind an accepted first try: nobody raises, the trajectory is the sequential one
        assert all(r["raised"] is None and r["same"] for r in recs), recs
    else:
        # the first rejection makes replica 1's try the next one in order: both ranks raise, rank 1 with its own error
        assert all(r["raised"] for r in recs), recs
        assert "injected" in recs[1]["raised"] and "replica 1" in recs[0]["raised"], recs
