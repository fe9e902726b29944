"""C-ABI library: loads, exports every symbol include/gtsam_amd.h declares (no compute without a GPU);
host logic: Problem packing, LM parameter presets, the LM state machine driven by an oracle-backed fake device,
noise-model dimension checks, and that the product fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gtsam_amd import lib
from gtsam_amd.params import LevenbergMarquardtParams as LMP
from gtsam_amd.problem import NOISE_UNIT, Problem, bal_problem, gtg_problem
from oracle import gtsam_oracle as O
from tests import problems as PB
from tests.conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gtsam_amd.h")).read()
    declared = set(re.findall(r"\b(gtg_[a-z_0-9]+)\s*\(", hdr)) - {"gtg_allreduce_fn"}
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    so = C.CDLL(lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(so, name), name
    lib.load()
    assert b"gtsam_amd" in lib.load().gtg_version()


def test_struct_layout_matches_header_field_order():
    hdr = open(os.path.join(ROOT, "include", "gtsam_amd.h")).read()
    body = hdr[hdr.index("typedef struct gtg_problem {"):hdr.index("} gtg_problem;")]
    names = re.findall(r"(?:const\s+)?(?:int32_t|int64_t|double)\s*\*?\s*([a-z_0-9]+);", body)
    assert names == [f[0] for f in gtg_problem._fields_]


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.GtsamAmdError):
        lib.DeviceGraph(Problem(var_type=np.array([0], np.int32)))


def test_params_presets_match_reference_defaults():
    """LevenbergMarquardtParams.h:69-98."""
    p = LMP()
    assert (p.maxIterations, p.relativeErrorTol, p.absoluteErrorTol, p.lambdaInitial, p.lambdaFactor, p.lambdaUpperBound,
            p.lambdaLowerBound, p.minModelFidelity, p.diagonalDamping, p.useFixedLambdaFactor) == \
        (100, 1e-5, 1e-5, 1e-5, 10.0, 1e5, 0.0, 1e-3, False, True)
    c = LMP.CeresDefaults()
    assert (c.maxIterations, c.relativeErrorTol, c.absoluteErrorTol, c.lambdaInitial, c.lambdaFactor, c.lambdaUpperBound,
            c.lambdaLowerBound, c.diagonalDamping, c.useFixedLambdaFactor, c.minDiagonal, c.maxDiagonal) == \
        (50, 1e-6, 0, 1e-4, 2.0, 1e32, 1e-16, True, False, 1e-6, 1e32)


def test_problem_packing_roundtrip():
    g = load_golden("dubrovnik_3_7")
    p, v0 = PB.dubrovnik_sfmexample(g)
    c = p.to_ctypes()
    assert c.n_vars == 10 and c.n_sfm == 19 and c.n_prior == 2 and c.n_noise == 3
    assert np.ctypeslib.as_array(c.sfm_cam, (19,)).tolist() == p.sfm_cam.tolist()
    assert v0.size == 3 * 17 + 7 * 3 and p.val_offsets()[-1] == v0.size and p.dim_offsets()[-1] == 48
    with pytest.raises(ValueError):
        p.add_noise(NOISE_UNIT, 2, [1.0])
    with pytest.raises(ValueError):
        p.add_prior(0, np.zeros(3), 0)


class FakeDevice:
    """Oracle-backed stand-in for lib.DeviceGraph (tests only): lets the host LM state machine of
    gtsam_amd/optimizer.py run on CPU so its accept/reject/lambda policy is pinned against the golden traces."""

    def __init__(self, problem, *a, **k):
        self.p = problem; self.v = None; self.trial = None; self.lin = None

    def set_values(self, v): self.v = np.array(v, float)
    def values(self): return self.v.copy()
    def error(self): return O.error(self.p, self.v)
    def linearize(self): pass

    def try_lambda(self, lam, dd, dmin, dmax):
        st, d, H, g, lin = O.solve_damped(self.p, self.v, lam, dd, dmin, dmax)
        if st:
            return 1, np.zeros(4)
        l0 = O.linear_error(self.p, lin, 0 * d); l1 = O.linear_error(self.p, lin, d)
        self.trial = O.retract(self.p, self.v, d)
        te = O.error(self.p, self.trial) if l0 - l1 >= 0 else np.inf
        return 0, np.array([l0, l1, te, np.linalg.norm(d)])

    def accept(self): self.v = self.trial


@pytest.mark.parametrize("which,preset", [("timesfm", "ceres"), ("default", "legacy"), ("sfmex", "legacy")])
def test_host_lm_state_machine_against_golden_traces(monkeypatch, which, preset):
    from gtsam_amd import optimizer
    monkeypatch.setattr(optimizer, "DeviceGraph", FakeDevice)
    g = load_golden("dubrovnik_3_7")
    p, v0 = PB.dubrovnik_sfmexample(g) if which == "sfmex" else PB.dubrovnik_timesfm(g)
    params = LMP.CeresDefaults() if preset == "ceres" else LMP()
    hooks = []
    params.iterationHook = lambda it, before, after: hooks.append((it, before, after))
    opt = optimizer.DeviceLevenbergMarquardt(p, v0, params)
    opt.optimize()
    tr = np.array(opt.trace)[:, :3]; ref = g[which + "_trace"]
    assert tr.shape == ref.shape and np.array_equal(tr[:, 0], ref[:, 0])
    assert np.abs(tr[:, 1] - ref[:, 1]).max() <= 1e-7 * np.abs(ref[:, 1]).max()
    assert np.allclose(tr[:, 2], ref[:, 2], rtol=1e-9)
    assert opt.iterations() == int(g[which + "_iterations"]) and len(hooks) == opt.iterations() + (len(tr) - 1 - opt.iterations())
    assert opt.getInnerIterations() == int(ref[-1, 0]) and abs(opt.lambda_() - ref[-1, 2]) <= 1e-9 * ref[-1, 2]


def test_check_convergence_semantics():
    """nonlinear/NonlinearOptimizer.cpp:182-231."""
    from gtsam_amd.optimizer import check_convergence as cc
    assert cc(1e-5, 1e-5, 0.0, 10.0, 10.0 - 1e-6)          # absolute decrease below tol
    assert cc(1e-5, 0.0, 0.0, 10.0, 10.0 - 5e-5)            # relative decrease below tol
    assert not cc(1e-5, 1e-5, 0.0, 10.0, 9.0)
    assert cc(1e-5, 1e-5, 1.0, 10.0, 0.5)                   # error below errorTol
    assert cc(1e-5, 1e-5, 0.0, 10.0, 11.0)                  # error increased -> "converged" (stops)
    assert not cc(0.0, -1.0, 0.0, 10.0, 10.0)               # relTol 0 disables the relative test


@pytest.mark.parametrize("key,which,preset", [("timesfm_ceres", "timesfm", "ceres"), ("timesfm_legacy", "timesfm", "legacy"),
                                              ("sfmex_legacy", "sfmex", "legacy")])
def test_logfile_matches_the_reference_csv(monkeypatch, tmp_path, key, which, preset):
    """LevenbergMarquardtParams::logFile (LevenbergMarquardtOptimizer.cpp:101-118, rows appended at :283-303): same rows
    as the CSV the real reference wrote for the same problem (fixture lm_logfile.npz from tests/golden/make_golden.py);
    the seconds column is the only one that may differ."""
    from gtsam_amd import optimizer
    monkeypatch.setattr(optimizer, "DeviceGraph", FakeDevice)
    g = load_golden("dubrovnik_3_7")
    p, v0 = PB.dubrovnik_sfmexample(g) if which == "sfmex" else PB.dubrovnik_timesfm(g)
    params = LMP.CeresDefaults() if preset == "ceres" else LMP()
    params.setLogFile(str(tmp_path / "lm.csv"))
    optimizer.DeviceLevenbergMarquardt(p, v0, params).optimize()
    mine = np.loadtxt(params.logFile, delimiter=",", ndmin=2)
    ref = load_golden("lm_logfile")[key]
    assert mine.shape == ref.shape
    assert np.array_equal(mine[:, [0, 4]], ref[:, [0, 4]])               # inner / outer iteration counters
    assert np.allclose(mine[:, [2, 3]], ref[:, [2, 3]], rtol=2e-6)        # error, lambda at the stream's 6 significant digits
    assert (np.diff(mine[:, 1]) >= 0).all()


def test_verbosity_messages(monkeypatch, capsys):
    """SUMMARY / TRYLAMBDA / TERMINATION print what the reference prints (LM.cpp:137-261, NonlinearOptimizer.cpp:62-231)."""
    from gtsam_amd import optimizer
    monkeypatch.setattr(optimizer, "DeviceGraph", FakeDevice)
    g = load_golden("dubrovnik_3_7")
    p, v0 = PB.dubrovnik_timesfm(g)
    params = LMP.CeresDefaults(); params.setVerbosityLM("SUMMARY"); params.setVerbosity("TERMINATION")
    optimizer.DeviceLevenbergMarquardt(p, v0, params).optimize()
    out = capsys.readouterr().out
    assert "Initial error: 2764.22, values: 10" in out and "iter      cost      cost_change    lambda  success iter_time" in out
    assert "iterations: 50 >? 50" in out and "Terminating because reached maximum iterations" in out   # the golden Ceres run ends there
    params = LMP(); params.setVerbosityLM("TRYLAMBDA"); params.setVerbosity("TERMINATION")
    optimizer.DeviceLevenbergMarquardt(p, v0, params).optimize()
    out = capsys.readouterr().out
    assert "trying lambda = 1e-05" in out and "increasing lambda" in out and "modelFidelity: " in out
    assert "converged" in out and "relativeDecrease: " in out and "iterations: 13 >? 100" in out


@pytest.mark.parametrize("preset", ["legacy", "ceres"])
def test_restart_from_values_and_lambda_continues_the_same_trajectory(monkeypatch, preset):
    """Checkpoint / resume as the reference offers it: the optimizer state is values() + lambda()
    (+ the current lambda factor for the Ceres policy); a new optimizer constructed from them continues on the
    uninterrupted run's trajectory."""
    import copy
    from gtsam_amd import optimizer
    monkeypatch.setattr(optimizer, "DeviceGraph", FakeDevice)
    g = load_golden("dubrovnik_3_7")
    p, v0 = PB.dubrovnik_timesfm(g)
    params = LMP.CeresDefaults() if preset == "ceres" else LMP()
    params.setMaxIterations(9)
    whole = optimizer.DeviceLevenbergMarquardt(p, v0, params); whole.optimize()
    first = optimizer.DeviceLevenbergMarquardt(p, v0, params)
    for _ in range(4):
        first.iterate()
    ckpt = dict(values=first.values_packed(), lam=first.lambda_(), factor=first._factor, done=first.iterations())
    p2 = copy.copy(params); p2.lambdaInitial = ckpt["lam"]; p2.lambdaFactor = ckpt["factor"]; p2.setMaxIterations(9 - ckpt["done"])
    second = optimizer.DeviceLevenbergMarquardt(p, ckpt["values"], p2); second.optimize()
    assert second.iterations() + ckpt["done"] == whole.iterations()
    assert abs(second.error() - whole.error()) <= 1e-12 * whole.error() and abs(second.lambda_() - whole.lambda_()) <= 1e-12 * whole.lambda_()
    assert np.abs(second.values_packed() - whole.values_packed()).max() <= 1e-12 * np.abs(whole.values_packed()).max()
