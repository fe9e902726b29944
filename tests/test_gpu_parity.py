"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden fixtures.

Tolerances (FP64; stated per quantity):
  per-factor whitened Jacobians / rhs        <= 1e-12 relative (same formulas, FMA / ordering only)
  Hessian diagonal, gradient                 <= 1e-10 relative (summation order)
  delta of one damped solve                  <= 1e-7 relative in max-norm (conditioning of the damped system)
  errors (chi^2 / 2) on identical inputs     <= 1e-9 relative
  LM trajectories: identical accept/reject sequence and per-iteration error <= 1e-6 relative on the
  fixtures where the reference's own trajectory is not FP-marginal; final error <= 1e-6 relative.
"""
import os
import sys

import numpy as np
import pytest

from gtsam_amd.params import LevenbergMarquardtParams as LMP
from tests import problems as PB
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from gtsam_amd import lib
    lib.load()
    return lib


def _cases():
    g = load_golden("dubrovnik_3_7")
    yield "dubrovnik_timesfm", PB.dubrovnik_timesfm(g), {k[len("timesfm_"):]: v for k, v in g.items() if k.startswith("timesfm_")}
    yield "dubrovnik_sfmex", PB.dubrovnik_sfmexample(g), {k[len("sfmex_"):]: v for k, v in g.items() if k.startswith("sfmex_")}
    for name, mk in PB.SYNTH.items():
        yield name, mk(), load_golden(name)


CASES = list(_cases())


@pytest.mark.parametrize("name,pv,gold", CASES, ids=[c[0] for c in CASES])
def test_error_and_jacobians_vs_reference_golden(gpu, name, pv, gold):
    from oracle import gtsam_oracle as O
    p, v0 = pv
    dev = gpu.DeviceGraph(p)
    dev.set_values(v0)
    e = dev.error()
    assert abs(e - float(gold["error"])) <= 1e-9 * abs(float(gold["error"]))
    assert abs(e - O.error(p, v0)) <= 1e-9 * abs(e)
    dev.linearize()
    for ft in range(4):
        key = f"jac{ft}"
        if key in gold:
            J = dev.jacobians(ft)
            assert J.shape == gold[key].shape
            assert rel(J, gold[key]) <= 1e-12, (name, ft)
            assert rel(J, O.jacobians_flat(p, v0, ft)) <= 1e-12
    hd = dev.hessian_diagonal()
    assert rel(hd, gold["hessian_diagonal"]) <= 1e-10
    H, g, _ = O.hessian_dense(p, v0)
    assert rel(dev.gradient(), g) <= 1e-10
    dev.close()


@pytest.mark.parametrize("name,pv,gold", CASES, ids=[c[0] for c in CASES])
def test_damped_solve_vs_reference_golden(gpu, name, pv, gold):
    p, v0 = pv
    dev = gpu.DeviceGraph(p)
    dev.set_values(v0)
    dev.linearize()
    for i in range(2):
        lam, dd = float(gold[f"solve{i}_lambda"]), bool(gold[f"solve{i}_diag"])
        rc, out = dev.try_lambda(lam, dd)
        assert rc == int(gold[f"solve{i}_status"])
        if rc:
            continue
        d = dev.delta()
        assert rel(d, gold[f"solve{i}_delta"]) <= 1e-7, (name, i, rel(d, gold[f"solve{i}_delta"]))
        le = gold[f"solve{i}_linerr"]
        assert abs(out[0] - le[0]) <= 1e-9 * abs(le[0])
        assert abs(out[1] - le[1]) <= 1e-7 * max(abs(le[1]), 1e-12 * abs(le[0]))
        assert rel(dev.trial_values(), gold[f"solve{i}_retract"]) <= 1e-7
        te = float(gold[f"solve{i}_trial_error"])
        if out[0] - out[1] >= 0:
            assert abs(out[2] - te) <= 1e-6 * abs(te)
    dev.close()


LM_CASES = [
    ("dubrovnik_timesfm", "ceres", "timesfm_"), ("dubrovnik_default", "legacy", "default_"),
    ("dubrovnik_sfmex", "legacy", "sfmex_"),
    ("posegraph_small", "legacy", ""), ("posegraph_bigrot", "legacy", ""),
    ("bal_small_unit", "ceres", ""), ("bal_small_iso", "ceres", ""),
] + [(n, "legacy", "") for n in PB.ROBUST_SYNTH]      # noiseModel::Robust, all eight m-estimators of the path


@pytest.mark.parametrize("name,preset,prefix", LM_CASES, ids=[c[0] for c in LM_CASES])
def test_lm_trajectory_vs_reference_golden(gpu, name, preset, prefix):
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt
    if name.startswith("dubrovnik") and name not in PB.SYNTH:
        g = load_golden("dubrovnik_3_7")
        p, v0 = PB.dubrovnik_sfmexample(g) if name == "dubrovnik_sfmex" else PB.dubrovnik_timesfm(g)
    else:
        g = load_golden(name)
        p, v0 = PB.SYNTH[name]()
    params = LMP.CeresDefaults() if preset == "ceres" else LMP()
    opt = DeviceLevenbergMarquardt(p, v0, params)
    opt.optimize()
    ref_trace = g[prefix + "trace"]
    tr = np.array(opt.trace)[:, :3]
    # final error (golden literal of tests/testGeneralSFMFactorB.cpp:44-63 is 0.0199833 +- 1e-5 for dubrovnik_default)
    assert abs(tr[-1, 1] - ref_trace[-1, 1]) <= 1e-6 * abs(ref_trace[-1, 1]) + 1e-12, (tr[-1], ref_trace[-1])
    if name == "dubrovnik_default":
        assert abs(opt.error() - 0.0199833) < 1e-5
    # identical accept/reject sequence: same number of outer rows and same inner-iteration counters
    assert tr.shape == ref_trace.shape, (tr.shape, ref_trace.shape)
    assert np.array_equal(tr[:, 0], ref_trace[:, 0])
    assert rel(tr[:, 1], ref_trace[:, 1]) <= 1e-6
    assert np.allclose(tr[:, 2], ref_trace[:, 2], rtol=1e-6, atol=0)
    assert rel(opt.values_packed(), g[prefix + ("values" if prefix else "final_values")]) <= 1e-5


@pytest.mark.parametrize("name", ["pose2_w100", "pose2_toy"])
def test_pose2_graph_vs_reference_golden(gpu, name):
    """BetweenFactor<Pose2> / PriorFactor<Pose2>: probes and the full LM trajectory against the real reference."""
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt
    gold = load_golden(name)
    pv = PB.pose2_graph(gold)
    test_error_and_jacobians_vs_reference_golden(gpu, name, pv, gold)
    test_damped_solve_vs_reference_golden(gpu, name, pv, gold)
    opt = DeviceLevenbergMarquardt(pv[0], pv[1], LMP())
    opt.optimize()
    tr = np.array(opt.trace)[:, :3]
    assert tr.shape == gold["trace"].shape and np.array_equal(tr[:, 0], gold["trace"][:, 0])
    assert rel(tr[:, 1], gold["trace"][:, 1]) <= 1e-6 and np.allclose(tr[:, 2], gold["trace"][:, 2], rtol=1e-6)
    assert rel(opt.values_packed(), gold["final_values"]) <= 1e-5


def test_pose2_w20000_full_trajectory(gpu):
    """BASELINE configs[0] on the GPU: w20000.txt (20 061 Pose2, reduced system 60 183 x 60 183, < 2 % of the tiles stored
    after RCM), legacy LM: 32 626 834.02 -> 13 520 404.4, then lambda is exhausted (24 inner iterations), exactly as
    the reference (6.4 s there)."""
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt
    g = load_golden("pose2_w20000")
    p, v0 = PB.pose2_graph(g)
    dev = gpu.DeviceGraph(p)
    dev.set_values(v0)
    assert abs(dev.error() - float(g["error"])) <= 1e-9 * float(g["error"])
    dev.linearize()
    assert rel(dev.jacobians(2)[:512], g["jac2_head"]) <= 1e-12
    dev.close()
    opt = DeviceLevenbergMarquardt(p, v0, LMP())
    opt.optimize()
    tr = np.array(opt.trace)[:, :3]
    assert tr.shape == g["trace"].shape and np.array_equal(tr[:, 0], g["trace"][:, 0]), (tr, g["trace"])
    assert rel(tr[:, 1], g["trace"][:, 1]) <= 1e-6 and np.allclose(tr[:, 2], g["trace"][:, 2], rtol=1e-6)
    assert rel(opt.values_packed(), g["final_values"]) <= 1e-4


_ND_CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from gtsam_amd import lib as L
from gtsam_amd.optimizer import DeviceLevenbergMarquardt
from gtsam_amd.params import LevenbergMarquardtParams as LMP
from tests import problems as PB
from tests.conftest import load_golden
sched, depth = %(sched)r, %(depth)d
rel = lambda a, b: float(np.abs(np.asarray(a, float) - np.asarray(b, float)).max() / max(np.abs(np.asarray(b, float)).max(), 1e-300))
g = load_golden("sphere2500")
p, v0 = PB.sphere2500(g)
dev = L.DeviceGraph(p)
pl = dev.df_plan()
assert pl["active"] == (sched == "dataflow")
if sched == "dataflow":
    n_wg = len(pl["chain_off"]) - 1
    assert (n_wg == 2) if depth == 0 else (2 < n_wg <= 32), n_wg     # (<= 16 chain slots of two workgroups since round 6)
dev.set_values(v0)
dev.linearize()
rc, out = dev.try_lambda(1e-5, False)
assert rc == 0 and rel(dev.delta(), g["solve_delta"]) <= 1e-6
assert dev.df_ctrl()[15] == 0 if sched == "dataflow" else True        # no lambda try had to be repeated
dev.close()
opt = DeviceLevenbergMarquardt(p, v0, LMP())
opt.optimize()
tr = np.array(opt.trace)[:, :3]
assert tr.shape == g["trace"].shape and np.array_equal(tr[:, 0], g["trace"][:, 0])
assert rel(tr[:, 1], g["trace"][:, 1]) <= 1e-6
print("ND_CHILD_OK")
"""


@pytest.mark.parametrize("sched,depth", [("dataflow", 2), ("dataflow", 3), ("dataflow", 4), ("dataflow", 0), ("streams", 2)])
def test_nested_dissection_schedules_are_equivalent(gpu, sched, depth):
    """Elimination-tree parallelism (the reference eliminates independent cliques concurrently, inference/ClusterTree-inst.h:218-317):
    a nested-dissection ordering (parts aligned to 256-column pairs, identity padding between them) gives the tile Cholesky
    independent parts.  dataflow: the parts are several diagonal chains inside the two persistent kernels (the default for
    sparse pose graphs: 4 levels since round 4, on up to 16 chain slots since round 6; chol_dataflow.hip::build_df_plan); streams: the round-1 tree schedule (chains on their own
    streams, cross-part updates on one in-order stream); depth 0: one chain (RCM).  Every variant: the reference's damped solve
    and the reference's full LM trajectory on sphere2500.

    Every variant runs in its own process (the switches are read once per process) under a bound: on the last GPU session of round 4
    the in-process form of the `streams` variant -- the only user of the round-1 tree schedule's nine streams -- did not return
    within 19 minutes, once, after more than sixty clean runs (profiles/r04_streams_tree_hang.txt).  A variant that does not
    finish FAILS here after five minutes instead of holding the box."""
    import subprocess
    env = dict(os.environ)
    env["GTG_ND_DEPTH"] = str(depth)
    if sched == "streams":
        env["GTG_CHOL"] = "streams"
    try:
        r = subprocess.run([sys.executable, "-c", _ND_CHILD % {"root": ROOT, "sched": sched, "depth": depth}], env=env, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired as e:
        pytest.fail(f"the {sched} schedule at nested-dissection depth {depth} did not finish within 300 s: {str(e.stderr)[-2000:]}")
    assert r.returncode == 0 and "ND_CHILD_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_backward_sweep_equals_the_per_row_launches(gpu, monkeypatch):
    """The one-launch backward sweep (k_bwd_sweep: left-looking, dependencies polled on the entries of x) against the round-1 form
    (GTG_BWD=steps: one launch per block row, right-looking): the same triangular solve in another association order."""
    for name in ("sphere2500", "bal_60"):
        if name == "sphere2500":
            p, v0 = PB.sphere2500(load_golden("sphere2500")); lam, dd = 1e-5, False
        else:
            from gtsam_amd import datasets as D
            from gtsam_amd.problem import bal_problem
            p, v0 = bal_problem(*D.synthetic_bal(60, 6000, seed=7)); lam, dd = 1e-4, True
        res = []
        for mode in (None, "steps"):
            if mode:
                monkeypatch.setenv("GTG_BWD", mode)
            else:
                monkeypatch.delenv("GTG_BWD", raising=False)
            dev = gpu.DeviceGraph(p)
            dev.set_values(v0); dev.linearize()
            rc, out = dev.try_lambda(lam, dd)
            assert rc == 0
            res.append((dev.delta().copy(), out.copy()))
            dev.close()
        monkeypatch.delenv("GTG_BWD", raising=False)
        assert rel(res[0][0], res[1][0]) <= 1e-11 and np.allclose(res[0][1], res[1][1], rtol=1e-11)


def test_orderings_give_the_same_step(gpu, monkeypatch):
    """The elimination order changes the fill and the schedule, never the step: a caller's ordering
    (gtg_set_reduced_ordering, the reference's params.ordering / Ordering argument of the optimizer's constructor,
    NonlinearOptimizerParams.h:108) -- here the reverse and a random permutation -- and the library's minimum-degree
    alternative (GTG_ORDERING=mindegree) against the default band ordering and the reference's golden step."""
    g = load_golden("sphere2500")
    p, v0 = PB.sphere2500(g)

    def step(**kw):
        dev = gpu.DeviceGraph(p, **kw)
        dev.set_values(v0)
        dev.linearize()
        rc, out = dev.try_lambda(1e-5, False)
        d, fl = dev.delta().copy(), (dev.cholesky_flops_block_level(), dev.cholesky_flops())   # (block level, over the stored tiles)
        dev.close()
        assert rc == 0
        return d, out[2], fl

    d0, e0, f0 = step()
    assert rel(d0, g["solve_delta"]) <= 1e-6
    n = 2500
    for order in (np.arange(n - 1, -1, -1), np.random.default_rng(5).permutation(n)):
        d, e, f = step(reduced_ordering=order.astype(np.int32))
        assert rel(d, d0) <= 1e-7 and abs(e - e0) <= 1e-9 * abs(e0), (rel(d, d0), e, e0)
        assert f != f0            # the order was really used: a different fill
    monkeypatch.setenv("GTG_ORDERING", "mindegree")
    d, e, f = step()
    assert rel(d, d0) <= 1e-7 and abs(e - e0) <= 1e-9 * abs(e0) and f != f0
    f_md = f
    monkeypatch.setenv("GTG_ORDERING", "rcm")            # (explicitly one band: the default for this graph is nested dissection)
    d, e, f_rcm = step()
    assert rel(d, d0) <= 1e-7 and abs(e - e0) <= 1e-9 * abs(e0)
    # "auto" keeps whichever of the two needs fewer flops over the 128 x 128 tiles it stores -- what the kernels execute; the
    # block-level count misleads (street-network BAL shape: minimum degree 42 against 57 GFLOP at block level, 1 086 against 85 over tiles)
    monkeypatch.setenv("GTG_ORDERING", "auto")
    d, e, f = step()
    assert rel(d, d0) <= 1e-7 and f[1] == min(f_rcm[1], f_md[1]), (f, f_rcm, f_md)


@pytest.mark.parametrize("name", ["dubrovnik_sfmex", "bal_small_iso", "posegraph_small", "projection_small", "pose2_w100"])
def test_pcg_solver_vs_oracle_and_direct(gpu, name):
    """gtg_try_lambda_pcg (block-Jacobi PCG on the implicit Schur complement) against the oracle's restatement of the
    reference's preconditionedConjugateGradient on the explicit Schur complement: same algorithm, so the same step and
    (within one) the same iteration count; with tight tolerances both equal the Cholesky step."""
    from oracle import gtsam_oracle as O
    if name == "dubrovnik_sfmex":
        p, v0 = PB.dubrovnik_sfmexample(load_golden("dubrovnik_3_7"))
    elif name == "pose2_w100":
        p, v0 = PB.pose2_graph(load_golden(name))
    else:
        p, v0 = PB.SYNTH[name]()
    dev = gpu.DeviceGraph(p)
    dev.set_values(v0)
    dev.linearize()
    # (the undamped-diagonal system of the BAL shape at lambda = 1e-3 is indefinite for the direct solver too: Levenberg there)
    for lam, dd, er, ea in ((1e-3, name == "bal_small_iso", 1e-3, 1e-3), (1e-4, True, 1e-13, 1e-26)):
        rc, out, its = dev.try_lambda_pcg(lam, dd, max_iterations=3000, epsilon_rel=er, epsilon_abs=ea)
        assert rc == 0
        d = dev.delta()
        info = dict(max_iterations=3000, epsilon_rel=er, epsilon_abs=ea)
        st, d_or, H, g, lin = O.solve_damped(p, v0, lam, dd, pcg=info)
        assert st == 0
        scale = max(np.abs(d_or).max(), 1e-300)
        if er > 1e-6:        # loose tolerance: identical stopping rule -> (almost) identical iterate
            assert abs(its - info["iterations"]) <= 1, (its, info["iterations"])
            # (CG stopped at a 1e-3 relative residual on an ill-conditioned system: rounding differences between two
            #  implementations are amplified to that order; the tight case below is the strong check)
            if its != info["iterations"]:   # the stopping test tipped one iteration apart: compare the iterates at the SAME count
                rc, out, its = dev.try_lambda_pcg(lam, dd, max_iterations=info["iterations"], epsilon_rel=1e-30, epsilon_abs=1e-300)
                assert rc == 0 and its == info["iterations"]
                d = dev.delta()
            assert np.abs(d - d_or).max() <= 5e-3 * scale
        else:                # tight: both are the direct solution (to the conditioning of the BAL shape: 5e-7 in the oracle)
            assert abs(its - info["iterations"]) <= 3, (its, info["iterations"])
            rc2, out2 = dev.try_lambda(lam, dd)
            assert rc2 == 0 and np.abs(d - dev.delta()).max() <= 1e-5 * scale and np.abs(d_or - dev.delta()).max() <= 1e-5 * scale
    dev.close()


def test_pcg_lm_trajectory_matches_direct(gpu):
    """LM with linearSolverType = Iterative (tight CG tolerances) walks the same accept/reject sequence as with the
    Cholesky solve: dubrovnik-3-7, tests/testGeneralSFMFactorB.cpp protocol -> 0.0199833."""
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt
    from gtsam_amd.params import PCGSolverParameters
    g = load_golden("dubrovnik_3_7")
    p, v0 = PB.dubrovnik_timesfm(g)
    prm = LMP(); prm.linearSolverType = "Iterative"; prm.iterativeParams = PCGSolverParameters(3000, 1, 1e-12, 1e-24)
    opt = DeviceLevenbergMarquardt(p, v0, prm)
    opt.optimize()
    tr = np.array(opt.trace)[:, :3]
    ref_tr = g["default_trace"]
    assert tr.shape == ref_tr.shape and np.array_equal(tr[:, 0], ref_tr[:, 0])
    assert rel(tr[:, 1], ref_tr[:, 1]) <= 1e-5 and abs(opt.error() - 0.0199833) < 1e-5


def test_robust_loss_literal(gpu):
    p, v = PB.robust_prior_literal()
    dev = gpu.DeviceGraph(p)
    dev.set_values(v)
    assert abs(dev.error() - 0.49505) < 1e-5                       # tests/testRobust.cpp:45-47
    dev.close()


def test_sphere2500_solve_and_full_trajectory(gpu):
    """configs[3]: sphere2500 pose graph (tile-sparse path, 8 % of the tiles stored).  One damped solve on identical
    inputs, then the full LM run against the reference's golden trace."""
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt
    g = load_golden("sphere2500")
    p, v0 = PB.sphere2500(g)
    dev = gpu.DeviceGraph(p)
    dev.set_values(v0)
    e0 = dev.error()
    assert abs(e0 - float(g["error0"])) <= 1e-9 * e0
    dev.linearize()
    rc, out = dev.try_lambda(1e-5, False)
    assert rc == int(g["solve_status"]) == 0
    assert rel(dev.delta(), g["solve_delta"]) <= 1e-6
    assert abs(out[1] - g["solve_linerr"][1]) <= 1e-6 * abs(g["solve_linerr"][1])
    dev.close()
    # the whole run of BASELINE.md's golden trace: 21 outer / 45 inner iterations, 12 280 978.77 -> 1 136.95214, every
    # accept / reject decision and every lambda as the reference's (from outer 12 on several lambdas are rejected per
    # iteration and the decisions are FP-marginal, BASELINE.md -- the sequences coincide nevertheless).
    opt = DeviceLevenbergMarquardt(p, v0, LMP())
    opt.optimize()
    tr = np.array(opt.trace)[:, :3]
    ref_trace = g["trace"]
    assert tr.shape == ref_trace.shape and np.array_equal(tr[:, 0], ref_trace[:, 0])      # all 21 outer / 45 inner iterations
    assert rel(tr[:, 1], ref_trace[:, 1]) <= 1e-6 and np.allclose(tr[:, 2], ref_trace[:, 2], rtol=1e-6)
    assert abs(opt.error() - ref_trace[-1, 1]) <= 1e-6 * ref_trace[-1, 1]
    assert abs(opt.error() - 1136.95214) < 0.05
    print("sphere2500 outer/inner:", opt.iterations(), opt.getInnerIterations(), "reference:", int(g["iterations"]), int(ref_trace[-1, 0]))


def test_two_handles_on_two_host_threads(gpu):
    """The factorisation schedule's streams and events belong to the handle: two optimizers driven concurrently from
    two host threads (ctypes releases the GIL inside the library) walk exactly the single-thread trajectories."""
    import threading
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt
    g = load_golden("sphere2500")
    p, v0 = PB.sphere2500(g)

    def run(out, i):
        opt = DeviceLevenbergMarquardt(p, v0, LMP())
        opt.optimize()
        out[i] = np.array(opt.trace)[:, :3]

    single = [None]
    run(single, 0)
    both = [None, None]
    threads = [threading.Thread(target=run, args=(both, i)) for i in range(2)]
    for t in threads: t.start()
    for t in threads: t.join()
    for tr in both:
        assert tr is not None and tr.shape == single[0].shape and np.array_equal(tr, single[0])
    assert abs(single[0][-1, 1] - 1136.95214) < 0.05


@pytest.mark.parametrize("n", [5, 100, 128, 300, 1000])
def test_dense_cholesky_vs_lapack(gpu, n):
    """The frontal kernel alone: L L^T reconstruction 1e-9 like gtsam/base/tests/testCholesky.cpp:26-67,
    solution against numpy; plus the failure semantics of base/cholesky.cpp:124-127 (non-PD -> indeterminate)."""
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n + 7)); A = A @ A.T + 1e-3 * np.eye(n)
    b = rng.normal(size=n)
    from gtsam_amd.problem import Problem
    dev = gpu.DeviceGraph(Problem(var_type=np.array([0], np.int32)))
    rc, L, x = dev.dense_cholesky(A, b)
    assert rc == 0
    assert rel(L @ L.T, A) <= 1e-12
    assert rel(L, np.linalg.cholesky(A)) <= 1e-9
    assert rel(x, np.linalg.solve(A, b)) <= 1e-8
    A2 = A.copy(); A2[n // 2, n // 2] = -1.0
    rc, _, _ = dev.dense_cholesky(A2, b)
    assert rc == 1
    dev.close()


def test_reference_cholesky_literal(gpu):
    """7x7 literal of gtsam/base/tests/testCholesky.cpp:26-67 (choleskyPartial, R^T R reconstruction 1e-9)."""
    ABC = np.array([[4.0375, 3.4584, 3.5735, 2.4815, 2.1471, 2.7400, 2.2063],
                    [0., 4.7267, 3.8423, 2.3624, 2.8091, 2.9579, 2.5914],
                    [0., 0., 5.1600, 2.0797, 3.4690, 3.2419, 2.9992],
                    [0., 0., 0., 1.8786, 1.0535, 1.4250, 1.3347],
                    [0., 0., 0., 0., 3.0788, 2.6283, 2.3791],
                    [0., 0., 0., 0., 0., 2.9227, 2.4056],
                    [0., 0., 0., 0., 0., 0., 2.5776]])
    A = ABC + np.triu(ABC, 1).T
    from gtsam_amd.problem import Problem
    dev = gpu.DeviceGraph(Problem(var_type=np.array([0], np.int32)))
    rc, L, _ = dev.dense_cholesky(A)
    assert rc == 0
    assert rel(L @ L.T, A) <= 1e-9
    dev.close()


def _underconstrained_literals():
    """gtsam/base/tests/testCholesky.cpp:101-138: A = L D L^T with D1 (last pivot 1e-12: rank test), D2 (zero pivots), D3 (negative)."""
    L = np.array([[1, 0, 0, 0, 0, 0],
                  [1.11177808157954, 1.06204809504665, 0.507342638873381, 1.34953401829486, 1, 0],
                  [0.155864888199928, 1.10933048588373, 0.501255576961674, 1, 0, 0],
                  [1.12108665967793, 1.01584408366945, 1, 0, 0, 0],
                  [0.776164062474843, 0.117617236580373, -0.0236628691347294, 0.814118199972143, 0.694309975328922, 1],
                  [0.1197220685104, 1, 0, 0, 0, 0]])
    d = [0.814723686393179, 0.811780089277421, 1.82596950680844, 0.240287537694585]
    for tail in ([1.34342584865901, 1e-12], [0.0, 0.0], [-0.5, -0.6]):
        yield L @ np.diag(d + tail) @ L.T


def test_underconstrained_and_negative_pivot_literals(gpu):
    """choleskyPartial's failure semantics on the device (base/cholesky.cpp:107-158): a non-positive pivot (Eigen LLT
    NumericalIssue) and the exponent test on the last two diagonal entries of the frontal block (difference >= 12) both give
    `false` in the reference -> GTG_INDETERMINATE here; the matrix is one frontal block, as in the reference's test."""
    from gtsam_amd.problem import Problem
    from oracle import ref
    dev = gpu.DeviceGraph(Problem(var_type=np.array([0], np.int32)))
    for A in _underconstrained_literals():
        if ref.available():
            assert ref.cholesky_partial(A, 6)[0] is False          # the literal's expectation, from the reference itself
        rc, _, _ = dev.dense_cholesky(A)
        assert rc == 1, "under-constrained / indefinite matrix accepted"
    # the same L with a benign diagonal passes, and so does a pivot ratio just inside the bound (2^-11 in R = 2^-22 in D)
    L = next(_underconstrained_literals())  # noqa: F841  (shape only)
    rng = np.random.default_rng(0)
    M = rng.normal(size=(6, 6)); A = M @ M.T + np.eye(6)
    assert dev.dense_cholesky(A)[0] == 0
    D = np.diag([1.0, 1.0, 1.0, 1.0, 1.0, 2.0 ** -20])
    assert dev.dense_cholesky(D)[0] == 0                           # exponent difference 10 < 12
    D = np.diag([1.0, 1.0, 1.0, 1.0, 1.0, 2.0 ** -26])
    assert dev.dense_cholesky(D)[0] == 1                           # exponent difference 13 >= 12
    if ref.available():
        assert ref.cholesky_partial(np.diag([1.0, 1.0, 1.0, 1.0, 1.0, 2.0 ** -20]), 6)[0] is True
        assert ref.cholesky_partial(np.diag([1.0, 1.0, 1.0, 1.0, 1.0, 2.0 ** -26]), 6)[0] is False
    dev.close()


def test_rank_test_fires_at_variable_ends_of_a_graph(gpu):
    """On a graph the test applies at the last pivot of every reduced variable (the finest partition into cliques the reference's
    junction tree can produce).  A pose graph whose last pose is constrained in 5 of its 6 directions only
    (a between factor with one enormous sigma) is under-constrained in exactly the reference's sense: both say indeterminate at
    small lambda, both solve it once the damping lifts the pivot."""
    from gtsam_amd import datasets as D
    from gtsam_amd.problem import NOISE_DIAGONAL
    from oracle import ref
    p, v0 = D.random_pose_graph(8, 0, seed=11)
    # replace the noise of the last chain edge (6 -> 7) by one that leaves the last translation direction free
    weak = p.add_noise(NOISE_DIAGONAL, 6, [0.1, 0.1, 0.1, 0.3, 0.3, 1e9])
    last = int(np.where((p.between_v1 == 6) & (p.between_v2 == 7))[0][0])
    p.between_noise[last] = weak
    dev = gpu.DeviceGraph(p)
    dev.set_values(v0); dev.linearize()
    rc_small, _ = dev.try_lambda(1e-30, False)
    rc_big, _ = dev.try_lambda(1.0, False)
    assert rc_small == 1 and rc_big == 0
    if ref.available():
        g = ref.RefGraph(p)
        assert g.solve(v0, 1e-30, False, ordering_kind=0)[0] == 1
        assert g.solve(v0, 1.0, False, ordering_kind=0)[0] == 0
    dev.close()


def test_full_size_properties(gpu):
    """BASELINE.json's full size (Ladybug-1723 shape): size-independent properties instead of an oracle run.
    (1) error(values) is reproducible bit-for-bit run to run (no FP atomics);
    (2) linear.error(0) == graph.error(values) (b = -whitened residual);
    (3) the damped step decreases the linearised cost, and |A delta - b|^2 is consistent with the
        normal equations:  L(0) - L(delta) == 0.5 * delta^T (g + lambda D delta) within roundoff;
    (4) retract(values, 0-step) is the identity: a huge lambda gives ||delta|| -> 0 and trial error -> error."""
    from gtsam_amd import datasets as D
    from gtsam_amd.problem import bal_problem
    p, v0 = bal_problem(*D.ladybug_1723())
    dev = gpu.DeviceGraph(p)
    dev.set_values(v0)
    e1 = dev.error(); e2 = dev.error()
    assert e1 == e2
    dev.linearize()
    rc, out = dev.try_lambda(1e-4, True)
    assert rc == 0
    assert abs(out[0] - e1) <= 1e-12 * e1
    assert out[1] < out[0]
    d = dev.delta(); g = dev.gradient(); hd = dev.hessian_diagonal()
    lamD = 1e-4 * np.clip(hd, 1e-6, 1e32)
    lhs = out[0] - out[1]
    rhs = 0.5 * float(d @ (g + lamD * d))
    assert abs(lhs - rhs) <= 1e-7 * abs(lhs), (lhs, rhs)
    assert out[2] < e1                       # the step is a real improvement on this problem
    rc, out2 = dev.try_lambda(1e12, True)
    assert rc == 0 and out2[3] < 1e-6 * out[3]
    assert abs(out2[2] - e1) <= 1e-6 * e1
    dev.close()
