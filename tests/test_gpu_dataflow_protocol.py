"""GPU: the publication protocol of the dataflow Cholesky (csrc/chol_dataflow.hip) -- two persistent kernels that exchange
tiles through HBM while they run.

1. Repeatability under contention (was tools/df_stress.py): the same optimisation run alone and from three host threads at
   once (three handles competing for the device; the library serialises their factorisations per device), several rounds:
   every trace must be BIT-identical to the first one -- a stale read, a lost flag or a schedule-dependent sum would show
   as a difference, a starved chain as a time-out error.
2. A/B of the fence-free publication (write-through stores + flags, the default) against the textbook protocol
   (release / acquire fences at agent scope, -DGTG_DF_FENCES=1, built as lib/libgtsam_amd_fences.so by the same Makefile):
   the two builds run the same sums in the same order, so LM traces, steps and errors must be bit-identical.
"""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FENCED = os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd_fences.so")


def _sphere():
    from tests import problems as PB
    from tests.conftest import load_golden
    from gtsam_amd.params import LevenbergMarquardtParams as LMP
    p, v0 = PB.sphere2500(load_golden("sphere2500"))
    return p, v0, LMP()


def _bal300():
    from gtsam_amd import datasets as D
    from gtsam_amd.params import LevenbergMarquardtParams as LMP
    from gtsam_amd.problem import bal_problem
    p, v0 = bal_problem(*D.synthetic_bal(300, 20000, seed=3))
    prm = LMP.CeresDefaults(); prm.setMaxIterations(6)
    return p, v0, prm


@pytest.mark.parametrize("make", [_sphere, _bal300], ids=["sphere2500", "bal300"])
def test_dataflow_repeatable_under_contention(make):
    import torch
    assert torch.cuda.is_available()
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt
    p, v0, prm = make()

    def run(out, i):
        try:
            opt = DeviceLevenbergMarquardt(p, v0, prm)
            opt.optimize()
            out[i] = (np.array(opt.trace)[:, :3], opt.values_packed())
            opt.dev.close()
        except Exception as e:  # noqa: BLE001
            out[i] = e

    ref = [None]
    run(ref, 0)
    assert not isinstance(ref[0], Exception), ref[0]
    # Strict bit-identity of all nine runs with the undisturbed one.  (Round 3 allowed one of nine to differ: the cause was a real
    # ordering bug -- the diagonal tile's progress word could overtake the write-through operand images of its panel,
    # chol_device.h::potrf_body -- not an unexplained cache effect; see profiles/r04_df_handoff.txt.)
    for rnd in range(3):
        res = [None, None, None]
        th = [threading.Thread(target=run, args=(res, i)) for i in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join(600)
        for r in res:
            assert not isinstance(r, Exception), (rnd, r)
            assert r[0].shape == ref[0][0].shape and np.array_equal(r[0], ref[0][0]), (rnd, r[0], ref[0][0])
            assert np.array_equal(r[1], ref[0][1]), rnd


_CHILD = r'''
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from gtsam_amd import lib as L
from gtsam_amd.optimizer import DeviceLevenbergMarquardt
from tests.test_gpu_dataflow_protocol import _sphere, _bal300
out = {"lib": L.LIB_PATH}
for name, make in (("sphere2500", _sphere), ("bal300", _bal300)):
    p, v0, prm = make()
    opt = DeviceLevenbergMarquardt(p, v0, prm)
    opt.dev.linearize()
    rc, o = opt.dev.try_lambda(1e-3, prm.diagonalDamping)
    d = opt.dev.delta()
    opt.optimize()
    tr = np.array(opt.trace)[:, :3]
    out[name] = dict(rc=int(rc), scalars=np.array(o[:3]).tobytes().hex(), delta=hashlib.sha256(d.tobytes()).hexdigest(),
                     trace=tr.tobytes().hex(), rows=int(tr.shape[0]), final=float(tr[-1, 1]),
                     values=hashlib.sha256(opt.values_packed().tobytes()).hexdigest(), df=bool(opt.dev.df_plan()["active"]))
print("RESULT " + json.dumps(out))
'''


def _child(lib_path):
    env = dict(os.environ)
    if lib_path:
        env["GTSAM_AMD_LIB"] = lib_path
    else:
        env.pop("GTSAM_AMD_LIB", None)
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_fenced_protocol_gives_identical_numbers():
    import torch
    assert torch.cuda.is_available()
    assert os.path.exists(FENCED), "lib/libgtsam_amd_fences.so not built (make -C gtsam_amd/csrc)"
    a = _child(None)
    b = _child(FENCED)
    assert a["lib"].endswith("libgtsam_amd.so") and b["lib"] == FENCED
    for name in ("sphere2500", "bal300"):
        assert a[name]["df"] and b[name]["df"], "both builds must run the dataflow schedule"
        assert a[name]["rc"] == b[name]["rc"] == 0
        for k in ("scalars", "delta", "trace", "values"):
            assert a[name][k] == b[name][k], (name, k, a[name]["final"], b[name]["final"])


_CHILD_FALLBACK = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from gtsam_amd.optimizer import DeviceLevenbergMarquardt
from tests.test_gpu_dataflow_protocol import _sphere
p, v0, prm = _sphere()
opt = DeviceLevenbergMarquardt(p, v0, prm)
opt.optimize()
tr = np.array(opt.trace)[:, :3]
print("RESULT " + json.dumps(dict(trace=tr.tobytes().hex(), rows=int(tr.shape[0]), final=float(tr[-1, 1]),
                                  fallbacks=int(opt.dev.df_ctrl()[15]))))
'''


@pytest.mark.parametrize("nd", ["0", None], ids=["one_chain", "default_ordering"])
def test_a_timed_out_dataflow_pass_is_repeated(nd):
    """A dependency wait of the dataflow factorisation that runs into its bound (here: the chain kernel is left out of the third
    factorisation, GTG_DF_TEST_TIMEOUT=3 -- what a chain kernel that the dispatcher never placed looks like) must not abort
    optimize(): the lambda try is computed once more, first with the SAME schedule (the repeat returns the bits of an undisturbed
    try, whatever the ordering: the LM trajectory does not depend on whether a wait timed out), and the handle counts one repeat.
    When the repeat times out as well (GTG_DF_TEST_TIMEOUT=3:2 leaves the chain kernel out of two factorisations in a row) the try
    goes to the stream / event schedule.  With one chain (GTG_ND_DEPTH=0) that schedule runs the same sums in the same order, so
    the trajectory is still bit for bit the undisturbed one; with the default ordering of this graph (nested dissection, several
    chains) it sums the cross-part updates in a different order: same rows, same accept / reject decisions, errors equal to 1e-6
    (the tolerance of the LM-trace comparisons with the reference)."""
    import torch
    assert torch.cuda.is_available()

    def child(extra_env):
        env = dict(os.environ); env.pop("GTSAM_AMD_LIB", None); env.pop("GTG_ND_DEPTH", None); env.update(extra_env)
        if nd is not None:
            env["GTG_ND_DEPTH"] = nd
        r = subprocess.run([sys.executable, "-c", _CHILD_FALLBACK % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        return json.loads(line[len("RESULT "):]), r.stderr
    a, _ = child({})
    b, err = child({"GTG_DF_TEST_TIMEOUT": "3"})
    assert a["fallbacks"] == 0 and b["fallbacks"] == 1, (a["fallbacks"], b["fallbacks"])
    assert "repeating the lambda try with the dataflow schedule" in err
    assert a["trace"] == b["trace"], (a["final"], b["final"], a["rows"], b["rows"])
    c, err = child({"GTG_DF_TEST_TIMEOUT": "3:2"})
    assert c["fallbacks"] == 2, c["fallbacks"]
    assert "repeating the lambda try with the stream schedule" in err
    ta = np.frombuffer(bytes.fromhex(a["trace"]), np.float64).reshape(-1, 3)
    tc = np.frombuffer(bytes.fromhex(c["trace"]), np.float64).reshape(-1, 3)
    assert ta.shape == tc.shape and np.array_equal(ta[:, 0], tc[:, 0]), (ta, tc)
    if nd == "0":
        assert a["trace"] == c["trace"], (a["final"], c["final"], a["rows"], c["rows"])
    else:
        # (the tolerance of every LM-trace comparison with the reference: rounding differences of one solve are amplified along the run)
        assert (np.abs(ta[:, 1] - tc[:, 1]) <= 1e-6 * np.abs(ta[:, 1])).all() and np.allclose(ta[:, 2], tc[:, 2], rtol=1e-6), (ta, tc)
