"""The linearisation without stored Jacobians (csrc/fused.h, round 5) against the stored-record build of the same sources
(`make -C gtsam_amd/csrc records` -> lib/libgtsam_amd_records.so, GeneralSFM records written by k_lin_sfm and re-read as in rounds 1 - 4).

What must hold, and to what:
  * the recomputed records (gtg_get_jacobians recomputes them on request) are the stored build's records BIT FOR BIT -- a record is a
    pure function of (values, measurement, noise row), evaluated by the same device function;
  * the landmark-side sums (k_lm_fused adds a landmark's observations in list order, as k_lm_diag does) and everything that only reads
    records are bit-identical as well, so the gradient's landmark part is;
  * the camera-side sums run over the same summands in another association order (k_cam_fused: 64-factor chunks per wavefront, four
    MFMA rows per step; k_red_diag: the list cut into four), so the Hessian diagonal / gradient of the cameras, the step and the LM
    trace agree to rounding: 1e-12 relative on the sums, 1e-9 on the step, the same accept / reject rows with errors to 1e-7.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd_records.so")

_CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from gtsam_amd import lib as L, datasets as D
from gtsam_amd.optimizer import DeviceLevenbergMarquardt
from gtsam_amd.params import LevenbergMarquardtParams as LMP
from gtsam_amd.problem import bal_problem
out = {"lib": L.LIB_PATH}
for name, (nc, npt, seed) in (("bal300", (300, 20000, 3)), ("bal12", (12, 9000, 5))):      # (bal12: several workgroups per camera, k_cam_combine)
    p, v0 = bal_problem(*D.synthetic_bal(nc, npt, seed=seed))
    prm = LMP.CeresDefaults(); prm.setMaxIterations(6)
    opt = DeviceLevenbergMarquardt(p, v0, prm)
    e0 = opt.dev.error()
    opt.dev.linearize()
    J = opt.dev.jacobians(0)
    g = opt.dev.gradient(); hd = opt.dev.hessian_diagonal()
    rc, o = opt.dev.try_lambda(1e-3, prm.diagonalDamping)
    d = opt.dev.delta()
    opt.optimize()
    tr = np.array(opt.trace)[:, :3]
    np.savez(%(out)r + "_" + name + ".npz", J=J, g=g, hd=hd, d=d, o=np.array(o[:4]), tr=tr, e0=e0, rc=rc, is_lm=np.array(p.var_type) == 1)
print("RESULT " + json.dumps(out))
'''


def _child(lib_path, out):
    env = dict(os.environ)
    if lib_path:
        env["GTSAM_AMD_LIB"] = lib_path
    else:
        env.pop("GTSAM_AMD_LIB", None)
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT, "out": out}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_fused_linearisation_equals_stored_records(tmp_path):
    import torch
    assert torch.cuda.is_available()
    if not os.path.exists(RECORDS):
        pytest.skip("lib/libgtsam_amd_records.so not built (make -C gtsam_amd/csrc records)")
    a = _child(None, str(tmp_path / "fused"))
    b = _child(RECORDS, str(tmp_path / "records"))
    assert a["lib"].endswith("libgtsam_amd.so") and b["lib"].endswith("libgtsam_amd_records.so")
    for name in ("bal300", "bal12"):
        F = np.load(str(tmp_path / "fused") + "_" + name + ".npz"); R = np.load(str(tmp_path / "records") + "_" + name + ".npz")
        assert F["e0"] == R["e0"] and int(F["rc"]) == int(R["rc"]) == 0
        assert np.array_equal(F["J"], R["J"])                                             # the records themselves: bit for bit
        scale_g = np.abs(R["g"]).max(); scale_h = np.abs(R["hd"]).max()
        assert np.abs(F["g"] - R["g"]).max() <= 1e-12 * scale_g and np.abs(F["hd"] - R["hd"]).max() <= 1e-12 * scale_h
        assert np.abs(F["d"] - R["d"]).max() <= 1e-9 * max(1.0, np.abs(R["d"]).max())
        assert np.allclose(F["o"], R["o"], rtol=1e-10, atol=0.0)
        assert F["tr"].shape == R["tr"].shape and np.array_equal(F["tr"][:, 0], R["tr"][:, 0])          # same rows, same lambdas
        assert (np.abs(F["tr"][:, 1] - R["tr"][:, 1]) <= 1e-7 * np.abs(R["tr"][:, 1])).all()
