"""Host side of gtg_upload_problem (symbolic analysis, schedule construction, table uploads) without a GPU.

The product library runs in a child process under tools/hipstub -- a dry-run HIP runtime in which "device" memory is host
memory and kernels do not run -- so only the HOST code executes; nothing numeric is computed there (it is not a CPU
fallback and the product never loads it).  The stub records a hash of every host-to-device copy; the sorted records are
the "signature" of everything the analysis built: factor tables, CSR incidence lists, the Schur block/term lists
(the summation order of the device), the RCM order and the tile schedule of the Cholesky.

Checked here: the analysis is deterministic and independent of the number of host threads (the device sums follow the
uploaded term order, so this is what keeps results bit-reproducible), also per shard and for a landmark seen twice by
one camera (the general path of the term generator).
"""
import copy
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import host_profile as HP  # noqa: E402


def _signature(workload, shards=1, threads=None):
    env = {} if threads is None else {"GTG_HOST_THREADS": str(threads)}
    rec = HP.run(workload, shards=shards, reps=1, env_extra=env, quiet=True)
    return [(r["signature"], r["reduced_dim"], round(r["cholesky_gflop"], 9), r["h2d_bytes"]) for r in rec["runs"]]


@pytest.fixture(scope="module")
def stub():
    if not os.path.exists(os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd.so")):
        pytest.skip("libgtsam_amd.so not built")
    return HP.build_stub()


def test_analysis_independent_of_host_threads(stub):
    w = "bal:60:6000:7"           # 60 cameras, 6000 points: ~26k observations -> more than one analysis thread
    one = _signature(w, threads=1)
    assert one == _signature(w, threads=3) == _signature(w, threads=8)
    assert one[0][1] == 60 * 9


def test_sharded_analysis_independent_of_host_threads(stub):
    w = "bal:60:6000:7"
    a = _signature(w, shards=2, threads=1)
    b = _signature(w, shards=2, threads=5)
    assert a == b and len(a) == 2
    assert a[0][0] != a[1][0]            # the two shards own different landmarks
    assert a[0][1] == a[1][1] == 540     # every shard holds the whole reduced system


def test_sharded_smart_graph_upload(stub):
    """Smart factors on a sharded graph: every shard keeps the measurements of the tracks whose hidden landmark it owns (the
    per-factor tables stay global), the reduced system -- the cameras -- and its layout are those of the whole graph on every shard."""
    w = "smart:smart_far_infinity"
    one = HP.run(w, shards=1, reps=1, quiet=True)["runs"]
    two = HP.run(w, shards=2, reps=1, quiet=True)["runs"]
    assert len(two) == 2 and two[0]["signature"] != two[1]["signature"]
    assert one[0]["reduced_dim"] == two[0]["reduced_dim"] == two[1]["reduced_dim"] == 90
    assert one[0]["structure_hash"] == two[0]["structure_hash"] == two[1]["structure_hash"]
    assert two[0]["h2d_bytes"] < one[0]["h2d_bytes"] and two[1]["h2d_bytes"] < one[0]["h2d_bytes"]


def test_pose_graphs_and_fixture_sizes(stub):
    s = _signature("sphere2500", threads=2)
    assert s[0][1] == 2500 * 6
    assert s == _signature("sphere2500", threads=1)
    d = _signature("dubrovnik_3_7")
    assert d[0][1] == 27


def test_camera_seeing_a_landmark_twice(stub):
    # duplicate some observations: the same (camera, landmark) pair twice exercises the diagonal-block terms
    # (E_a E_b^T and E_b E_a^T) of the generator; must not depend on the thread count either
    assert _signature("baldup:40:3000:3", threads=1) == _signature("baldup:40:3000:3", threads=4)


def test_world_size_2_gloo_sharded_upload_with_the_real_library(stub):
    """The N > 1 set-up path end to end on CPU: two `gloo` ranks run the library's sharded gtg_upload_problem under the
    stub with the REAL all-reduce callback (gtsam_amd.distributed.make_allreduce).  Every shard only has the Schur
    blocks of its own landmarks, but ordering, offsets, padding, tile schedule and exchange list are derived from the
    whole graph: both ranks must report the layout of the single-handle run, and the library's own consistency
    exchange must pass.  (A layout derived from the shard's own blocks differs between shards as soon as the reduced
    system is reordered, i.e. from 16 cameras up -- the summed buffers would not line up.)"""
    w = "bal:60:6000:7"
    recs = HP.run_gloo(w, world=2)
    assert [r["ok"] for r in recs] == [True, True]
    single = HP.run(w, shards=1, reps=1, quiet=True)["runs"][0]
    assert recs[0]["structure_hash"] == recs[1]["structure_hash"] == single["structure_hash"]
    assert recs[0]["cholesky_gflop"] == recs[1]["cholesky_gflop"] == single["cholesky_gflop"]
    # the same for a graph of smart factors (each follows its hidden landmark to a shard)
    recs = HP.run_gloo("smart:smart_far_infinity", world=2)
    assert [r["ok"] for r in recs] == [True, True] and recs[0]["structure_hash"] == recs[1]["structure_hash"]


def test_sharded_upload_fails_loudly_when_ranks_and_shards_do_not_match(stub):
    recs = HP.run_gloo("bal:60:6000:7", world=2, claim_shards=3)     # the all-reduce spans 2 ranks, the library is told 3
    assert all(not r["ok"] and "disagree on the layout" in r["error"] for r in recs)


def test_layout_check_runs_before_the_first_exchange_when_the_callback_comes_late(stub):
    ok = HP.run_gloo("bal:60:6000:7", world=2, late_callback=True)
    assert all(r["ok"] for r in ok) and ok[0]["structure_hash"] == ok[1]["structure_hash"]
    bad = HP.run_gloo("bal:60:6000:7", world=2, claim_shards=3, late_callback=True)
    assert all(not r["ok"] and "disagree on the layout" in r["error"] for r in bad)


def test_three_shards_and_pose_graph_layouts(stub):
    recs = HP.run_gloo("sphere2500", world=3)                         # between factors go round-robin
    single = HP.run("sphere2500", shards=1, reps=1, quiet=True)["runs"][0]
    assert all(r["ok"] and r["structure_hash"] == single["structure_hash"] for r in recs)


_PROTOCOL = r'''
import ctypes as C, json
import numpy as np
from gtsam_amd import lib as L
from gtsam_amd.problem import NOISE_UNIT, NOISE_ISOTROPIC, Problem, VAR_POINT3, VAR_POSE3, VAR_SFM_CAMERA, bal_problem
from gtsam_amd import datasets as D
lib = L.load()
out = {}
def err(): return lib.gtg_last_error().decode()
def upload(p, shard=0, n=1):
    h = C.c_void_p(); assert lib.gtg_create(C.byref(h), 0) == 0
    cp = p.to_ctypes(); rc = lib.gtg_upload_problem(h, C.byref(cp), shard, n); e = err()
    return h, rc, e
good, v0 = bal_problem(*D.synthetic_bal(6, 40, seed=1))
# --- call protocol on a good problem ---------------------------------------------------------------
h, rc, e = upload(good); out["upload_ok"] = rc
o4 = np.zeros(4)
out["try_before_linearize"] = [lib.gtg_try_lambda(h, 1e-3, 0, 1e-6, 1e32, o4.ctypes.data), err()]
out["accept_before_try"] = [lib.gtg_accept(h), err()]
bad = np.zeros(3); out["set_values_wrong_size"] = [lib.gtg_set_values(h, bad.ctypes.data, 3), err()]
out["set_values"] = lib.gtg_set_values(h, v0.ctypes.data, v0.size)
out["linearize"] = lib.gtg_linearize(h)
out["lambda_zero"] = [lib.gtg_try_lambda(h, 0.0, 0, 1e-6, 1e32, o4.ctypes.data), err()]
out["gradient_wrong_size"] = [lib.gtg_get_gradient(h, bad.ctypes.data, 3), err()]
out["jacobians_unknown_type"] = [lib.gtg_get_jacobians(h, 9, bad.ctypes.data, 3), err()]
nc = int((good.var_type == VAR_SFM_CAMERA).sum())
order = np.arange(nc, dtype=np.int32)[::-1].copy()
out["reorder_ok"] = lib.gtg_set_reduced_ordering(h, order.ctypes.data, nc)
out["try_after_reorder_needs_linearize"] = [lib.gtg_try_lambda(h, 1e-3, 0, 1e-6, 1e32, o4.ctypes.data), err()]
dup = np.zeros(nc, np.int32); out["reorder_not_permutation"] = [lib.gtg_set_reduced_ordering(h, dup.ctypes.data, nc), err()]
out["destroy"] = lib.gtg_destroy(h); out["destroy_null"] = lib.gtg_destroy(None)
out["values_size_null"] = int(lib.gtg_values_size(None))
# --- content the upload must reject (the reference's exceptions) --------------------------------------
def variant(fn):
    q, _ = bal_problem(*D.synthetic_bal(6, 40, seed=1)); fn(q); hh, rc, e = upload(q); lib.gtg_destroy(hh); return [rc, e]
def wrong_dim(q): q.noise_dim = q.noise_dim.copy(); q.noise_dim[q.sfm_noise[0]] = 3
def bad_key(q): q.sfm_point = q.sfm_point.copy(); q.sfm_point[0] = 10 ** 6
def cam_as_point(q): q.sfm_point = q.sfm_point.copy(); q.sfm_point[0] = q.sfm_cam[0]
def bad_type(q): q.var_type = q.var_type.copy(); q.var_type[0] = 17
def bad_noise(q): q.noise_kind = q.noise_kind.copy(); q.noise_kind[0] = 9
def bad_estimator(q): q.noise_robust = np.full(q.noise_kind.size, 2, np.int32); q.noise_robust_param = np.zeros(q.noise_kind.size)
for name, fn in [("wrong_dim", wrong_dim), ("bad_key", bad_key), ("cam_as_point", cam_as_point), ("bad_type", bad_type),
                 ("bad_noise", bad_noise), ("bad_estimator", bad_estimator)]:
    out[name] = variant(fn)
hh, rc, e = upload(good, shard=2, n=2); out["bad_shard"] = [rc, e]; lib.gtg_destroy(hh)
hh, rc, e = upload(good, shard=0, n=2); lib.gtg_set_values(hh, v0.ctypes.data, v0.size)
out["sharded_without_allreduce"] = [rc, lib.gtg_linearize(hh), err()]; lib.gtg_destroy(hh)
hh = C.c_void_p(); out["bad_device"] = [lib.gtg_create(C.byref(hh), 64), err()]   # the stub shows 8 devices
print("RESULT " + json.dumps(out))
'''


def test_c_abi_protocol_and_error_behaviour(stub):
    """The boundary's error behaviour without a GPU: wrong call order, wrong sizes, content the reference rejects with an
    exception (noise dimension NonlinearFactor.cpp:97-104, unknown key, estimator parameter LossFunctions.cpp) come back as
    GTG_ERR_USAGE (-1) with a message, never as a crash or a silent success; a sharded upload without an all-reduce
    callback is a runtime error (-2)."""
    r = HP.run_snippet(_PROTOCOL)
    assert r["upload_ok"] == 0 and r["set_values"] == 0 and r["linearize"] == 0 and r["reorder_ok"] == 0
    assert r["destroy"] == 0 and r["destroy_null"] == 0 and r["values_size_null"] == -1
    for key, text in [("try_before_linearize", "call gtg_linearize first"), ("accept_before_try", "no trial values"),
                      ("set_values_wrong_size", "wrong size"), ("lambda_zero", "lambda must be > 0"),
                      ("gradient_wrong_size", "wrong size"), ("jacobians_unknown_type", "unknown factor type"),
                      ("try_after_reorder_needs_linearize", "call gtg_linearize first"),
                      ("reorder_not_permutation", "not a permutation"), ("wrong_dim", "NoiseModel has wrong dimension"),
                      ("bad_key", "not in Values"), ("cam_as_point", "keys must be (SFM_CAMERA, POINT3)"),
                      ("bad_type", "unknown variable type"), ("bad_noise", "unsupported noise model kind"),
                      ("bad_estimator", "m-estimator parameter must be > 0"), ("bad_shard", "bad shard"), ("bad_device", "bad device id")]:
        assert r[key][0] == -1 and text in r[key][1], (key, r[key])
    # the callback may be registered after the upload; the first exchange without one is a runtime error (-2)
    assert r["sharded_without_allreduce"][:2] == [0, -2] and "no allreduce callback" in r["sharded_without_allreduce"][2]


def test_degenerate_graphs_through_the_host_path(stub):
    """A single pose, a variable without factors, no factors at all, landmarks only, a landmark with one observation,
    camera counts around the reordering threshold (16) and the tile boundary: upload, analysis, one linearize / try_lambda
    issue sequence and the plan getters run through (the same driver runs under ASan / TSan in tools/sanitize)."""
    import subprocess
    env = dict(os.environ); env["LD_PRELOAD"] = stub
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize", "run_host_paths.py"), "edge"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    ok = [line for line in r.stdout.splitlines() if line.startswith("ok ")]
    assert len(ok) == 10 and any("bal 17 cameras nt 2" in line for line in ok)


def test_smart_factors_upload_hidden_landmarks_behind_the_callers_variables(stub):
    """Host side of the smart factors under the dry-run runtime: every factor gets a hidden landmark behind the caller's
    variables (sizes reported to the caller are those of the cameras only), its measurements become observations of the
    reduced system's cameras, and malformed tables are rejected loudly."""
    d = HP.run_snippet('''
import sys, json
import numpy as np
from gtsam_amd import lib as L
from tests import problems as PB
p, v0 = PB.SMART["smart_orbit_degenerate"]()
g = L.DeviceGraph(p)
out = {"val_size": int(g.val_size), "dim_size": int(g.dim_size), "reduced_dim": int(g.reduced_dim), "n_vars": int(p.n_vars), "n_smart": int(p.n_smart)}
g.set_values(v0)
bad = 0
try:
    g.set_values(np.concatenate([v0, np.zeros(3)]))          # the hidden landmarks are not the caller's to set
except L.GtsamAmdError:
    bad += 1
g.close()
q, _ = PB.SMART["smart_orbit"]()
q.smart_cam = q.smart_cam.copy(); q.smart_cam[0] = 10**6     # not a variable
try:
    L.DeviceGraph(q)
except L.GtsamAmdError:
    bad += 1
q, _ = PB.SMART["smart_orbit"]()
q.smart_params = q.smart_params.copy(); q.smart_params[4] = 7.0   # unknown degeneracy mode
try:
    L.DeviceGraph(q)
except L.GtsamAmdError:
    bad += 1
q, _ = PB.SMART["smart_orbit"]()
q.smart_params = q.smart_params.copy(); q.smart_params[5] = 1.0   # IMPLICIT_SCHUR: the reference's direct solvers cannot eliminate it either
try:
    L.DeviceGraph(q)
except L.GtsamAmdError:
    bad += 1
out["rejected"] = bad
print("RESULT " + json.dumps(out))
''')
    assert d["val_size"] == 17 * d["n_vars"] and d["dim_size"] == 9 * d["n_vars"] == d["reduced_dim"]
    assert d["n_smart"] > 100 and d["rejected"] == 4


def test_big_device_blocks_are_kept_for_the_next_handle(stub):
    """api.hip keeps released device blocks of >= 16 MB for the next handle of the process (GTG_ALLOC_CACHE_MB per device, default
    2048, 0 = off): with it the second construction of the same problem asks the runtime for less memory, a block is only re-issued
    for a request of 80 - 100 % of its size, and gtg_release_cached_memory() gives everything back; switched off, every
    construction allocates the same."""
    code = '''
import ctypes, json
from tools import host_profile as HP
from gtsam_amd import lib as L
stub = ctypes.CDLL(HP.STUB); stub.hipstub_allocated.restype = ctypes.c_longlong
problem, _ = HP.problem_for("bal:300:20000:3")
small, _ = HP.problem_for("bal:60:6000:7")
marks = [stub.hipstub_allocated()]
g = L.DeviceGraph(problem); np_ = int(g.reduced_dim); g.close(); marks.append(stub.hipstub_allocated())
g = L.DeviceGraph(problem); g.close(); marks.append(stub.hipstub_allocated())
g = L.DeviceGraph(small); g.close(); marks.append(stub.hipstub_allocated())
g = L.DeviceGraph(problem); g.close(); marks.append(stub.hipstub_allocated())
L.load().gtg_release_cached_memory.restype = ctypes.c_longlong
released = int(L.load().gtg_release_cached_memory())
print("RESULT " + json.dumps({"reduced_dim": np_, "marks": marks, "released": released}))
'''
    on = HP.run_snippet(code, env_extra={"GTG_ALLOC_CACHE_MB": "4096"})
    off = HP.run_snippet(code, env_extra={"GTG_ALLOC_CACHE_MB": "0"})
    dflt = HP.run_snippet(code)
    first, second, small, again = np.diff(on["marks"])
    f0, s0, m0, a0 = np.diff(off["marks"])
    assert f0 == s0 == a0 == first and off["released"] == 0    # switched off: nothing is kept, every construction allocates the same
    assert np.diff(dflt["marks"])[1] <= first - (16 << 20) and dflt["released"] >= 16 << 20      # the default keeps them too
    assert second <= first - (16 << 20) and again <= first - (16 << 20)   # with the cache the big blocks are re-issued ...
    assert small == m0                                          # ... but not to a much smaller problem
    assert on["released"] >= 16 << 20                           # and the release call hands them back
