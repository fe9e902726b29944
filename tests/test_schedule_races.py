"""Race check of the multi-stream schedule of the reduced-system Cholesky on the CPU (tools/race_check.py).

The library runs one whole gtg_try_lambda under tools/hipstub; the stub records every launch, event record and stream wait
in issue order.  The happens-before relation HIP guarantees (stream FIFO + event edges) must order every two launches of
the factorisation that touch a common 128x128 tile with at least one write -- the look-ahead chain on the priority stream
against the bulk updates on the main stream, and for the elimination-tree schedule (GTG_ND_DEPTH) the per-part chains
against the cross-part update stream -- and the factorisation as a whole against what precedes and follows it on the
handle's stream.  A negative control removes the event edges from the same trace and must find races."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import host_profile as HP  # noqa: E402

_CHILD = r'''
import ctypes, json, sys
sys.path.insert(0, %(root)r)
from tools import host_profile as HP, race_check as RC
from gtsam_amd import lib as L
stub = RC.bind(ctypes.CDLL(HP.STUB))
problem, v0 = HP.problem_for(%(workload)r)
g = L.DeviceGraph(problem); g.set_values(v0); g.linearize()
stub.hipstub_trace_enable(1)
rc, out = g.try_lambda(1e-3, True)
stub.hipstub_trace_enable(0)
ops = RC.trace(stub)
res = RC.check(ops)
res["rc"] = rc
# negative control: the same trace without the event edges (every stream on its own)
res["without_events"] = RC.check([o for o in ops if o["type"] not in (RC.OP_RECORD, RC.OP_WAIT)])["races"]
pl = g.cholesky_plan(); res["parts"] = len(pl["part_parent"]); res["nt"] = pl["nt"]
print("RESULT " + json.dumps(res))
'''


@pytest.fixture(scope="module")
def stub():
    if not os.path.exists(os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd.so")):
        pytest.skip("libgtsam_amd.so not built")
    return HP.build_stub()


@pytest.mark.parametrize("workload,nd", [("bal:300:20000:3", 0), ("sphere2500", 0), ("ladybug1723", 0), ("sphere2500", 2), ("sphere2500", 3),
                                         ("bal:300:20000:3", 2), ("w20000", 3)])
def test_no_unordered_tile_conflicts(stub, workload, nd):
    # the launch-sequence schedule (GTG_CHOL=streams: look-ahead on two streams), also over a nested-dissection plan (GTG_ND_DEPTH: the
    # parts one after the other, their cross-part updates behind the bulk updates -- the form that replaced the multi-stream tree in round 5).  The
    # default dataflow schedule has two launches and orders its tile accesses through flags inside the kernels: its ordering
    # argument (a task only reads tiles that are final in ticket order) is what tests/test_chol_plan.py::_execute_df checks.
    env = {"GTG_CHOL": "streams", "GTG_ND_DEPTH": str(nd)}    # (since round 3 a nested-dissection plan runs on the dataflow kernels by default)
    r = HP.run_snippet(_CHILD % {"root": ROOT, "workload": workload}, env_extra=env, timeout=900)
    assert r["rc"] == 0 and r["factorisation_launches"] >= r["nt"]          # at least one panel launch per block column
    assert r["streams"] >= 2 and r["ordered_conflicts_checked"] > r["factorisation_launches"]
    if nd:
        assert r["parts"] > 1 and r["streams"] == 2                       # the parts as ONE chain: panel stream + the handle's stream
    assert r["races"] == 0, r["first_races"]
    assert r["without_events"] > 0                                         # the checker does see races when the edges are gone
