"""The two extractors against each other, on the CPU: the C++ shim (gtsam_amd/host, built against the real GTSAM of
oracle/_ref; graphs from the reference's shipped files through the reference's own loaders, tests/cpp/test_shim_extractor.cpp)
and the Python mirror (gtsam_amd/api.py / problem.py; the same files through gtsam_amd/io.py or the golden copies of what the
reference's loaders returned).  Both run under tools/hipstub, which hashes every host-to-device copy: equal records mean the
library was handed identical tables -- variable order, factor tables, shared noise rows (one row per distinct model, however
many objects), packed values -- and built the identical symbolic analysis from them."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import host_profile as HP  # noqa: E402

DATA = "/root/reference/examples/Data/"
EXE = os.path.join(ROOT, "tests", "_build", "test_shim_extractor")

_CHILD = r'''
import ctypes, json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from tools import host_profile as HP
from gtsam_amd import io, lib as L
from gtsam_amd.problem import NOISE_DIAGONAL, pose_graph_problem
from tests import problems as PB
from tests.conftest import load_golden
stub = ctypes.CDLL(HP.STUB)
stub.hipstub_h2d_record.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_ulonglong)]
def records(p, v0):
    stub.hipstub_reset()
    g = L.DeviceGraph(p); g.set_values(np.ascontiguousarray(v0, np.float64))
    n = ctypes.c_longlong(); h = ctypes.c_ulonglong(); out = []
    for i in range(stub.hipstub_h2d_count()):
        stub.hipstub_h2d_record(i, ctypes.byref(n), ctypes.byref(h)); out.append("%%d:%%d" %% (n.value, h.value))
    g.close(); return out
res = {}
res["sfmexample_bal_dubrovnik_3_7"] = records(*PB.dubrovnik_sfmexample(load_golden("dubrovnik_3_7")))
res["pose2slam_w100"] = records(*PB.pose2_graph(load_golden("pose2_w100")))
# the reference's own loader through the oracle harness: bit-identical numbers to what the C++ side read (gtsam_amd/io.py
# agrees with it to 1e-14, tests/test_io.py, which is not enough for equal hashes)
from oracle import ref
d = ref.load_g2o3d(%(data)r + "pose3example.txt")
p = pose_graph_problem(len(d["vertex_keys"]), d["v1"], d["v2"], d["z"], d["noise_kind"], d["noise"])
p.add_prior(0, d["vertex_poses"][0], p.add_noise(NOISE_DIAGONAL, 6, np.sqrt([1e-6] * 3 + [1e-4] * 3)))
res["pose3slam_pose3example"] = records(p, d["vertex_poses"].reshape(-1))
print("RESULT " + json.dumps(res))
'''


def test_cpp_and_python_extractors_hand_the_library_identical_tables():
    if not (os.path.exists(EXE) and os.path.isdir(DATA) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libgtsam_ref.so"))):
        pytest.skip("shim extractor test / reference data / oracle/_ref not present on this machine")
    stub = HP.build_stub()
    env = dict(os.environ); env["LD_PRELOAD"] = stub
    r = subprocess.run([EXE, DATA], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    cpp = {}
    for line in r.stdout.splitlines():
        if line.startswith("CASE "):
            parts = line.split()
            cpp[parts[1]] = parts[2:]
    # the extraction runs on host threads (a contiguous range of the graph each, tables concatenated in graph order, rows of the noise /
    # calibration tables handed out afterwards in first-occurrence order): the same tables for any thread count
    # (and the walk over the Values cut into key ranges -- three walkers over the two symbol ranges of the SfM example, over the plain
    # integer keys of the pose graphs: the same variable order)
    env_mt = dict(env); env_mt["GTG_HOST_THREADS"] = "5"; env_mt["GTG_EXTRACT_GRAIN"] = "3"; env_mt["GTG_VALUES_WALKERS"] = "3"
    r_mt = subprocess.run([EXE, DATA], env=env_mt, capture_output=True, text=True, timeout=300)
    assert r_mt.returncode == 0 and "ALL PASSED" in r_mt.stdout, r_mt.stdout[-2000:] + r_mt.stderr[-2000:]
    # (the records as a multiset per case: the library uploads the measurements and noise rows on a helper thread beside its analysis, so
    # the ORDER of the copies is not fixed)
    cases = lambda out: [(ln.split()[1], sorted(ln.split()[2:])) for ln in out.splitlines() if ln.startswith("CASE ")]
    assert cases(r_mt.stdout) == cases(r.stdout)
    py = HP.run_snippet(_CHILD % {"root": ROOT, "data": DATA})
    assert set(cpp) == set(py) == {"sfmexample_bal_dubrovnik_3_7", "pose2slam_w100", "pose3slam_pose3example"}
    for name in cpp:
        assert len(cpp[name]) > 30
        assert sorted(cpp[name]) == sorted(py[name]), name
