"""GPU: the Schur term lists built on the device (csrc/device_analysis.hip: count, scan, emit, stable radix sort, run-length
encode) against the host threads' lists (GTG_HOST_ANALYSIS=1).  The lists fix the summation order of the Schur complement, so
equal lists mean bit-identical numbers: the step (which depends on every entry of the reduced system and of its right-hand
side), the four scalars of the lambda try and the layout hash of the two handles must be EQUAL, not close -- on graphs that are reordered (blocks change orientation after the ordering), on one where a camera sees a landmark
twice (mirrored diagonal terms), on projection factors and on a graph without landmarks."""
import numpy as np
import pytest

from tests import problems as PB
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _problems():
    from tools import host_profile as HP
    yield "bal_60_cameras", HP.problem_for("bal:60:6000:7")
    yield "camera_sees_landmark_twice", HP.problem_for("baldup:40:3000:3")
    yield "projection_small", PB.SYNTH["projection_small"]()
    yield "dubrovnik_3_7", PB.dubrovnik_timesfm(load_golden("dubrovnik_3_7"))
    yield "posegraph_small (no landmarks)", PB.SYNTH["posegraph_small"]()


@pytest.mark.parametrize("name,pv", list(_problems()), ids=[n for n, _ in _problems()])
def test_device_lists_equal_host_lists(name, pv, monkeypatch):
    import torch
    assert torch.cuda.is_available()
    from gtsam_amd import lib as L
    p, v0 = pv

    def run():
        dev = L.DeviceGraph(p)
        dev.set_values(v0)
        dev.linearize()
        rc, out = dev.try_lambda(1e-3, True)
        res = (rc, out.copy(), dev.delta().copy(), dev.structure_hash(), dev.cholesky_flops())
        dev.close()
        return res

    monkeypatch.delenv("GTG_HOST_ANALYSIS", raising=False)
    a = run()
    monkeypatch.setenv("GTG_HOST_ANALYSIS", "1")
    b = run()
    assert a[0] == b[0] == 0
    assert a[3] == b[3] and a[4] == b[4]
    assert np.array_equal(a[2], b[2]), np.abs(a[2] - b[2]).max()       # the step, bit for bit
    assert np.array_equal(a[1], b[1])                                   # linear errors, trial error, |delta|


def _ordering_problems():
    from tools import host_profile as HP
    from gtsam_amd import datasets as D
    from gtsam_amd.problem import bal_problem
    yield "bal_60_cameras", HP.problem_for("bal:60:6000:7")[0]
    yield "bal_300_cameras", HP.problem_for("bal:300:20000:3")[0]
    yield "camera_sees_landmark_twice", HP.problem_for("baldup:40:3000:3")[0]
    yield "streets1723 (long-range loop closures)", HP.problem_for("streets1723")[0]
    yield "ladybug1723 (the bench shape)", bal_problem(*D.ladybug_1723())[0]
    yield "sphere2500 as one part", PB.sphere2500(load_golden("sphere2500"))[0]
    yield "dubrovnik16 (16 cameras: complete graph)", bal_problem(*D.dubrovnik_16())[0]


@pytest.mark.parametrize("name,p", list(_ordering_problems()), ids=[n for n, _ in _ordering_problems()])
def test_device_ordering_equals_host_ordering(name, p, monkeypatch):
    """Reverse Cuthill-McKee of the reduced variables on the device (csrc/device_ordering.hip: adjacency by radix sort, one workgroup
    walking the BFS levels) against the host's serial queue (GTG_HOST_ORDERING=1): the same elimination order position for position,
    hence the same layout hash and the same tile schedule (inference/Ordering.cpp:42-124 is the reference's place for this step)."""
    import torch
    assert torch.cuda.is_available()
    from gtsam_amd import lib as L
    monkeypatch.setenv("GTG_ND_DEPTH", "0")          # one part: the case the device kernel covers (nested dissection stays on the host)

    def run():
        dev = L.DeviceGraph(p)
        res = (dev.reduced_order().copy(), dev.structure_hash(), dev.cholesky_flops())
        dev.close()
        return res

    monkeypatch.delenv("GTG_HOST_ORDERING", raising=False)
    a = run()
    monkeypatch.setenv("GTG_HOST_ORDERING", "1")
    b = run()
    assert np.array_equal(a[0], b[0]), (np.flatnonzero(a[0] != b[0])[:10], a[0][:10], b[0][:10])
    assert a[1] == b[1] and a[2] == b[2]


def _symbolic_problems():
    from tools import host_profile as HP
    from gtsam_amd import datasets as D
    from gtsam_amd.problem import bal_problem
    yield "bal_300_cameras", HP.problem_for("bal:300:20000:3")
    yield "streets1723 (long-range loop closures)", HP.problem_for("streets1723")
    yield "ladybug1723 (the bench shape)", bal_problem(*D.ladybug_1723())
    yield "dubrovnik16 (below the device pass's size: host either way)", bal_problem(*D.dubrovnik_16())


@pytest.mark.parametrize("name,pv", list(_symbolic_problems()), ids=[n for n, _ in _symbolic_problems()])
def test_device_tile_marks_and_plan_tables_equal_host(name, pv, monkeypatch):
    """After the ordering: the marks of the Schur blocks in the tile / strip structure (device_analysis.hip::k_da_tile_marks) and the
    dataflow plan's device tables (chol_dataflow.hip::k_df_resolve) against the host loops they replace (GTG_HOST_SYMBOLIC=1): the same
    layout hash (tile structure, exchange list, block set), the same flop counts (stored tiles; executed = after the sub-tile masks of
    the strip-level symbolic factorisation), the same task / step / chain tables word for word, and the same step bit for bit."""
    import torch
    assert torch.cuda.is_available()
    from gtsam_amd import lib as L
    p, v0 = pv

    def run():
        dev = L.DeviceGraph(p)
        dev.set_values(v0)
        dev.linearize()
        rc, out = dev.try_lambda(1e-3, True)
        res = (rc, out.copy(), dev.delta().copy(), dev.structure_hash(), dev.cholesky_flops(), dev.cholesky_flops_executed(), dev.df_device_tables())
        dev.close()
        return res

    monkeypatch.delenv("GTG_HOST_SYMBOLIC", raising=False)
    a = run()
    monkeypatch.setenv("GTG_HOST_SYMBOLIC", "1")
    b = run()
    assert a[0] == b[0] == 0
    assert a[3] == b[3] and a[4] == b[4] and a[5] == b[5]
    for x, y in zip(a[6], b[6]):
        assert x.shape == y.shape and np.array_equal(x, y)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1])
