"""GPU: the library's sharded path (n_shards = 2) on ONE device: two handles, each owning half of the landmarks,
driven by two host threads; the all-reduce callback sums the two device buffers in place.  The combined result
must equal the single-handle run: same delta, same errors, same LM trajectory."""
import threading

import numpy as np
import pytest

from gtsam_amd.params import LevenbergMarquardtParams as LMP
from tests import problems as PB
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


class TwoWaySum:
    """allreduce for two handles living in one process (test plumbing only)."""

    def __init__(self):
        import torch
        self.torch = torch
        self.slots = [None, None]
        self.barrier = threading.Barrier(2)

    def fn(self, rank):
        from gtsam_amd.distributed import _DevicePtr
        torch = self.torch

        def allreduce(ptr, n, stream):
            torch.cuda.synchronize()
            self.slots[rank] = torch.as_tensor(_DevicePtr(ptr, n), device="cuda")
            self.barrier.wait()
            if rank == 0:
                s = self.slots[0] + self.slots[1]
                self.slots[0].copy_(s); self.slots[1].copy_(s)
                torch.cuda.synchronize()
            self.barrier.wait()
        return allreduce


@pytest.mark.parametrize("case", ["dubrovnik_sfmex", "bal_small_unit", "posegraph_small", "bal_60_cameras", "posegraph_200",
                                  "smart_orbit_degenerate", "smart_far_infinity", "smart_far_jacobian_svd"])
def test_two_shards_equal_one(case):
    import torch
    assert torch.cuda.is_available()
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt
    if case == "dubrovnik_sfmex":
        p, v0 = PB.dubrovnik_sfmexample(load_golden("dubrovnik_3_7")); params = LMP()
    elif case == "bal_small_unit":
        p, v0 = PB.SYNTH[case](); params = LMP.CeresDefaults()
    elif case.startswith("smart_"):
        # smart factors follow their hidden landmark: each shard triangulates and eliminates the tracks it owns (failed tracks as
        # nothing / as points at infinity, Hessian- and Jacobian-mode constants of the linear error), the sums meet in the exchange
        p, v0 = PB.SMART[case](); params = LMP.CeresDefaults()
    elif case == "bal_60_cameras":
        # 540 reduced dimensions = 5 tiles, RCM-reordered: the shards' layouts only agree because they are derived from
        # the whole graph (each shard has the Schur blocks of half of the landmarks)
        from gtsam_amd import datasets as D
        from gtsam_amd.problem import bal_problem
        p, v0 = bal_problem(*D.synthetic_bal(60, 6000, seed=7)); params = LMP.CeresDefaults(); params.setMaxIterations(8)
    elif case == "posegraph_200":
        from gtsam_amd import datasets as D
        p, v0 = D.random_pose_graph(200, 60, seed=4); params = LMP(); params.setMaxIterations(10)   # 1200 dimensions, 10 tiles
    else:
        p, v0 = PB.SYNTH[case](); params = LMP()
    single = DeviceLevenbergMarquardt(p, v0, params)
    single.dev.linearize()
    rc1, out1 = single.dev.try_lambda(1e-3, params.diagonalDamping)
    d1 = single.dev.delta()
    single.optimize()

    sumr = TwoWaySum()
    res = [None, None]

    def run(rank):
        try:
            opt = DeviceLevenbergMarquardt(p, v0, params, shard=rank, n_shards=2, allreduce=sumr.fn(rank))
            opt.dev.linearize()
            rc, out = opt.dev.try_lambda(1e-3, params.diagonalDamping)
            d = opt.dev.delta()
            opt.optimize()
            res[rank] = (rc, out, d, np.array(opt.trace)[:, :3], opt.values_packed())
        except Exception as e:  # noqa: BLE001
            res[rank] = e
            sumr.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    for r in res:
        assert not isinstance(r, Exception), r
    # the sums over shards associate differently from the single handle's: rounding differences are amplified by the
    # conditioning of the damped system (the two larger cases: thousands of landmarks / a 1200-dimensional pose graph)
    big = case in ("bal_60_cameras", "posegraph_200")
    tol_d, tol_e, tol_v = (1e-6, 1e-6, 1e-4) if big else (1e-9, 1e-7, 1e-6)
    for rc, out, d, trace, vals in res:
        assert rc == rc1
        assert np.abs(d - d1).max() <= tol_d * np.abs(d1).max()
        assert np.allclose(out[:3], out1[:3], rtol=max(tol_d, 1e-9))
        ref = np.array(single.trace)[:, :3]
        assert trace.shape == ref.shape and np.array_equal(trace[:, 0], ref[:, 0])
        assert np.abs(trace[:, 1] - ref[:, 1]).max() <= tol_e * np.abs(ref[:, 1]).max()
        assert np.abs(vals - single.values_packed()).max() <= tol_v * np.abs(vals).max()
    # both shards hold identical values (lock-step)
    assert np.array_equal(res[0][4], res[1][4])


@pytest.mark.parametrize("name", ["ladybug1723", "venice1778"])
def test_two_shards_equal_one_at_headline_size(name):
    """BASELINE config 5 is a SHARDED config (Venice, landmarks sharded): two shards of the L1723 and of the Venice shape on
    one device against the single handle -- the first lambda try of the Ceres preset (lambda = 1e-4, diagonal damping: the
    numbers LM's accept / reject decision is made from) and the first LM iterations.  The partial Schur complements of the
    two shards are summed by the exchange, so the sums associate differently from the single handle's: the step agrees
    to 1e-7 of its max-norm (the tolerance of one damped solve against the reference, tests/test_gpu_headline_parity.py),
    the errors to 1e-9, and every lambda try is accepted / rejected alike."""
    import torch
    assert torch.cuda.is_available()
    from gtsam_amd import datasets as D
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt
    from gtsam_amd.problem import bal_problem
    gen = {"ladybug1723": D.ladybug_1723, "venice1778": D.venice_1778}[name]
    p, v0 = bal_problem(*gen())
    params = LMP.CeresDefaults(); params.setMaxIterations(3 if name == "venice1778" else 8)
    lam = 1e-4

    def one(opt):
        e0 = opt.dev.error()
        opt.dev.linearize()
        rc, out = opt.dev.try_lambda(lam, True)
        d = opt.dev.delta()
        opt.optimize()
        return rc, e0, np.array(out[:3]), d, np.array(opt.trace)[:, :3], opt.values_packed()

    single = DeviceLevenbergMarquardt(p, v0, params)
    rc1, e1, out1, d1, tr1, val1 = one(single)
    h1 = single.dev.structure_hash()
    single.dev.close()
    assert rc1 == 0
    sumr = TwoWaySum()
    res = [None, None]

    def run(rank):
        try:
            opt = DeviceLevenbergMarquardt(p, v0, params, shard=rank, n_shards=2, allreduce=sumr.fn(rank))
            res[rank] = one(opt) + (opt.dev.structure_hash(),)
            opt.dev.close()
        except Exception as e:  # noqa: BLE001
            res[rank] = e
            sumr.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(900)
    for r in res:
        assert not isinstance(r, Exception), r
    for rc, e0, out, d, tr, vals, h in res:
        assert rc == rc1 and h == h1                      # one layout of the reduced system on every shard and on the single handle
        assert abs(e0 - e1) <= 1e-9 * e1
        assert np.abs(d - d1).max() <= 1e-7 * np.abs(d1).max(), np.abs(d - d1).max() / np.abs(d1).max()
        assert np.allclose(out, out1, rtol=1e-7, atol=0)
        assert tr.shape == tr1.shape and np.array_equal(tr[:, 0], tr1[:, 0]), (tr, tr1)      # same accept / reject sequence
        assert (np.abs(tr[:, 1] - tr1[:, 1]) <= 1e-6 * np.abs(tr1[:, 1])).all(), (tr, tr1)
        assert np.abs(vals - val1).max() <= 1e-5 * np.abs(val1).max()
    assert np.array_equal(res[0][5], res[1][5])           # lock step: both shards hold identical values


@pytest.mark.parametrize("case", ["bal_60_cameras", "bal_small_unit", "posegraph_small"])
def test_two_shards_pcg_equal_one(case):
    """gtg_try_lambda_pcg on a sharded graph: b, the block-Jacobi blocks and every product S p are partial sums per shard and
    all-reduced (one exchange of the reduced vector per product), the rest of the CG runs redundantly in lock step.  Same
    iteration count and, up to the association of the sums, the same step as the single handle."""
    import torch
    assert torch.cuda.is_available()
    from gtsam_amd import lib as L
    if case == "bal_60_cameras":
        from gtsam_amd import datasets as D
        from gtsam_amd.problem import bal_problem
        p, v0 = bal_problem(*D.synthetic_bal(60, 6000, seed=7)); diag = True
    else:
        p, v0 = PB.SYNTH[case](); diag = case == "bal_small_unit"
    cg = dict(max_iterations=300, min_iterations=1, epsilon_rel=1e-10, epsilon_abs=1e-14)

    def solve(dev):
        dev.set_values(v0)
        dev.linearize()
        rc, out, its = dev.try_lambda_pcg(1e-3, diag, **cg)
        return rc, out, its, dev.delta().copy()

    single = L.DeviceGraph(p)
    rc1, out1, its1, d1 = solve(single)
    single.close()
    assert rc1 == 0 and its1 > 1
    sumr = TwoWaySum()
    res = [None, None]

    def run(rank):
        try:
            dev = L.DeviceGraph(p, shard=rank, n_shards=2, allreduce=sumr.fn(rank))
            res[rank] = solve(dev)
            dev.close()
        except Exception as e:  # noqa: BLE001
            res[rank] = e
            sumr.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    for r in res:
        assert not isinstance(r, Exception), r
    for rc, out, its, d in res:
        assert rc == 0 and abs(its - its1) <= 1
        assert np.abs(d - d1).max() <= 1e-7 * np.abs(d1).max()
        assert np.allclose(out[:3], out1[:3], rtol=1e-8)
    assert np.array_equal(res[0][3], res[1][3])   # lock step: bitwise the same step on both shards


def test_nccl_allreduce_callback_on_a_raw_device_pointer():
    """The callback bench.py registers for N > 1: a raw device pointer wrapped zero-copy and all-reduced with the
    `nccl` (= RCCL) backend.  World size 1 here (one GPU on the box): checks the wrapping + collective plumbing."""
    import socket
    import torch
    import torch.distributed as dist
    from gtsam_amd.distributed import make_allreduce
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        t = torch.arange(1000, dtype=torch.float64, device="cuda")
        fn = make_allreduce()
        fn(t.data_ptr(), t.numel(), 0)
        assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64))
    finally:
        dist.destroy_process_group()
