"""N > 1 path on CPU: world_size-2 `gloo` processes exercise
  * gtsam_amd.distributed.make_allreduce (the callback the library invokes with a pointer + length),
  * the sharding rule of the C ABI (gtg_upload_problem: landmark factors follow (landmark rank % n_shards),
    the other factors go round-robin) and the algebra the library relies on: the damped reduced camera
    system is the SUM over shards of per-shard partial systems (damping added by shard 0 only, from the
    all-reduced Hessian diagonal), so one all-reduce of [S | g] per lambda try is the only big exchange
    (SURVEY.md section 8(e)).
The per-shard partial systems are formed with the oracle (tests only); the GPU-side sharded path is covered
by tests/test_gpu_sharding.py on the device."""
import ctypes
import os
import socket

import numpy as np
import pytest

from gtsam_amd.problem import Problem, VAR_POINT3, bal_problem
from tests import problems as PB
from tests.conftest import load_golden


def shard_problem(p: Problem, shard: int, n: int) -> Problem:
    """The same filter gtg_upload_problem applies (gtsam_amd/csrc/api.hip)."""
    lm_rank = -np.ones(p.n_vars, np.int64)
    pts = np.where(p.var_type == VAR_POINT3)[0]
    lm_rank[pts] = np.arange(pts.size)
    q = Problem(var_type=p.var_type.copy(), noise_kind=p.noise_kind, noise_dim=p.noise_dim, noise_off=p.noise_off,
                noise_data=p.noise_data, calib=p.calib, sensor=p.sensor)
    k = (lm_rank[p.sfm_point] % n) == shard if p.n_sfm else np.zeros(0, bool)
    q.sfm_cam, q.sfm_point, q.sfm_noise = p.sfm_cam[k], p.sfm_point[k], p.sfm_noise[k]
    q.sfm_z = p.sfm_z.reshape(-1, 2)[k].reshape(-1)
    k = (np.arange(p.n_between) % n) == shard
    q.between_v1, q.between_v2, q.between_noise = p.between_v1[k], p.between_v2[k], p.between_noise[k]
    q.between_z = p.between_z.reshape(-1, 12)[k].reshape(-1)
    for i in range(p.n_prior):
        v = int(p.prior_var[i])
        mine = (lm_rank[v] % n) == shard if lm_rank[v] >= 0 else (i % n) == shard
        if mine:
            t = int(p.var_type[v]); size = {0: 12, 1: 17, 2: 3}[t]
            q.add_prior(v, p.prior_data[p.prior_off[i]:p.prior_off[i] + size], int(p.prior_noise[i]))
    return q


def partial_reduced_system(q: Problem, v, lam, hdiag_full, shard):
    """[S_r | g_r] of one shard: J^T J of its factors, damping lambda*clamp(diag) on its OWN landmarks and (shard 0
    only) on the cameras, landmarks eliminated by their 3x3 blocks."""
    from oracle import gtsam_oracle as O
    H, g, _ = O.hessian_dense(q, v)
    doff = q.dim_offsets()
    cams = [i for i in range(q.n_vars) if q.var_type[i] != VAR_POINT3]
    ic = np.concatenate([np.arange(doff[i], doff[i + 1]) for i in cams])
    S = H[np.ix_(ic, ic)].copy(); gr = g[ic].copy()
    if shard == 0:
        S += np.diag(lam * np.clip(hdiag_full[ic], 1e-6, 1e32))
    for i in range(q.n_vars):
        if q.var_type[i] != VAR_POINT3:
            continue
        ip = np.arange(doff[i], doff[i + 1])
        V = H[np.ix_(ip, ip)]
        if not np.any(V):
            continue                                     # landmark owned by another shard
        V = V + np.diag(lam * np.clip(np.diag(V), 1e-6, 1e32))
        W = H[np.ix_(ic, ip)]
        S -= W @ np.linalg.solve(V, W.T); gr -= W @ np.linalg.solve(V, g[ip])
    return S, gr


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from gtsam_amd.distributed import make_allreduce
    from oracle import gtsam_oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        allreduce = make_allreduce()
        g = load_golden("dubrovnik_3_7")
        p, v0 = PB.dubrovnik_sfmexample(g)
        q = shard_problem(p, rank, world)
        lam = 1e-3
        # exchange 1: Hessian diagonal (what gtg_linearize all-reduces)
        hd = np.ascontiguousarray(O.hessian_diagonal(q, v0))
        allreduce(hd.ctypes.data, hd.size, 0)
        assert np.allclose(hd, O.hessian_diagonal(p, v0), rtol=1e-12)
        # exchange 2: the reduced system + rhs (what gtg_try_lambda all-reduces)
        S, gr = partial_reduced_system(q, v0, lam, hd, rank)
        buf = np.ascontiguousarray(np.concatenate([S.reshape(-1), gr]))
        allreduce(buf.ctypes.data, buf.size, 0)
        n = gr.size
        S = buf[:n * n].reshape(n, n); gr = buf[n * n:]
        x = np.linalg.solve(S, gr)
        st, delta, *_ = O.solve_damped(p, v0, lam, True)
        doff = p.dim_offsets()
        ic = np.concatenate([np.arange(doff[i], doff[i + 1]) for i in range(p.n_vars) if p.var_type[i] != VAR_POINT3])
        err = float(np.abs(x - delta[ic]).max() / np.abs(delta[ic]).max())
        # scalars exchange: partial nonlinear errors add up
        e = np.array([O.error(q, v0)]); allreduce(e.ctypes.data, 1, 0)
        ok = st == 0 and err < 1e-9 and abs(e[0] - O.error(p, v0)) < 1e-9 * e[0]
        out[rank] = 1 if ok else 0
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_sharded_reduced_system():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0, 0])
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0
    assert list(out) == [1, 1]


def test_shard_filter_partitions_every_factor_exactly_once():
    g = load_golden("sphere2500")
    p, _ = PB.sphere2500(g)
    for n in (2, 3, 8):
        parts = [shard_problem(p, r, n) for r in range(n)]
        assert sum(q.n_between for q in parts) == p.n_between and sum(q.n_prior for q in parts) == p.n_prior
    gd = load_golden("dubrovnik_3_7")
    p, _ = PB.dubrovnik_sfmexample(gd)
    parts = [shard_problem(p, r, 2) for r in range(2)]
    assert sum(q.n_sfm for q in parts) == p.n_sfm and sum(q.n_prior for q in parts) == p.n_prior
    # a landmark's observations never straddle shards
    assert set(np.unique(parts[0].sfm_point)).isdisjoint(set(np.unique(parts[1].sfm_point)))


def shard_smart_problem(p: Problem, shard: int, n: int) -> Problem:
    """gtg_upload_problem's rule for smart factors: factor i follows its hidden landmark, the (user landmarks + i)-th POINT3
    variable, to shard (rank % n); the other factors as in shard_problem."""
    q = shard_problem(p, shard, n)
    n_user_lm = int((p.var_type == VAR_POINT3).sum())
    for i in range(p.n_smart):
        if (n_user_lm + i) % n != shard:
            continue
        k0, k1 = int(p.smart_ptr[i]), int(p.smart_ptr[i + 1])
        prm = p.smart_params.reshape(-1, 8)[i]
        q.add_smart(p.smart_cam[k0:k1], p.smart_z.reshape(-1, 2)[k0:k1], int(p.smart_noise[i]), prm[0], prm[1], prm[2], prm[3], int(prm[4]),
                    int(prm[5]), bool(prm[6]))
    return q


@pytest.mark.parametrize("name", ["smart_orbit_degenerate", "smart_far_infinity", "smart_far_jacobian_svd"])
def test_sharded_smart_graph_sums_to_the_whole(name):
    """The algebra the sharded smart path relies on (tests/test_gpu_sharding.py runs it on the device): every smart factor lives on
    exactly one shard, and error, Hessian diagonal (of the Schur-complemented factors), the reduced system of the cameras and both
    linear errors of a step are the SUMS over the shards of what each shard's factors give -- the exchanges that exist for
    explicit landmarks carry them.  Oracle only."""
    from oracle import gtsam_oracle as O
    p, v0 = PB.SMART[name]()
    for n in (2, 3):
        parts = [shard_smart_problem(p, r, n) for r in range(n)]
        assert sum(q.n_smart for q in parts) == p.n_smart and sum(q.n_prior for q in parts) == p.n_prior
        assert abs(sum(O.error(q, v0) for q in parts) - O.error(p, v0)) <= 1e-12 * O.error(p, v0)
        hd = sum(O.hessian_diagonal(q, v0) for q in parts)
        assert np.abs(hd - O.hessian_diagonal(p, v0)).max() <= 1e-12 * np.abs(hd).max()
        H = sum(O.hessian_dense(q, v0)[0] for q in parts); g = sum(O.hessian_dense(q, v0)[1] for q in parts)
        Hw, gw, _ = O.hessian_dense(p, v0)
        assert np.abs(H - Hw).max() <= 1e-12 * np.abs(Hw).max() and np.abs(g - gw).max() <= 1e-12 * np.abs(gw).max()
        st, d, _, _, lin = O.solve_damped(p, v0, 1e-3, True)
        lins = [O.solve_damped(q, v0, 1e-3, True)[4] for q in parts]          # (each shard's linearisation; the step is the whole graph's)
        for x in (np.zeros_like(d), d):
            whole = O.linear_error(p, lin, x)
            assert abs(sum(O.linear_error(q, lq, x) for q, lq in zip(parts, lins)) - whole) <= 1e-11 * abs(whole)
