"""Executable specification of the device-side symbolic analysis planned for the next round.

The host builds the Schur term lists with bucketed counting sorts on threads (csrc/analysis.hip).  On the device the same
lists are to come from three data-parallel primitives: enumerate every landmark's observation pairs at the offsets of an
exclusive scan, STABLE radix sort of the terms by their block key (row position * n + column position), run-length encode
the sorted keys.  This test states that formulation in numpy and checks that it reproduces, bit for bit, the term lists the
library uploads (compared through the upload hashes of tools/hipstub, with GTG_ORDERING=natural so that the positions are the
caller's order): same terms, same order inside every block (landmark order -- the summation order of the device), same
handling of a camera that observes a landmark twice."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import host_profile as HP  # noqa: E402

_CHILD = r'''
import ctypes, json, sys
sys.path.insert(0, %(root)r)
from tools import host_profile as HP
from gtsam_amd import lib as L
stub = ctypes.CDLL(HP.STUB)
stub.hipstub_h2d_record.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_ulonglong)]
problem, _ = HP.problem_for(%(workload)r)
stub.hipstub_reset()
g = L.DeviceGraph(problem)
n = ctypes.c_longlong(); h = ctypes.c_ulonglong(); recs = []
for i in range(stub.hipstub_h2d_count()):
    stub.hipstub_h2d_record(i, ctypes.byref(n), ctypes.byref(h)); recs.append([n.value, str(h.value)])
print("RESULT " + json.dumps(recs))
'''


def _fnv(a):
    """The stub's record hash: FNV-1a over 8-byte little-endian words, then the remaining bytes."""
    b = np.ascontiguousarray(a).tobytes()
    h = 1469598103934665603; M = (1 << 64) - 1
    nw = len(b) // 8
    for w in np.frombuffer(b[:8 * nw], dtype="<u8").tolist():
        h = ((h ^ w) * 1099511628211) & M
    for c in b[8 * nw:]:
        h = ((h ^ c) * 1099511628211) & M
    return h


def _sort_based_term_lists(problem):
    """enumerate -> stable sort by block key -> run-length encode (positions = the caller's order of the cameras)."""
    cam_of = problem.sfm_cam.astype(np.int64); lm_of = problem.sfm_point.astype(np.int64)
    red_vars = np.where(problem.var_type != 2)[0]; lm_vars = np.where(problem.var_type == 2)[0]
    red_index = -np.ones(problem.n_vars, np.int64); red_index[red_vars] = np.arange(red_vars.size)
    lm_index = -np.ones(problem.n_vars, np.int64); lm_index[lm_vars] = np.arange(lm_vars.size)
    pos = red_index[cam_of]; lm = lm_index[lm_of]
    nrv = red_vars.size
    order = np.argsort(lm, kind="stable")                       # landmark -> its observations in factor order (CSR)
    cnt = np.bincount(lm, minlength=lm_vars.size); ptr = np.concatenate([[0], np.cumsum(cnt)])
    kmax = int(cnt.max()) if cnt.size else 0
    stride = kmax * (kmax + 1) + 1                                # emission slots per landmark: 2 per (a, b) pair
    oa_l = []
    for k in np.unique(cnt):                                      # all landmarks with k observations at once
        if k == 0:
            continue
        ls = np.where(cnt == k)[0]
        obs = order[ptr[ls][:, None] + np.arange(k)[None, :]]     # [n_l, k] observation ids
        a, b = np.tril_indices(k)                                 # a >= b, row-major: the emission order (a, then b <= a)
        xa, xb = obs[:, a], obs[:, b]; pa, pb = pos[xa], pos[xb]
        swap = pa < pb
        oa = np.where(swap, xb, xa); ob = np.where(swap, xa, xb); hi = np.maximum(pa, pb); lo = np.minimum(pa, pb)
        dup = (pa == pb) & (xa != xb)                             # same camera twice: the mirrored term right behind
        # interleave: term, then (where dup) its mirror; emission index keeps (landmark, a, b, mirror) order
        n_l, n_t = oa.shape
        oa2 = np.stack([oa, ob], 2).reshape(n_l, 2 * n_t); ob2 = np.stack([ob, oa], 2).reshape(n_l, 2 * n_t)
        key2 = np.repeat(hi * nrv + lo, 2, axis=1)
        keep = np.stack([np.ones_like(dup), dup], 2).reshape(n_l, 2 * n_t)
        seq = ls[:, None] * stride + np.arange(2 * n_t)[None, :]  # global emission order: landmark-major
        oa_l.append(np.stack([seq[keep], oa2[keep], ob2[keep], key2[keep]], 1))
    allt = np.concatenate(oa_l, 0)
    allt = allt[np.argsort(allt[:, 0], kind="stable")]            # emission order (what the scan offsets give on the device)
    srt = np.argsort(allt[:, 3], kind="stable")                   # the stable radix sort by block key
    keys = allt[srt, 3]
    starts = np.flatnonzero(np.concatenate([[True], keys[1:] != keys[:-1]]))   # run-length encode
    return allt[srt, 1].astype(np.int32), allt[srt, 2].astype(np.int32), np.concatenate([starts, [keys.size]]).astype(np.int64)


@pytest.mark.parametrize("workload", ["bal:60:6000:7", "baldup:40:3000:3", "bal:300:20000:3"])
def test_sort_based_formulation_reproduces_the_host_lists(workload):
    if not os.path.exists(os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd.so")):
        pytest.skip("libgtsam_amd.so not built")
    recs = HP.run_snippet(_CHILD % {"root": ROOT, "workload": workload}, env_extra={"GTG_ORDERING": "natural"})
    have = {(int(n), int(h)) for n, h in recs}
    problem, _ = HP.problem_for(workload)
    oa, ob, ptr = _sort_based_term_lists(problem)
    assert (oa.nbytes, _fnv(oa)) in have, "pair_oa differs"
    assert (ob.nbytes, _fnv(ob)) in have, "pair_ob differs"
    assert (ptr.nbytes, _fnv(ptr)) in have, "pair_ptr differs"


def _sort_based_incidence_lists(problem):
    """device_analysis.hip::device_incidence_lists in numpy: per observation (reduced index, landmark), the landmark -> observation
    lists by a STABLE sort of the observations by landmark + offsets by binary search in the sorted keys, and the reduced
    variable -> factor lists by a stable sort of one entry per (factor, key) -- numbered GeneralSFM observations, projection
    observations, between factors (factor i: first key, then second key), priors -- by reduced index (keys that are not reduced
    variables sort behind everything and are dropped)."""
    VAR_POINT3 = 2
    red_vars = np.where(problem.var_type != VAR_POINT3)[0]; lm_vars = np.where(problem.var_type == VAR_POINT3)[0]
    red_index = -np.ones(problem.n_vars, np.int64); red_index[red_vars] = np.arange(red_vars.size)
    lm_index = -np.ones(problem.n_vars, np.int64); lm_index[lm_vars] = np.arange(lm_vars.size)
    cam = np.concatenate([problem.sfm_cam, problem.proj_pose]).astype(np.int64)
    pt = np.concatenate([problem.sfm_point, problem.proj_point]).astype(np.int64)
    obs_red = red_index[cam].astype(np.int32); obs_lm = lm_index[pt].astype(np.int32)
    order = np.argsort(obs_lm, kind="stable")
    lm_obs = order.astype(np.int32)
    lm_obs_ptr = np.searchsorted(obs_lm[order], np.arange(lm_vars.size + 1), side="left").astype(np.int64)
    n_sfm, n_proj, n_btw, n_pri = problem.n_sfm, problem.n_proj, problem.n_between, problem.n_prior
    bt = np.stack([problem.between_v1, problem.between_v2], 1).reshape(-1).astype(np.int64)       # factor i: v1, v2
    v = np.concatenate([problem.sfm_cam.astype(np.int64), problem.proj_pose.astype(np.int64), bt, problem.prior_var.astype(np.int64)])
    key = red_index[v]; key = np.where(key >= 0, key, red_vars.size)
    srt = np.argsort(key, kind="stable")
    inc_ptr = np.searchsorted(key[srt], np.arange(red_vars.size + 1), side="left").astype(np.int64)
    seq = srt[:inc_ptr[-1]]
    b0, b1 = n_sfm + n_proj, n_sfm + n_proj + 2 * n_btw
    kind = np.where(seq < n_sfm, 0, np.where(seq < b0, 1, np.where(seq < b1, 2 + ((seq - b0) & 1), 4))).astype(np.int32)
    idx = np.where(seq < n_sfm, seq, np.where(seq < b0, seq - n_sfm, np.where(seq < b1, (seq - b0) >> 1, seq - b1))).astype(np.int32)
    return obs_red, obs_lm, lm_obs_ptr, lm_obs, inc_ptr, kind, idx


@pytest.mark.parametrize("workload", ["bal:60:6000:7", "baldup:40:3000:3", "dubrovnik_3_7", "smart:smart_far_infinity"])
def test_sort_based_incidence_lists_reproduce_the_host_lists(workload):
    """The same for the incidence lists the device pass builds in front of the term lists (landmark -> observations, reduced
    variable -> factors): the sort-based formulation against the lists the host's counting sorts upload."""
    if not os.path.exists(os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd.so")):
        pytest.skip("libgtsam_amd.so not built")
    recs = HP.run_snippet(_CHILD % {"root": ROOT, "workload": workload}, env_extra={"GTG_ORDERING": "natural"})
    have = {(int(n), int(h)) for n, h in recs}
    problem, _ = HP.problem_for(workload)
    if workload.startswith("smart:"):          # the library's own view of a smart graph: hidden landmarks, measurements as observations
        from gtsam_amd.problem import Problem
        q = Problem(var_type=np.concatenate([problem.var_type, np.full(problem.n_smart, 2, np.int32)]))
        q.sfm_cam = problem.smart_cam.copy()
        q.sfm_point = (problem.n_vars + np.repeat(np.arange(problem.n_smart), np.diff(problem.smart_ptr))).astype(np.int32)
        q.prior_var = problem.prior_var.copy()
        problem = q
    names = ("obs_red", "obs_lm", "lm_obs_ptr", "lm_obs", "red_inc_ptr", "red_inc_kind", "red_inc_idx")
    for name, a in zip(names, _sort_based_incidence_lists(problem)):
        assert (a.nbytes, _fnv(a)) in have, name + " differs"


# ---- reverse Cuthill-McKee, level-synchronous: the specification for moving the ordering of the camera graph onto the device ---------
def _camera_graph(problem):
    """Adjacency (CSR, neighbours ascending) of the reduced variables: two cameras share an edge when they see a common landmark,
    two poses when a between factor joins them -- the off-diagonal blocks of the reduced system."""
    VAR_POINT3 = 2
    red_vars = np.where(problem.var_type != VAR_POINT3)[0]
    red_index = -np.ones(problem.n_vars, np.int64); red_index[red_vars] = np.arange(red_vars.size)
    cam = red_index[np.concatenate([problem.sfm_cam, problem.proj_pose]).astype(np.int64)]
    pt = np.concatenate([problem.sfm_point, problem.proj_point]).astype(np.int64)
    order = np.argsort(pt, kind="stable"); cam = cam[order]; pt = pt[order]
    starts = np.flatnonzero(np.r_[True, pt[1:] != pt[:-1], True])
    rows, cols = [], []
    for k in np.unique(np.diff(starts)):
        if k < 2:
            continue
        s0 = starts[:-1][np.diff(starts) == k]
        obs = cam[s0[:, None] + np.arange(k)[None, :]]
        a, b = np.tril_indices(k, -1)
        rows.append(obs[:, a].reshape(-1)); cols.append(obs[:, b].reshape(-1))
    rows.append(red_index[problem.between_v1.astype(np.int64)]); cols.append(red_index[problem.between_v2.astype(np.int64)])
    r = np.concatenate(rows); c = np.concatenate(cols)
    keep = r != c
    r, c = np.concatenate([r[keep], c[keep]]), np.concatenate([c[keep], r[keep]])
    n = red_vars.size
    key = np.unique(r * n + c)
    r, c = key // n, key % n
    ptr = np.searchsorted(r, np.arange(n + 1), side="left")
    return red_vars, ptr, c


def _level_sync_rcm(ptr, adj):
    """Reverse Cuthill-McKee as DATA-PARALLEL steps per BFS level -- what a device version would run -- reproducing the host's
    serial queue (csrc/analysis.hip) node for node:
      frontier expansion: every unvisited neighbour of the level is claimed by the EARLIEST node of the level it is adjacent to
        (a segmented minimum over the level's adjacency), and the next level is those nodes sorted by (position of the claiming
        node, degree, id) -- the order in which the serial algorithm appends them;
      start node: two sweeps of "the last BFS level's node of smallest degree" from the component's first node (in a plain BFS the
        next level is sorted by (claiming position, id); the candidate is the LAST node of the last level unless a node of that
        level has a strictly smaller degree, then the first such node of minimum degree);
      components in ascending order of their first node; the concatenated order reversed."""
    n = ptr.size - 1
    deg = np.diff(ptr)
    active = np.ones(n, bool)

    def expand(level, visited, by_degree):
        seg = np.repeat(np.arange(level.size), deg[level])
        nb = np.concatenate([adj[ptr[v]:ptr[v + 1]] for v in level]) if level.size else np.zeros(0, np.int64)
        ok = active[nb] & ~visited[nb]
        nb, seg = nb[ok], seg[ok]
        if nb.size == 0:
            return nb
        o = np.lexsort((seg, nb))                       # per neighbour: its earliest claiming position
        nb, seg = nb[o], seg[o]
        first = np.r_[True, nb[1:] != nb[:-1]]
        nb, seg = nb[first], seg[first]
        o = np.lexsort((nb, deg[nb], seg)) if by_degree else np.lexsort((nb, seg))
        return nb[o]

    def far_node(start):
        visited = np.zeros(n, bool); visited[start] = True
        level = np.array([start])
        while True:
            nxt = expand(level, visited, by_degree=False)
            if nxt.size == 0:
                break
            visited[nxt] = True; level = nxt
        m = deg[level].min()
        return int(level[-1]) if deg[level[-1]] == m else int(level[np.flatnonzero(deg[level] == m)[0]])

    order = []
    for seed in range(n):
        if not active[seed]:
            continue
        start = far_node(far_node(seed))
        visited = np.zeros(n, bool); visited[start] = True
        level = np.array([start]); comp = [level]
        while True:
            nxt = expand(level, visited, by_degree=True)
            if nxt.size == 0:
                break
            visited[nxt] = True; comp.append(nxt); level = nxt
        comp = np.concatenate(comp)
        active[comp] = False
        order.append(comp)
    return np.concatenate(order)[::-1]


@pytest.mark.parametrize("workload", ["bal:60:6000:7", "bal:300:20000:3", "baldup:40:3000:3", "streets1723"])
def test_level_synchronous_rcm_reproduces_the_host_ordering(workload):
    """The ordering of the reduced variables is the last symbolic step still computed by host code (on the camera graph: 1 723
    nodes, 0.2 M edges for the bench shapes).  This states reverse Cuthill-McKee as per-level data-parallel steps (segmented
    minimum + sort) and checks that it reproduces the library's serial implementation position for position
    (gtg_debug_reduced_order under the dry-run runtime; GTG_ORDERING=rcm: one chain, no nested dissection)."""
    if not os.path.exists(os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd.so")):
        pytest.skip("libgtsam_amd.so not built")
    have = HP.run_snippet('''
import json
from tools import host_profile as HP
from gtsam_amd import lib as L
p, _ = HP.problem_for(%r)
g = L.DeviceGraph(p)
print("RESULT " + json.dumps(g.reduced_order().tolist()))
''' % workload, env_extra={"GTG_ORDERING": "rcm"})
    problem, _ = HP.problem_for(workload)
    red_vars, ptr, adj = _camera_graph(problem)
    want = red_vars[_level_sync_rcm(ptr, adj)]
    assert np.array_equal(np.asarray(have), want)
