"""Executable specification of the GROUPED Schur complement (csrc/schur_groups.hip, GTG_SCHUR=groups).

k_schur_pairs takes one wavefront per block pair (a, b) of the reduced system and reads two E slots per term from memory: 12.1 M
slot reads for the 6.07 M terms of the L1723 shape.  The grouped form cuts the cameras into GROUPS of 8 consecutive positions of the
elimination order; a workgroup owns a PAIR of groups (ga >= gb) and walks its CELLS -- one cell per landmark seen from both groups:
the landmark's observations in ga (A entries) and in gb (B entries) -- in landmark order, staging every slot of a chunk of cells
into LDS once.  Wavefront w owns row camera (8 ga + w): for each of its A entries it multiplies with every B entry of the cell
(diagonal group pair: the B entries at positions <= its own) into one of eight accumulators, chosen by the column camera.  Every
block (a, b) thus receives its terms from ONE wavefront, in landmark order: the same additions in the same order as
k_schur_pairs, as long as no camera sees a landmark twice (then the four terms of that landmark in the diagonal block come in
another order).

This file states the list layout (what analysis.hip::build_schur_groups uploads) and the kernel's walk in numpy, and checks
  * the walk reproduces the pair-major term sequences of tests/test_device_analysis_spec.py block by block;
  * the library's uploads under the dry-run runtime are these lists, bit for bit (upload hashes of tools/hipstub);
  * the chunking rule never asks for more LDS slots than the kernel has."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import host_profile as HP  # noqa: E402
from tests.test_device_analysis_spec import _CHILD, _fnv, _sort_based_term_lists  # noqa: E402

G = 8            # cameras per group = wavefronts per workgroup (schur_groups.hip::kGroup)
NSLOT = 128      # E slots of one chunk in LDS (schur_groups.hip::kChunkSlots)
CHUNK_CELLS = 64  # cells a chunk is cut from at most (one per lane of the wavefront that builds the chunk's tables)


def incidence(problem):
    """landmark -> observations (factor order) and observation -> position of its camera, positions = the caller's order."""
    red_vars = np.where(problem.var_type != 2)[0]; lm_vars = np.where(problem.var_type == 2)[0]
    red_index = -np.ones(problem.n_vars, np.int64); red_index[red_vars] = np.arange(red_vars.size)
    lm_index = -np.ones(problem.n_vars, np.int64); lm_index[lm_vars] = np.arange(lm_vars.size)
    cam = np.concatenate([problem.sfm_cam, problem.proj_pose]).astype(np.int64)
    pt = np.concatenate([problem.sfm_point, problem.proj_point]).astype(np.int64)
    obs_pos = red_index[cam]; obs_lm = lm_index[pt]
    order = np.argsort(obs_lm, kind="stable")
    ptr = np.searchsorted(obs_lm[order], np.arange(lm_vars.size + 1), side="left")
    return ptr.astype(np.int64), order.astype(np.int32), obs_pos.astype(np.int32), int(red_vars.size)


def group_lists(lm_ptr, lm_obs, obs_pos, nrv):
    """The lists of analysis.hip::build_schur_groups.
       gs_obs       every landmark's observations, stably sorted by the position of their camera (same segments as lm_obs);
                    uploaded as observation | (position mod 8) << 28 -- the camera's place inside its group rides along
       cells        per landmark, groups ascending g_0 < g_1 < ...: for ia, for ib <= ia: (key = g_ia NG + g_ib, a0, b0, p | q << 16),
                    a0 / b0 = index of the group's run in gs_obs, p / q = its length; then STABLE sort by key
       pair_key / pair_ptr   run-length encoding of the sorted keys
       order        the group pairs by descending number of cells (stable): the order in which workgroups take them"""
    NG = (nrv + G - 1) // G
    gs_obs = np.empty_like(lm_obs)
    key = []; a0 = []; b0 = []; pq = []
    for l in range(lm_ptr.size - 1):
        s, e = int(lm_ptr[l]), int(lm_ptr[l + 1])
        seg = lm_obs[s:e]
        srt = np.argsort(obs_pos[seg], kind="stable")
        gs_obs[s:e] = seg[srt]
        grp = obs_pos[gs_obs[s:e]] // G
        starts = np.flatnonzero(np.concatenate([[True], grp[1:] != grp[:-1]])) if e > s else np.zeros(0, np.int64)
        lens = np.diff(np.concatenate([starts, [e - s]]))
        for ia in range(starts.size):
            for ib in range(ia + 1):
                key.append(int(grp[starts[ia]]) * NG + int(grp[starts[ib]]))
                a0.append(s + int(starts[ia])); b0.append(s + int(starts[ib])); pq.append(int(lens[ia]) | (int(lens[ib]) << 16))
    key = np.array(key, np.uint32); a0 = np.array(a0, np.int32); b0 = np.array(b0, np.int32); pq = np.array(pq, np.int32)
    srt = np.argsort(key, kind="stable")
    key, a0, b0, pq = key[srt], a0[srt], b0[srt], pq[srt]
    starts = np.flatnonzero(np.concatenate([[True], key[1:] != key[:-1]])) if key.size else np.zeros(0, np.int64)
    pair_key = key[starts].astype(np.uint32)
    pair_ptr = np.concatenate([starts, [key.size]]).astype(np.int64)
    order = np.argsort(-(np.diff(pair_ptr)), kind="stable").astype(np.int32)
    gs_obs_packed = (gs_obs.astype(np.int64) | ((obs_pos[gs_obs].astype(np.int64) % G) << 28)).astype(np.uint32).view(np.int32)   # what is uploaded
    return dict(gs_obs=gs_obs, gs_obs_packed=gs_obs_packed, a0=a0, b0=b0, pq=pq, pair_key=pair_key, pair_ptr=pair_ptr, order=order, NG=NG)


def chunks_of(L, j):
    """The kernel's cut of group pair j into chunks: up to CHUNK_CELLS cells, as many of them (in order) as fit NSLOT staged slots;
    a cell of a diagonal group pair stages its entries once (B = A)."""
    ga, gb = int(L["pair_key"][j]) // L["NG"], int(L["pair_key"][j]) % L["NG"]
    c, end = int(L["pair_ptr"][j]), int(L["pair_ptr"][j + 1])
    out = []
    while c < end:
        n = 0; slots = 0
        while c + n < end and n < CHUNK_CELLS:
            p = int(L["pq"][c + n]) & 0xffff; q = (int(L["pq"][c + n]) >> 16) & 0xffff
            need = p if ga == gb else p + q
            if slots + need > NSLOT:
                break
            slots += need; n += 1
        assert n > 0, "a single cell must fit the slot buffer (build_schur_groups refuses the graph otherwise)"
        out.append((c, n, slots)); c += n
    return out


def walk(L, obs_pos):
    """The kernel's walk: per block (row position, column position) the (oa, ob) terms in the order in which they are accumulated."""
    blocks = {}
    for j in L["order"].tolist():
        ga, gb = int(L["pair_key"][j]) // L["NG"], int(L["pair_key"][j]) % L["NG"]
        for c0, n, _ in chunks_of(L, j):
            for w in range(G):                       # wavefront w: row camera G ga + w (the wavefronts run side by side; per block only one writes)
                for c in range(c0, c0 + n):
                    p = int(L["pq"][c]) & 0xffff; q = (int(L["pq"][c]) >> 16) & 0xffff
                    A = L["gs_obs"][L["a0"][c]:L["a0"][c] + p]; B = L["gs_obs"][L["b0"][c]:L["b0"][c] + q]
                    for oa in A.tolist():
                        pa = int(obs_pos[oa])
                        if pa % G != w:
                            continue
                        for ob in B.tolist():
                            pb = int(obs_pos[ob])
                            if ga == gb and pb % G > w:
                                continue
                            blocks.setdefault((pa, pb), []).append((oa, ob))
    return blocks


def pair_major_blocks(problem, obs_pos):
    oa, ob, ptr = _sort_based_term_lists(problem)
    blocks = {}
    for i in range(ptr.size - 1):
        t0, t1 = int(ptr[i]), int(ptr[i + 1])
        blocks[(int(obs_pos[oa[t0]]), int(obs_pos[ob[t0]]))] = list(zip(oa[t0:t1].tolist(), ob[t0:t1].tolist()))
    return blocks


@pytest.mark.parametrize("workload", ["bal:60:6000:7", "bal:300:20000:3"])
def test_grouped_walk_adds_the_same_terms_in_the_same_order(workload):
    problem, _ = HP.problem_for(workload)
    lm_ptr, lm_obs, obs_pos, nrv = incidence(problem)
    L = group_lists(lm_ptr, lm_obs, obs_pos, nrv)
    got = walk(L, obs_pos); want = pair_major_blocks(problem, obs_pos)
    assert got.keys() == want.keys()
    for k in want:
        assert got[k] == want[k], k
    # every chunk fits the LDS buffer, and the lists are as large as the traffic model says
    staged = sum(s for j in range(L["pair_key"].size) for _, _, s in chunks_of(L, j))
    terms = sum(len(v) for v in want.values())
    assert staged < 2 * terms


def test_a_camera_seeing_a_landmark_twice_gets_the_same_terms_in_another_order():
    problem, _ = HP.problem_for("baldup:40:3000:3")
    lm_ptr, lm_obs, obs_pos, nrv = incidence(problem)
    L = group_lists(lm_ptr, lm_obs, obs_pos, nrv)
    got = walk(L, obs_pos); want = pair_major_blocks(problem, obs_pos)
    assert got.keys() == want.keys()
    differ = 0
    for k in want:
        assert sorted(got[k]) == sorted(want[k]), k
        differ += got[k] != want[k]
        if k[0] != k[1]:
            assert got[k] == want[k], k          # only diagonal blocks can hold the four terms of a doubly seen landmark
    assert differ > 0


@pytest.mark.parametrize("workload", ["bal:60:6000:7", "baldup:40:3000:3"])
def test_library_uploads_these_lists(workload):
    if not os.path.exists(os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd.so")):
        pytest.skip("libgtsam_amd.so not built")
    recs = HP.run_snippet(_CHILD % {"root": ROOT, "workload": workload}, env_extra={"GTG_NO_REORDER": "1", "GTG_SCHUR": "groups"})
    have = {(int(n), int(h)) for n, h in recs}
    problem, _ = HP.problem_for(workload)
    lm_ptr, lm_obs, obs_pos, nrv = incidence(problem)
    L = group_lists(lm_ptr, lm_obs, obs_pos, nrv)
    for name in ("gs_obs_packed", "a0", "b0", "pq", "pair_key", "pair_ptr", "order"):
        a = L[name]
        assert (a.nbytes, _fnv(a)) in have, name + " differs"


def test_without_the_switch_nothing_of_it_is_built():
    if not os.path.exists(os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd.so")):
        pytest.skip("libgtsam_amd.so not built")
    recs = HP.run_snippet(_CHILD % {"root": ROOT, "workload": "bal:60:6000:7"}, env_extra={"GTG_NO_REORDER": "1"})
    have = {(int(n), int(h)) for n, h in recs}
    problem, _ = HP.problem_for("bal:60:6000:7")
    lm_ptr, lm_obs, obs_pos, nrv = incidence(problem)
    L = group_lists(lm_ptr, lm_obs, obs_pos, nrv)
    assert (L["a0"].nbytes, _fnv(L["a0"])) not in have
