"""The grouped Schur kernel's OWN source text (csrc/schur_groups_kernel.h) executed on host threads (tools/kernel_emu: one thread per
work-item, __syncthreads / __shfl_up / __ballot / the f64 MFMA as rendezvous of the threads of a workgroup / wavefront), against the
pair-major sums of k_schur_pairs formed with the same arithmetic: the reduced system must come out BIT-identical, which checks what the
numpy statement (tests/test_schur_groups_spec.py) cannot -- the kernel's chunk cut, its staging indices, the LDS tables, the
accumulator switch and the write-out.  The padding of every E slot and the unused entries of a 6-dimensional camera's slots are NaN
here: a lane that used one would poison its block."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import host_profile as HP  # noqa: E402
from tests.test_device_analysis_spec import _sort_based_term_lists  # noqa: E402
from tests.test_schur_groups_spec import G, group_lists, incidence  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SRC = os.path.join(ROOT, "tools", "kernel_emu", "schur_groups_emu.cpp")
LIB = os.path.join(ROOT, "tests", "_build", "libschur_groups_emu.so")


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ to build the kernel emulator with")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    deps = [SRC, os.path.join(ROOT, "tools", "kernel_emu", "emu_hip.h"), os.path.join(ROOT, "gtsam_amd", "csrc", "schur_groups_kernel.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(d) for d in deps):
        subprocess.run([CLANG, "-std=c++20", "-O2", "-pthread", "-fPIC", "-shared", "-Wno-psabi", "-o", LIB, SRC], check=True)
    return ctypes.CDLL(LIB)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("workload,mixed_dims,pipelined", [("bal:40:2500:3", True, 1), ("bal:20:400:1", True, 0), ("bal:20:400:1", False, 1)])
def test_emulated_kernel_reproduces_the_pair_major_sums_bit_for_bit(emu, workload, mixed_dims, pipelined):
    problem, _ = HP.problem_for(workload)
    lm_ptr, lm_obs, obs_pos, nrv = incidence(problem)
    L = group_lists(lm_ptr, lm_obs, obs_pos, nrv)
    rng = np.random.default_rng(5)
    red_dim = np.full(nrv, 9, np.int32)
    if mixed_dims:
        red_dim[rng.random(nrv) < 0.4] = 6
    red_off = np.concatenate([[0], np.cumsum(red_dim)[:-1]]).astype(np.int64)
    NP = int(red_dim.sum())
    n_obs = lm_obs.size
    E = np.full((n_obs, 32), np.nan)
    for o in range(n_obs):
        d = int(red_dim[obs_pos[o]])
        E[o, :3 * d] = rng.standard_normal(3 * d)
    pos_red = np.arange(nrv, dtype=np.int32)            # positions = the caller's order (GTG_NO_REORDER)
    S1 = np.zeros((NP, NP)); S2 = np.zeros((NP, NP))
    emu.emu_schur_groups.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 12 + [ctypes.c_int64, ctypes.c_int]
    emu.emu_schur_pairs_reference.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 9 + [ctypes.c_int64]
    pk = L["pair_key"].astype(np.int32)
    rc = emu.emu_schur_groups(int(pk.size), int(L["NG"]), nrv, _ptr(L["order"]), _ptr(pk), _ptr(L["pair_ptr"]), _ptr(L["a0"]), _ptr(L["b0"]),
                              _ptr(L["pq"]), _ptr(L["gs_obs_packed"]), _ptr(pos_red), _ptr(red_dim), _ptr(red_off), _ptr(E), _ptr(S1), NP, pipelined)
    assert rc == 0
    oa, ob, ptr = _sort_based_term_lists(problem)
    prow = obs_pos[oa[ptr[:-1]]].astype(np.int32); pcol = obs_pos[ob[ptr[:-1]]].astype(np.int32)
    emu.emu_schur_pairs_reference(int(prow.size), _ptr(prow), _ptr(pcol), _ptr(ptr), _ptr(oa), _ptr(ob), _ptr(red_dim), _ptr(red_off), _ptr(E), _ptr(S2), NP)
    assert np.isfinite(S1).all() and np.isfinite(S2).all()
    assert np.abs(S2).max() > 0
    assert np.array_equal(S1, S2), float(np.abs(S1 - S2).max())
    # and both are the Schur complement's sums: dense float64 reference, summation order aside
    D = np.zeros((NP, NP))
    Ez = np.nan_to_num(E[:, :27]).reshape(n_obs, 9, 3)
    for l in range(lm_ptr.size - 1):
        obs = lm_obs[lm_ptr[l]:lm_ptr[l + 1]]
        for a in obs:
            for b in obs:
                pa, pb = int(obs_pos[a]), int(obs_pos[b])
                if pa > pb or (pa == pb):
                    da, db = int(red_dim[pa]), int(red_dim[pb])
                    D[red_off[pa]:red_off[pa] + da, red_off[pb]:red_off[pb] + db] -= Ez[a, :da] @ Ez[b, :db].T
    assert np.abs(S1 - D).max() <= 1e-11 * max(1.0, np.abs(D).max())


@pytest.mark.parametrize("workload", ["bal:40:2500:3", "baldup:40:3000:3"])
def test_emulated_list_kernels_build_the_stated_lists(emu, workload):
    """The device version of the list builder (schur_groups.hip::device_schur_groups, GTG_SCHUR_LISTS=device): its two per-landmark kernels run
    on host threads, the stable sort by group pair done in numpy -- the sorted observation lists and the cells are those of the statement."""
    problem, _ = HP.problem_for(workload)
    lm_ptr, lm_obs, obs_pos, nrv = incidence(problem)
    L = group_lists(lm_ptr, lm_obs, obs_pos, nrv)
    n_lm = lm_ptr.size - 1; n_obs = lm_obs.size
    obs_red = obs_pos.copy()                              # positions = the caller's order: observation -> reduced variable -> itself
    red_pos = np.arange(nrv, dtype=np.int32)
    gobs = np.zeros(n_obs, np.int32); gpos = np.zeros(n_obs, np.int32); cnt = np.zeros(n_lm + 1, np.int64)
    emu.emu_sg_sort_count.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 7
    emu.emu_sg_emit.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 9
    emu.emu_sg_sort_count(n_lm, _ptr(lm_ptr), _ptr(lm_obs), _ptr(obs_red), _ptr(red_pos), _ptr(gobs), _ptr(gpos), _ptr(cnt))
    assert np.array_equal(gobs, L["gs_obs_packed"])
    off = np.concatenate([[0], np.cumsum(cnt[:-1])]).astype(np.int64)
    total = int(cnt.sum())
    assert total == L["a0"].size
    key = np.zeros(total, np.uint32); idx = np.zeros(total, np.uint32); a0 = np.zeros(total, np.int32); b0 = np.zeros(total, np.int32); pq = np.zeros(total, np.int32)
    bad = np.zeros(4, np.int32)
    emu.emu_sg_emit(n_lm, int(L["NG"]), _ptr(lm_ptr), _ptr(gpos), _ptr(off), _ptr(key), _ptr(idx), _ptr(a0), _ptr(b0), _ptr(pq), _ptr(bad))
    assert bad[0] == 0 and np.array_equal(idx, np.arange(total, dtype=np.uint32))
    srt = np.argsort(key, kind="stable")                  # = rocprim::radix_sort_pairs (stable) + k_sg_gather
    assert np.array_equal(a0[srt], L["a0"]) and np.array_equal(b0[srt], L["b0"]) and np.array_equal(pq[srt], L["pq"])
    ks = key[srt]
    starts = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    assert np.array_equal(ks[starts], L["pair_key"]) and np.array_equal(np.concatenate([starts, [total]]), L["pair_ptr"])
