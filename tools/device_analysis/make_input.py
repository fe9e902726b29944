#!/usr/bin/env python3
"""Writes the input of tools/device_analysis/proto.hip for a workload: the landmark -> observation CSR and the observations'
camera positions (what the device kernels read), followed by the term lists the HOST analysis builds for them (pair_oa,
pair_ob, pair_ptr; computed with the numpy statement of tests/test_device_analysis_spec.py, which is pinned bit for bit
against the library's uploads).  Layout: int64 head[5] = n_lm, n_obs, nrv, n_terms, n_blocks; int64 ptr[n_lm + 1];
int32 lm_obs[n_obs]; int32 obs_pos[n_obs]; int32 pair_oa[n_terms]; int32 pair_ob[n_terms]; int64 pair_ptr[n_blocks + 1].

    python tools/device_analysis/make_input.py ladybug1723 /tmp/l1723.bin
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build(problem):
    from tests.test_device_analysis_spec import _sort_based_term_lists
    oa, ob, pptr = _sort_based_term_lists(problem)
    red_vars = np.where(problem.var_type != 2)[0]; lm_vars = np.where(problem.var_type == 2)[0]
    red_index = -np.ones(problem.n_vars, np.int64); red_index[red_vars] = np.arange(red_vars.size)
    lm_index = -np.ones(problem.n_vars, np.int64); lm_index[lm_vars] = np.arange(lm_vars.size)
    pos = red_index[problem.sfm_cam.astype(np.int64)]; lm = lm_index[problem.sfm_point.astype(np.int64)]
    order = np.argsort(lm, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(np.bincount(lm, minlength=lm_vars.size))]).astype(np.int64)
    head = np.array([lm_vars.size, pos.size, red_vars.size, oa.size, pptr.size - 1], np.int64)
    return head, ptr, order.astype(np.int32), pos.astype(np.int32), oa, ob, pptr


def main():
    from tools import host_profile as HP
    workload, out = sys.argv[1], sys.argv[2]
    problem, _ = HP.problem_for(workload)
    parts = build(problem)
    with open(out, "wb") as f:
        for a in parts:
            f.write(np.ascontiguousarray(a).tobytes())
    print(f"{workload}: {parts[0].tolist()} -> {out} ({os.path.getsize(out) / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
