#!/usr/bin/env python3
"""tools/df_subtile_count.py [workload] -- what would sub-tile skipping inside the bulk kernel's contraction save?  Runs the library's set-up
under tools/hipstub (no GPU), takes the elimination order and the dataflow task lists, redoes the symbolic factorisation at 16-row /
16-column granularity in numpy and counts the 16 x 16 x 4 MFMAs of the contraction steps:
  executed today (every stored 128 x 128 tile full), with structurally empty 16 x 16 OUTPUT sub-tiles skipped, and with empty operand
  strips skipped as well (exact at 16-granularity), per SIMD imbalance ignored."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tools", "hipstub", "libhipstub.so")
sys.path.insert(0, ROOT)


def main():
    if os.environ.get("LD_PRELOAD", "").find("libhipstub") < 0:
        import tools.host_profile as HP
        HP.build_stub()
        env = dict(os.environ, LD_PRELOAD=STUB, GTG_HOST_ANALYSIS="1", GTG_HOST_ORDERING="1")
        sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env).returncode)
    import tools.host_profile as HP
    w = sys.argv[1] if len(sys.argv) > 1 else "ladybug1723"
    p, _ = HP.problem_for(w)
    from gtsam_amd import lib as L
    g = L.DeviceGraph(p)
    order = g.reduced_order()          # variable id at every position
    pl = g.df_plan()
    nt = pl["nt"]; T = pl["tasks"]; kl = pl["klist"]
    n_red = int(g.reduced_dim)
    # reduced variables of a BAL problem: the cameras, dimension 9 each (tools/host_profile.py workloads ladybug / venice / dubrovnik)
    assert p.n_sfm > 0, "camera systems only"
    ncam = len(order); dim = n_red // ncam
    pos = np.empty(ncam, np.int64); pos[order] = np.arange(ncam)
    off = pos * dim
    cam, pt = np.asarray(p.sfm_cam, np.int64), np.asarray(p.sfm_point, np.int64)
    # camera pairs that share a landmark
    o = np.argsort(pt, kind="stable"); cam_s, pt_s = cam[o], pt[o]
    starts = np.flatnonzero(np.r_[True, pt_s[1:] != pt_s[:-1], True])
    n16 = (nt * 128) // 16
    M = np.zeros((n16, n16), bool)
    def mark(a, b):
        ra0, ra1 = off[a] // 16, (off[a] + dim - 1) // 16
        rb0, rb1 = off[b] // 16, (off[b] + dim - 1) // 16
        for x in (ra0, ra1):
            for y in (rb0, rb1):
                M[max(x, y), min(x, y)] = True
    for s, e in zip(starts[:-1], starts[1:]):
        cs = np.unique(cam_s[s:e])
        for i in range(len(cs)):
            for j in range(i + 1):
                mark(cs[i], cs[j])
    M |= np.eye(n16, dtype=bool)
    # symbolic factorisation at 16-granularity (right-looking)
    Lm = np.tril(M)
    for k in range(n16):
        rows = np.flatnonzero(Lm[k + 1:, k]) + k + 1
        if len(rows):
            Lm[np.ix_(rows, rows)] |= np.tril(np.ones((len(rows), len(rows)), bool))
    # tile-level structure implied by the 16-level one vs the plan's stored tiles
    I, J, koff, kcnt = T[:, 0], T[:, 1], T[:, 2], T[:, 3]
    full = out_skip = exact = 0
    for t in range(len(T)):
        i, j = int(I[t]), int(J[t])
        if i >= nt:      # right-hand-side row: one row tile
            continue
        Lo = Lm[8 * i:8 * i + 8, 8 * j:8 * j + 8] if i != j else np.tril(np.ones((8, 8), bool)) & (Lm[8 * i:8 * i + 8, 8 * j:8 * j + 8] | True)
        for e in range(int(koff[t]), int(koff[t] + kcnt[t])):
            k = int(kl[e])
            A = Lm[8 * i:8 * i + 8, 8 * k:8 * k + 8]; B = Lm[8 * j:8 * j + 8, 8 * k:8 * k + 8]
            full += 64 * 32
            out_skip += int(Lo.sum()) * 32
            # exact: output sub-tile (r, c) x operand strip s (16 columns = 4 MFMA k-steps) needs A[r, s] and B[c, s]
            exact += int((A.astype(np.int64) @ B.astype(np.int64).T)[Lo].sum()) * 4
    print(json.dumps({"workload": w, "nt": int(nt), "mfma_full_tiles": full, "mfma_output_subtiles": out_skip, "mfma_exact_16": exact,
                      "output_subtile_fraction": out_skip / full, "exact_fraction": exact / full}))
    g.close()


if __name__ == "__main__":
    main()
