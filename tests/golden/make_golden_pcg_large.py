#!/usr/bin/env python3
"""tests/golden/make_golden_pcg_large.py -- the reference's ITERATIVE solver at the headline size.

    python tests/golden/make_golden_pcg_large.py [max_iterations [epsilon_rel]]

Runs borglab/gtsam's own PCGSolver + BlockJacobiPreconditioner (gtsam/linear/PCGSolver.cpp:51-64,
Preconditioner.cpp) through oracle/_ref on the damped system of the FIRST lambda try (lambda = 1e-4, diagonal
damping, the Ceres preset of timing/timeSFMBAL.h) of the seeded L1723-shaped problem of gtsam_amd/datasets.py, and
writes tests/golden/ladybug1723_pcg.npz: the step (camera part in full, every 97th landmark entry, norms), the
solver settings, and its distance from the reference's DIRECT step of the same system (ladybug1723.npz) -- which is
how far the reference's own CG was from converged when it stopped, i.e. the tolerance a parity test can ask for.
Build container only (needs /root/reference and `make -C oracle ref`)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gtsam_amd import datasets as D  # noqa: E402
from gtsam_amd.problem import bal_problem  # noqa: E402
from oracle import ref  # noqa: E402
from tests.golden.make_golden_large import checksum, compress, OUT, STRIDE  # noqa: E402


def main():
    max_it = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    eps_rel = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
    lam = 1e-4
    p, v0 = bal_problem(*D.ladybug_1723())
    nC = int((p.var_type == 1).sum())
    g = ref.RefGraph(p)
    t0 = time.time()
    d = g.solve_pcg(v0, lam, True, max_iterations=max_it, epsilon_rel=eps_rel, epsilon_abs=1e-300)
    secs = time.time() - t0
    gold = np.load(os.path.join(OUT, "ladybug1723.npz"))
    assert checksum(v0, p.sfm_cam, p.sfm_point, p.sfm_z) == gold["checksum"]
    dist_cam = np.abs(d[:9 * nC] - gold["delta_cam"]).max() / float(gold["delta_norminf"])
    dist_lm = np.abs(d[9 * nC::STRIDE] - gold["delta_lm_sample"]).max() / float(gold["delta_norminf"])
    out = dict(checksum=gold["checksum"], n_cams=nC, stride=STRIDE, solve_lambda=lam, max_iterations=max_it,
               epsilon_rel=eps_rel, epsilon_abs=1e-300, ref_seconds=secs, dist_from_direct=max(dist_cam, dist_lm))
    for k, v in compress(d, nC * 9).items():
        out["delta_" + k] = v
    np.savez_compressed(os.path.join(OUT, "ladybug1723_pcg.npz"), **out)
    print("reference PCG:", secs, "s; distance from the reference's direct step (max-norm, relative):", dist_cam, dist_lm, flush=True)


if __name__ == "__main__":
    main()
