#!/usr/bin/env python3
"""tools/smart_probe.py -- timing/timeSFMBALsmart.cpp's protocol on a synthetic scene with a sane field of view (150 cameras on an arc,
20 000 tracks, 218 262 measurements; the synthetic L1723 shape has measurements hundreds of focal lengths off the axis, where
Cal3Bundler::calibrate does not converge and the reference throws): one SmartProjectionFactor per track, Ceres LM parameters.
Prints one JSON line; the reference (oracle/_ref, 1 thread) takes 6.1 s for the same optimisation in the build container and ends
at 47 120.9186."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_amd import datasets as D  # noqa: E402
from gtsam_amd.optimizer import DeviceLevenbergMarquardt  # noqa: E402
from gtsam_amd.params import LevenbergMarquardtParams as LMP  # noqa: E402
from gtsam_amd.problem import smart_bal_problem  # noqa: E402


def main():
    cams, pts, oc, op, oz = D.synthetic_orbit_scene(n_cams=150, n_points=20000, seed=9, see=0.06)
    p, v0 = smart_bal_problem(cams, oc, op, oz, min_observations=2)
    runs = []
    for rep in range(3):
        t0 = time.perf_counter()
        opt = DeviceLevenbergMarquardt(p, v0, LMP.CeresDefaults())
        t1 = time.perf_counter()
        opt.dev.enable_timing(True); opt.dev.reset_timing()
        opt.optimize()
        t2 = time.perf_counter()
        runs.append({"setup_s": t1 - t0, "optimize_s": t2 - t1, "iterations": opt.iterations(), "inner": opt.getInnerIterations(),
                     "error0": opt.trace[0][1], "error": opt.error(),
                     "phase_ms_total": {k: round(v[0], 3) for k, v in opt.dev.phase_ms().items()}})
        opt.dev.close()
    print(json.dumps({"smart_factors": int(p.n_smart), "measurements": int(p.smart_cam.size), "cameras": int(p.n_vars),
                      "reference_final_error": 47120.9186, "reference_seconds_1_thread_build_container": 6.14, "runs": runs}))


if __name__ == "__main__":
    main()
