#!/usr/bin/env python3
"""tools/df_stress.py [seconds] [threads] [problem] -- repeatability of the dataflow Cholesky with several handles alive on one device.

The same optimisation is run once alone (the reference trajectory) and then, for `seconds` of wall clock, from `threads` host
threads at once (one handle each; the library serialises their factorisations per device).  Every trace and every final value must
be BIT-identical to the first one.  Prints one JSON line per differing / failing run and a summary line at the end:
  {"optimisations": N, "different": d, "errors": e, "fallbacks": f, ...}
(fallbacks = lambda tries repeated with the other schedule after a dependency wait ran into its bound)."""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_amd import lib as L  # noqa: E402
from gtsam_amd.optimizer import DeviceLevenbergMarquardt  # noqa: E402
from gtsam_amd.params import LevenbergMarquardtParams as LMP  # noqa: E402


def make(problem):
    if problem == "sphere2500":
        from tests import problems as PB
        from tests.conftest import load_golden
        p, v0 = PB.sphere2500(load_golden("sphere2500"))
        return p, v0, LMP()
    from gtsam_amd import datasets as D
    from gtsam_amd.problem import bal_problem
    p, v0 = bal_problem(*D.synthetic_bal(300, 20000, seed=3))
    prm = LMP.CeresDefaults(); prm.setMaxIterations(6)
    return p, v0, prm


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    problem = sys.argv[3] if len(sys.argv) > 3 else "sphere2500"
    p, v0, prm = make(problem)

    def run(out, i):
        try:
            opt = DeviceLevenbergMarquardt(p, v0, prm)
            opt.optimize()
            ctl = opt.dev.df_ctrl()
            out[i] = (np.array(opt.trace)[:, :3], opt.values_packed(), int(ctl[15]), int(ctl[6]), int(ctl[7]), opt.dev.df_poll_stats())
            opt.dev.close()
        except Exception as e:  # noqa: BLE001
            out[i] = str(e)

    ref = [None]; run(ref, 0)
    assert not isinstance(ref[0], str), ref[0]
    n = diff = err = fb = rnd = shadow = rmw = 0
    poll = np.zeros(5, np.int64)   # long waits, ended on the RMW poll, of those still stale for sc1, ended on the shadow word, of those still stale
    t0 = time.time()
    while time.time() - t0 < seconds:
        res = [None] * nthreads
        threads = [threading.Thread(target=run, args=(res, i)) for i in range(nthreads)]
        for t in threads: t.start()
        for t in threads: t.join()
        for r in res:
            n += 1
            if isinstance(r, str):
                err += 1; print(json.dumps({"round": rnd, "error": r[:200]}), flush=True); continue
            fb += r[2]; shadow += r[3]; rmw += r[4]; poll += np.array(r[5], np.int64)
            if r[0].shape != ref[0][0].shape or not np.array_equal(r[0], ref[0][0]) or not np.array_equal(r[1], ref[0][1]):
                diff += 1
                msg = f"shape {r[0].shape} vs {ref[0][0].shape}"
                if r[0].shape == ref[0][0].shape:
                    d = np.abs(r[0] - ref[0][0]); k = np.unravel_index(np.argmax(d), d.shape)
                    first = int(np.argmax((r[0] != ref[0][0]).any(axis=1)))
                    msg = f"first differing row {first}; max abs {d.max():.3e} at row {k[0]} col {k[1]}; final {r[0][-1, 1]!r} vs {ref[0][0][-1, 1]!r}"
                print(json.dumps({"round": rnd, "differs": msg, "repeated_tries_of_this_run": r[2]}), flush=True)
        rnd += 1
    print(json.dumps({"problem": problem, "lib": os.path.basename(L.LIB_PATH), "threads": nthreads, "seconds": round(time.time() - t0, 1),
                      "optimisations": n, "different": diff, "errors": err, "fallbacks": fb, "waits_ended_on_shadow_words": shadow, "waits_ended_on_rmw_poll": rmw,
                      "waits_of_256_polls_or_more": int(poll[0]), "rmw_ended_and_next_sc1_load_still_old": int(poll[2]), "shadow_ended_and_next_sc1_load_still_old": int(poll[4]),
                      "factorisations_per_optimisation": int(ref[0][0].shape[0])}), flush=True)
    return 0 if diff == 0 and err == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
