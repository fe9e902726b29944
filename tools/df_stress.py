#!/usr/bin/env python3
"""tools/df_stress.py -- repeatability of the dataflow Cholesky under contention: the same optimisation run alone and from two
host threads at once (two handles = four persistent kernels competing for the CUs), several rounds; every trace must be
bit-identical to the first one.  Prints one JSON line per round."""
import json
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_amd.optimizer import DeviceLevenbergMarquardt  # noqa: E402
from gtsam_amd.params import LevenbergMarquardtParams as LMP  # noqa: E402
from tests import problems as PB  # noqa: E402
from tests.conftest import load_golden  # noqa: E402


def main():
    g = load_golden("sphere2500")
    p, v0 = PB.sphere2500(g)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6

    def run(out, i):
        try:
            opt = DeviceLevenbergMarquardt(p, v0, LMP())
            opt.optimize()
            out[i] = np.array(opt.trace)[:, :3]
        except Exception as e:  # noqa: BLE001
            out[i] = str(e)

    ref = [None]; run(ref, 0)
    for rnd in range(rounds):
        both = [None, None, None]
        threads = [threading.Thread(target=run, args=(both, i)) for i in range(3)]
        for t in threads: t.start()
        for t in threads: t.join()
        res = []
        for tr in both:
            if isinstance(tr, str): res.append("error: " + tr[:100])
            elif tr.shape != ref[0].shape: res.append(f"shape {tr.shape} vs {ref[0].shape}")
            elif np.array_equal(tr, ref[0]): res.append("identical")
            else:
                d = np.abs(tr - ref[0]); k = np.unravel_index(np.argmax(d), d.shape)
                res.append(f"differs: max abs {d.max():.3e} at row {k[0]} col {k[1]} (rel {d.max() / max(abs(ref[0][k]), 1e-300):.2e})")
        print(json.dumps({"round": rnd, "results": res}), flush=True)


if __name__ == "__main__":
    main()
