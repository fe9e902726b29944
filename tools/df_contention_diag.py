#!/usr/bin/env python3
"""tools/df_contention_diag.py ROUNDS THREADS MODE [sphere|bal300] -- the contention test of tests/test_gpu_dataflow_protocol.py
as a rate measurement with a post-mortem.  MODE: close (every optimizer's handle is destroyed when it is done: its device memory is
reused by the next one), keep (handles stay alive: no reuse of device memory), gc (left to the garbage collector).
One JSON line: runs, time-outs (with which dependency wait gave up: gtg_debug_df_ctrl), different trajectories."""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_amd import lib as L  # noqa: E402
from gtsam_amd.optimizer import DeviceLevenbergMarquardt  # noqa: E402
from tests.test_gpu_dataflow_protocol import _sphere, _bal300  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    mode = sys.argv[3] if len(sys.argv) > 3 else "close"
    which = sys.argv[4] if len(sys.argv) > 4 else "sphere"
    p, v0, prm = (_sphere if which == "sphere" else _bal300)()
    kept = []
    if os.environ.get("DIAG_SERIALIZE"):   # one API call of ANY handle at a time: no kernels of another handle beside a factorisation
        big = threading.Lock()
        for name in ("linearize", "try_lambda", "error", "set_values", "accept", "values", "delta"):
            orig = getattr(L.DeviceGraph, name)
            def wrap(orig):
                def f(self, *a, **k):
                    with big:
                        return orig(self, *a, **k)
                return f
            setattr(L.DeviceGraph, name, wrap(orig))

    def run(out, i):
        opt = None
        try:
            opt = DeviceLevenbergMarquardt(p, v0, prm)
            opt.optimize()
            out[i] = np.array(opt.trace)[:, :3]
        except Exception as e:  # noqa: BLE001
            ctrl = opt.dev.df_ctrl().tolist() if opt is not None else None
            out[i] = f"{str(e)[:60]} ctrl={ctrl} it={opt.iterations() if opt else None}"
        finally:
            if opt is not None:
                if mode == "close": opt.dev.close()
                elif mode == "keep": kept.append(opt)

    def run_interleaved(res):
        """ONE host thread, nth live handles, their iterate() calls interleaved (round robin)."""
        opts = [DeviceLevenbergMarquardt(p, v0, prm) for _ in range(nth)]
        nit = len(ref[0]) - 1
        try:
            for _ in range(nit):
                for o in opts: o.iterate()
            for i, o in enumerate(opts): res[i] = np.array(o.trace)[:, :3]
        except Exception as e:  # noqa: BLE001
            for i in range(nth): res[i] = str(e)[:80]
        for o in opts: o.dev.close()

    ref = [None]; run(ref, 0)
    t0 = time.time()
    n = 0; errors = []; diffs = []
    for rnd in range(rounds):
        res = [None] * nth
        if mode == "interleave":
            run_interleaved(res)
        else:
            th = [threading.Thread(target=run, args=(res, i)) for i in range(nth)]
            for t in th: t.start()
            for t in th: t.join()
        for tr in res:
            n += 1
            if isinstance(tr, str): errors.append((rnd, tr))
            elif tr.shape != ref[0].shape: diffs.append((rnd, str(tr.shape)))
            elif not np.array_equal(tr, ref[0]):
                d = np.abs(tr - ref[0]) / np.maximum(np.abs(ref[0]), 1e-300)
                k = np.unravel_index(np.argmax(d), d.shape)
                first = int(np.flatnonzero((tr != ref[0]).any(1))[0])
                diffs.append((rnd, f"first differing row {first} of {tr.shape[0]}, max rel diff {d.max():.3e} at {k}"))
    print(json.dumps({"lib": os.path.basename(L.LIB_PATH), "mode": mode, "threads": nth, "problem": which, "runs": n,
                      "seconds": round(time.time() - t0, 2), "timeouts": len(errors), "different": len(diffs),
                      "errors": errors[:4], "diffs": diffs[:6], "serialize": bool(os.environ.get("DIAG_SERIALIZE")), "policy": os.environ.get("GTG_DF_POLICY", "auto"), "sched": os.environ.get("GTG_CHOL", "dataflow")}), flush=True)


if __name__ == "__main__":
    main()
