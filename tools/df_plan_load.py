#!/usr/bin/env python3
"""Work per ticket-order iteration of the dataflow factorisation's plan against what the bulk kernel gets done in one chain period.

The library builds its plan for the workload in a child process under tools/hipstub (host code only, no GPU); iteration q of the ticket
order = the final pieces of block column q + the early pieces queued behind column q - 1.  Cost model: 17 us per 128^3 contraction step
+ 8 us per task (round-5 trace), capacity = 246 workgroups x 40 us (the chain period).

    python tools/df_plan_load.py [workload]

(Round 6 used it to judge a load-levelling pass over the early pieces -- a GTG_DF_LEVEL switch in build_df_plan_host, since removed: the
levelled plan was slower on the hardware, profiles/r06i_level_sweep.txt.)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import host_profile as HP  # noqa: E402

CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, %r)
from tools import host_profile as HP
from gtsam_amd import lib as L
problem, _ = HP.problem_for(%r)
g = L.DeviceGraph(problem)
df = g.df_plan()
print("RESULT " + json.dumps({"nt": int(df["nt"]), "tasks": np.asarray(df["tasks"]).tolist(), "klist": np.asarray(df["klist"]).tolist()}))
'''


def profile(workload):
    d = HP.run_snippet(CHILD % (ROOT, workload))
    T = np.array(d["tasks"], np.int64).reshape(-1, 6); nt = d["nt"]
    q = 0; it = np.zeros(len(T), np.int64)
    for t, (I, J, ko, kc, r, R) in enumerate(T):       # a final piece of a tile away from the chain marks its column's iteration
        if r == R - 1 and I - J > 3 and I < nt:
            q = max(q, J)
        it[t] = q
    load = np.bincount(it, weights=T[:, 3] * 17.0 + 8.0, minlength=nt)
    cap = 40.0 * 246
    return {"workload": workload, "tasks": int(len(T)), "steps": int(T[:, 3].sum()),
            "bulk_work_ms": round(float(load.sum()) / 246 / 1e3, 3), "iterations_over_capacity": int((load > cap).sum()),
            "excess_over_capacity_ms": round(float(np.maximum(load - cap, 0).sum()) / 246 / 1e3, 3),
            "max_load_over_capacity": round(float(load.max() / cap), 2),
            "load_percent_of_capacity_by_iteration": (100 * load / cap).astype(int).tolist()}


if __name__ == "__main__":
    w = sys.argv[1] if len(sys.argv) > 1 else "ladybug1723"
    print(json.dumps(profile(w)), flush=True)
