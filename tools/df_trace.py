#!/usr/bin/env python3
"""tools/df_trace.py -- timeline of one dataflow factorisation (GTG_DF_TRACE=1) on the Ladybug-1723 shape: where the serial
chain spends its time per block column, how busy the bulk workgroups are, task duration statistics.  Writes a JSON summary
(and the raw stamps as .npz next to it when --raw is given)."""
import json
import os
import sys

import numpy as np

os.environ["GTG_DF_TRACE"] = "1"
os.environ.setdefault("GTG_CHOL", "df")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_amd import datasets as D  # noqa: E402
from gtsam_amd import lib as L  # noqa: E402
from gtsam_amd.problem import bal_problem  # noqa: E402


def main():
    p, v0 = bal_problem(*D.ladybug_1723())
    dev = L.DeviceGraph(p)
    dev.set_values(v0); dev.linearize()
    for _ in range(3):
        dev.try_lambda(1e-4, True)
    tasks, chain = dev.df_trace()
    pl = dev.df_plan()
    T = pl["tasks"]; nt = pl["nt"]
    t0 = min(tasks[:, 0].min(), chain[:, 0].min())
    us = lambda x: (x - t0) / 100.0
    start, acc, done = us(tasks[:, 0]), us(tasks[:, 1]), us(tasks[:, 2])
    c_in, c_out = us(chain[:, 0]), us(chain[:, 1])
    total = max(done.max(), c_out.max())
    I, J, kcnt = T[:, 0], T[:, 1], T[:, 3]
    last = T[:, 4] == T[:, 5] - 1
    pd = {int(J[i]): i for i in range(len(T)) if I[i] == J[i] and last[i]}
    sub = {int(J[i]): i for i in range(len(T)) if I[i] == J[i] + 1 and last[i]}
    rows = []
    for j in range(1, nt):
        if j - 1 in sub and j in pd:
            s, d = sub[j - 1], pd[j]
            rows.append([c_out[j - 1] - c_in[j - 1],          # potrf of tile j-1
                         done[s] - c_out[j - 1],               # tile (j, j-1) final after the diagonal tile
                         done[d] - done[s],                    # PD(j) final after that
                         c_in[j] - done[d],                    # chain notices
                         c_in[j] - c_in[j - 1]])               # period
    rows = np.array(rows)
    busy = (done - start).sum(); accum = (acc - start).sum()
    out = {"total_us": float(total), "n_tasks": int(len(T)), "chain_potrf_us_mean": float(rows[:, 0].mean()),
           "sub_after_potrf_us_mean": float(rows[:, 1].mean()), "pd_after_sub_us_mean": float(rows[:, 2].mean()),
           "chain_notice_us_mean": float(rows[:, 3].mean()), "period_us_mean": float(rows[:, 4].mean()),
           "period_us_p10_p50_p90": [float(x) for x in np.percentile(rows[:, 4], [10, 50, 90])],
           "sum_periods_us": float(rows[:, 4].sum()),
           "task_resident_us_total": float(busy), "task_contraction_phase_us_total": float(accum),
           "ksteps_total": int(kcnt.sum()), "us_per_kstep_resident": float(accum / max(kcnt.sum(), 1)),
           "contraction_phase_waiting_us_total": float((tasks[:, 3] >> 40).sum() / 100.0),
           "first_task_start_us": float(start.min()), "chain_first_in_us": float(c_in[0]),
           "workgroup_slots": int(len(set((tasks[:, 3] & ((1 << 40) - 1)).tolist())))}
    print(json.dumps(out))
    if "--raw" in sys.argv:
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "df_trace_raw.npz"),
                            tasks=tasks, chain=chain, plan=T)
    dev.close()


if __name__ == "__main__":
    main()
