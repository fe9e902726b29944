#!/usr/bin/env python3
"""tools/df_wait_analysis.py -- where the bulk workgroups of the dataflow Cholesky spend their resident time (GTG_DF_TRACE=1, L1723): per kind of
task the contraction phase, the flag waits inside it (tr[3] >> 40) and what follows the contraction (substitution / hand-over)."""
import os, sys, json
import numpy as np
os.environ["GTG_DF_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gtsam_amd import lib as L
from gtsam_amd.params import LevenbergMarquardtParams
(p, v0), _ = bench.build_workload("ladybug1723")
prm = LevenbergMarquardtParams.CeresDefaults()
dev = L.DeviceGraph(p); dev.set_values(v0); dev.linearize()
for _ in range(3): dev.try_lambda(prm.lambdaInitial, prm.diagonalDamping, prm.minDiagonal, prm.maxDiagonal)
tasks, chain = dev.df_trace(); pl = dev.df_plan(); T = pl["tasks"]
I, J, kc, r, R = T[:,0], T[:,1], T[:,3], T[:,4], T[:,5]
waited = (tasks[:,3] >> 40) / 100.0
res = (tasks[:,2] - tasks[:,0]) / 100.0
con = (tasks[:,1] - tasks[:,0]) / 100.0
fin = r == R - 1
out = {"total_waited_us": float(waited.sum()), "resident_us": float(res.sum())}
def grp(name, m):
    out[name] = {"tasks": int(m.sum()), "ksteps": int(kc[m].sum()), "resident_us": round(float(res[m].sum()),0), "contraction_us": round(float(con[m].sum()),0), "waited_us": round(float(waited[m].sum()),0),
                 "after_contraction_us": round(float((res[m]-con[m]).sum()),0)}
nt = pl["nt"]
grp("early pieces", ~fin)
grp("final pieces, diagonal (PD)", fin & (I == J))
grp("final pieces, I=J+1 (chain's tile)", fin & (I == J + 1))
grp("final pieces, I-J in 2..4", fin & (I - J >= 2) & (I - J <= 4) & (I < nt))
grp("final pieces, I-J >= 5", fin & (I - J >= 5) & (I < nt))
grp("rhs row", I == nt)
# waiting per k-step for early pieces by distance of youngest operand
print(json.dumps(out, indent=1))
dev.close()
