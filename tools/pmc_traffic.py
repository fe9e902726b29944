#!/usr/bin/env python3
"""Reduce the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; summaries from tools/rocprof_pmc.py) to HBM bytes per
Cholesky factorisation and per launch of the other hot kernels.  FETCH_SIZE / WRITE_SIZE count kilobytes... units and
the gfx950 correction exactly as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: counters
are in KB, and FETCH_SIZE is doubled on gfx950.
usage: tools/pmc_traffic.py fetch.csv write.csv out.json"""
import csv
import json
import sys


def load(path):
    out = {}
    for row in csv.DictReader(open(path)):
        out[row["kernel"]] = (int(row["dispatches"]), float(row["sum"]))
    return out


fetch, write = load(sys.argv[1]), load(sys.argv[2])
CHOL = ("k_panel128", "k_potrf128", "k_trsm128", "k_syrk", "k_df_single", "k_df_bulk", "k_df_chain")


def total(tab, pred):
    return sum(s for k, (n, s) in tab.items() if pred(k))


def is_chol(k):
    return any(c in k for c in CHOL)


# dataflow schedule (default): one k_df_single dispatch per factorisation in the profiled form (GTG_DF_SINGLE=1: rocprofv3's counter
# collection serialises kernels, so the two cooperating kernels of the production form cannot run under it; same tasks, same
# tiles read and written, one workgroup per CU instead of a separate chain kernel); stream / event schedule: one k_panel128 per
# block column
ndf = max((n for k, (n, s) in fetch.items() if "k_df_single" in k), default=0)
nfac = max((n for k, (n, s) in fetch.items() if "k_panel128" in k or "k_potrf128" in k), default=0)
nt = 122.0   # block columns of the L1723-shaped reduced system (15 507 / 128, rounded up)
facs = float(ndf) if ndf else (nfac / nt if nfac else 0.0)
fb, wb = total(fetch, is_chol) * 1024.0, total(write, is_chol) * 1024.0
res = {
    "unit": "bytes per tile-sparse Cholesky factorisation (dataflow schedule, profiled as k_df_single; or k_panel128 + k_syrk<1|2> with GTG_CHOL=streams), RCM-ordered L1723 shape",
    "FETCH_SIZE_raw_bytes": fb / facs if facs else None, "WRITE_SIZE_bytes": wb / facs if facs else None,
    "fetch_corrected_bytes": 2.0 * fb / facs if facs else None,
    "hbm_bytes_sparse": (2.0 * fb + wb) / facs if facs else None,
    "factorisations_in_profile": facs,
    "note": "FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE in KB; separate --pmc passes of "
            "`python bench.py --steps 2 --warmup 0 --cpu-baseline off --skip-dense-roofline`",
    "per_launch_other_kernels_bytes": {},
}
for k, (n, s) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
    if is_chol(k) or n == 0:
        continue
    w = write.get(k, (n, 0.0))
    name = k.split("(")[0]
    res["per_launch_other_kernels_bytes"][name] = {"launches": n, "read": 2.0 * s * 1024.0 / n, "written": w[1] * 1024.0 / max(w[0], 1)}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "per_launch_other_kernels_bytes"}, indent=1))
