#!/usr/bin/env python3
"""Drives the HOST code of the library (upload + symbolic analysis, one linearize / try_lambda / PCG issue sequence, plan
getters; single handle and two shards) under tools/hipstub with a sanitizer-instrumented build: tools/sanitize/host_sanitizers.sh.
Development tool; nothing numeric is computed (kernels do not run under the stub)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gtsam_amd import lib as L
L.LIB_PATH = os.environ.get("SAN_LIB", "/tmp/asan/libgtsam_amd.so")
from tools import host_profile as HP
import numpy as np
for w in sys.argv[1:]:
    nd = None
    problem, v0 = HP.problem_for(w)
    for shards in (1, 2):
        for shard in range(shards):
            def lockstep(ptr, n, stream, shards=shards):
                buf = np.frombuffer((ctypes.c_double * n).from_address(ptr), dtype=np.float64); buf *= shards
            g = L.DeviceGraph(problem, shard=shard, n_shards=shards, allreduce=lockstep if shards > 1 else None)
            g.set_values(v0); g.linearize(); rc, out = g.try_lambda(1e-3, True)
            pl = g.cholesky_plan()
            if shards == 1:
                rc2, out2, its = g.try_lambda_pcg(1e-3, True, max_iterations=5)
            g.close()
    print("ok", w, flush=True)
