"""Host side of gtg_upload_problem (symbolic analysis, schedule construction, table uploads) without a GPU.

The product library runs in a child process under tools/hipstub -- a dry-run HIP runtime in which "device" memory is host
memory and kernels do not run -- so only the HOST code executes; nothing numeric is computed there (it is not a CPU
fallback and the product never loads it).  The stub records a hash of every host-to-device copy; the sorted records are
the "signature" of everything the analysis built: factor tables, CSR incidence lists, the Schur block/term lists
(the summation order of the device), the RCM order and the tile schedule of the Cholesky.

Checked here: the analysis is deterministic and independent of the number of host threads (the device sums follow the
uploaded term order, so this is what keeps results bit-reproducible), also per shard and for a landmark seen twice by
one camera (the general path of the term generator).
"""
import copy
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import host_profile as HP  # noqa: E402


def _signature(workload, shards=1, threads=None):
    env = {} if threads is None else {"GTG_HOST_THREADS": str(threads)}
    rec = HP.run(workload, shards=shards, reps=1, env_extra=env, quiet=True)
    return [(r["signature"], r["reduced_dim"], round(r["cholesky_gflop"], 9), r["h2d_bytes"]) for r in rec["runs"]]


@pytest.fixture(scope="module")
def stub():
    if not os.path.exists(os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd.so")):
        pytest.skip("libgtsam_amd.so not built")
    return HP.build_stub()


def test_analysis_independent_of_host_threads(stub):
    w = "bal:60:6000:7"           # 60 cameras, 6000 points: ~26k observations -> more than one analysis thread
    one = _signature(w, threads=1)
    assert one == _signature(w, threads=3) == _signature(w, threads=8)
    assert one[0][1] == 60 * 9


def test_sharded_analysis_independent_of_host_threads(stub):
    w = "bal:60:6000:7"
    a = _signature(w, shards=2, threads=1)
    b = _signature(w, shards=2, threads=5)
    assert a == b and len(a) == 2
    assert a[0][0] != a[1][0]            # the two shards own different landmarks
    assert a[0][1] == a[1][1] == 540     # every shard holds the whole reduced system


def test_pose_graphs_and_fixture_sizes(stub):
    s = _signature("sphere2500", threads=2)
    assert s[0][1] == 2500 * 6
    assert s == _signature("sphere2500", threads=1)
    d = _signature("dubrovnik_3_7")
    assert d[0][1] == 27


def test_camera_seeing_a_landmark_twice(stub):
    # duplicate some observations: the same (camera, landmark) pair twice exercises the diagonal-block terms
    # (E_a E_b^T and E_b E_a^T) of the generator; must not depend on the thread count either
    assert _signature("baldup:40:3000:3", threads=1) == _signature("baldup:40:3000:3", threads=4)


def test_world_size_2_gloo_sharded_upload_with_the_real_library(stub):
    """The N > 1 set-up path end to end on CPU: two `gloo` ranks run the library's sharded gtg_upload_problem under the
    stub with the REAL all-reduce callback (gtsam_amd.distributed.make_allreduce).  Every shard only has the Schur
    blocks of its own landmarks, but ordering, offsets, padding, tile schedule and exchange list are derived from the
    whole graph: both ranks must report the layout of the single-handle run, and the library's own consistency
    exchange must pass.  (A layout derived from the shard's own blocks differs between shards as soon as the reduced
    system is reordered, i.e. from 16 cameras up -- the summed buffers would not line up.)"""
    w = "bal:60:6000:7"
    recs = HP.run_gloo(w, world=2)
    assert [r["ok"] for r in recs] == [True, True]
    single = HP.run(w, shards=1, reps=1, quiet=True)["runs"][0]
    assert recs[0]["structure_hash"] == recs[1]["structure_hash"] == single["structure_hash"]
    assert recs[0]["cholesky_gflop"] == recs[1]["cholesky_gflop"] == single["cholesky_gflop"]


def test_sharded_upload_fails_loudly_when_ranks_and_shards_do_not_match(stub):
    recs = HP.run_gloo("bal:60:6000:7", world=2, claim_shards=3)     # the all-reduce spans 2 ranks, the library is told 3
    assert all(not r["ok"] and "disagree on the layout" in r["error"] for r in recs)


def test_three_shards_and_pose_graph_layouts(stub):
    recs = HP.run_gloo("sphere2500", world=3)                         # between factors go round-robin
    single = HP.run("sphere2500", shards=1, reps=1, quiet=True)["runs"][0]
    assert all(r["ok"] and r["structure_hash"] == single["structure_hash"] for r in recs)
