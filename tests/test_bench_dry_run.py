"""bench.py's control flow for N = 1 and N = 2 without a GPU (tests/bench_dry_run.py: stub + gloo; numbers meaningless)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import host_profile as HP  # noqa: E402

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline", "time_to_converged_s", "time_to_converged_setup_s"}


@pytest.mark.parametrize("world,mode", [(1, "shard"), (2, "shard"), (2, "speculative")])
def test_bench_contract_and_multi_rank_flow(world, mode):
    if not os.path.exists(os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd.so")):
        pytest.skip("libgtsam_amd.so not built")
    stub = HP.build_stub()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, LD_PRELOAD=stub, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BENCH_PARALLELISM=mode)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_dry_run.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    lines = [line for line in outs[0][0].splitlines() if line.startswith("{")]
    assert len(lines) == 1                                              # rank 0 prints ONE JSON line
    assert all(not any(line.startswith("{") for line in so.splitlines()) for so, _ in outs[1:])   # the other ranks print none
    rec = json.loads(lines[0])
    assert KEYS <= set(rec) and rec["n_gpus"] == world and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["metric"] == "LM iterations/sec" and rec["unit"] == "iterations/s" and rec["higher_is_better"] is True
    assert rec["scaling"] == "strong" and rec["dtype"] == "f64" and rec["data"] == "synthetic" and rec["vs_baseline"] is None
    assert "workload" in rec["config"] and "model" not in rec["config"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rec["roofline"])
    if world > 1:     # the other modes ride along in the same line (bench.py --extra-modes auto)
        other = "speculative" if mode == "shard" else "shard"
        assert set(rec["extra_modes"]) == {other, "pcg_shard"}, rec["extra_modes"]
        for k, v in rec["extra_modes"].items():
            assert "failed" not in v and v["steps"] >= 2 and v["value"] > 0, (k, v)
    else:
        assert rec["extra_modes"] is None
    assert {"lambda_tries", "tries_per_iteration", "device_phase_ms_per_iteration"} <= set(rec)
    assert rec["config"]["parallelism"].startswith("single GPU" if world == 1 else ("speculative-lambda x2" if mode == "speculative" else "landmark-shard x2"))
