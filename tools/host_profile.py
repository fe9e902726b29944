#!/usr/bin/env python3
"""Host-side profile of gtg_create + gtg_upload_problem (symbolic analysis, schedule construction, table uploads) WITHOUT a GPU.

Runs the product library under tools/hipstub (a dry-run HIP runtime: device memory = host memory, kernels do not run), so
only host code executes.  Prints the setup breakdown (GTG_DEBUG_TIMING) and a signature of everything the library
uploaded (sorted (bytes, hash) records): two builds that print the same signature built identical device tables.

    python tools/host_profile.py [ladybug1723|venice1778|dubrovnik16|sphere2500|w20000|dubrovnik_3_7] [--shards N] [--sig-only]

Development tool only -- never used by the product path, the tests' GPU legs, smoke() or bench.py.
"""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tools", "hipstub", "libhipstub.so")


def build_stub():
    src = os.path.join(ROOT, "tools", "hipstub", "hipstub.c")
    if not os.path.exists(STUB) or os.path.getmtime(STUB) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", STUB, src], check=True)
    return STUB


def problem_for(name):
    sys.path.insert(0, ROOT)
    import numpy as np
    from gtsam_amd import datasets as D
    from gtsam_amd.problem import bal_problem
    from tests import problems as PB
    gold = os.path.join(ROOT, "tests", "golden")
    if name == "ladybug1723": return bal_problem(*D.ladybug_1723())
    if name == "streets1723": return bal_problem(*D.streets_1723())
    if name == "venice1778": return bal_problem(*D.venice_1778())
    if name == "dubrovnik16": return bal_problem(*D.dubrovnik_16())
    if name == "sphere2500": return PB.sphere2500(dict(np.load(os.path.join(gold, "sphere2500.npz"))))
    if name == "w20000": return PB.pose2_graph(dict(np.load(os.path.join(gold, "pose2_w20000.npz"))))
    if name == "dubrovnik_3_7": return PB.dubrovnik_timesfm(dict(np.load(os.path.join(gold, "dubrovnik_3_7.npz"))))
    if name.startswith("smart:"): return PB.SMART[name.split(":", 1)[1]]()      # smart:<fixture name of tests/problems.py>
    if name.startswith("bal:"):   # bal:<cams>:<points>:<seed>
        _, nc, npt, seed = name.split(":")
        return bal_problem(*D.synthetic_bal(int(nc), int(npt), seed=int(seed)))
    if name.startswith("baldup:"):   # the same, with every 7th observation duplicated (a camera seeing a landmark twice)
        _, nc, npt, seed = name.split(":")
        pr, v0 = bal_problem(*D.synthetic_bal(int(nc), int(npt), seed=int(seed)))
        idx = np.arange(0, pr.n_sfm, 7)
        pr.sfm_cam = np.concatenate([pr.sfm_cam, pr.sfm_cam[idx]]); pr.sfm_point = np.concatenate([pr.sfm_point, pr.sfm_point[idx]])
        pr.sfm_noise = np.concatenate([pr.sfm_noise, pr.sfm_noise[idx]])
        pr.sfm_z = np.concatenate([pr.sfm_z, (pr.sfm_z.reshape(-1, 2)[idx] + 0.25).reshape(-1)])
        return pr, v0
    raise SystemExit(f"unknown workload {name}")


def child(args):
    """Runs inside the LD_PRELOAD=libhipstub.so process."""
    stub = ctypes.CDLL(STUB)
    stub.hipstub_h2d_record.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_ulonglong)]
    problem, _ = problem_for(args.workload)
    from gtsam_amd import lib as L
    out = {"workload": args.workload, "shards": args.shards, "runs": []}

    def lockstep(ptr, n, stream):
        """Stand-in all-reduce for shards analysed one after the other in this process: as if every shard contributed
        the same buffer (true for the layout check of gtg_upload_problem, the only exchange of the set-up path)."""
        buf = np.frombuffer((ctypes.c_double * n).from_address(ptr), dtype=np.float64)
        buf *= args.shards

    import numpy as np
    for shard in range(args.shards):
        best = None
        for rep in range(args.reps):
            stub.hipstub_reset()
            t = time.perf_counter()
            g = L.DeviceGraph(problem, shard=shard, n_shards=args.shards, allreduce=lockstep if args.shards > 1 else None)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
            recs = []
            n = ctypes.c_longlong(); h = ctypes.c_ulonglong()
            for i in range(stub.hipstub_h2d_count()):
                stub.hipstub_h2d_record(i, ctypes.byref(n), ctypes.byref(h))
                recs.append((n.value, h.value))
            info = {"reduced_dim": int(g.reduced_dim), "cholesky_gflop": g.cholesky_flops() / 1e9, "h2d_bytes": int(stub.hipstub_bytes_h2d()),
                    "structure_hash": g.structure_hash()}
            g.close()
        sig = hashlib.sha256(json.dumps(sorted(recs)).encode()).hexdigest()[:16]
        out["runs"].append({"shard": shard, "setup_ms_best": best * 1e3, "uploads": len(recs), "signature": sig, **info})
    print("HOSTPROFILE " + json.dumps(out), flush=True)


def gloo_child(args):
    """One rank of a world_size-N `gloo` job: the library's sharded upload with the REAL all-reduce callback of
    gtsam_amd.distributed (under the stub the "device" pointers it is handed are host pointers, which is what the gloo
    path of make_allreduce takes).  Exercises the N > 1 set-up path end to end on CPU: shard filter, whole-graph
    structure, layout-consistency exchange."""
    import torch.distributed as dist
    problem, values0 = problem_for(args.workload)
    from gtsam_amd import lib as L
    from gtsam_amd.distributed import make_allreduce
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.port))
    dist.init_process_group("gloo", rank=args.gloo_rank, world_size=args.world)
    try:
        n_shards = args.claim_shards or args.world
        try:
            if args.late_callback:   # callback registered after the upload: the layout check runs in front of the first exchange
                g = L.DeviceGraph(problem, shard=args.gloo_rank % n_shards, n_shards=n_shards)
                g.set_allreduce(make_allreduce()); g.set_values(values0); g.linearize()
            else:
                g = L.DeviceGraph(problem, shard=args.gloo_rank % n_shards, n_shards=n_shards, allreduce=make_allreduce())
            out = {"rank": args.gloo_rank, "ok": True, "structure_hash": g.structure_hash(), "reduced_dim": int(g.reduced_dim),
                   "cholesky_gflop": g.cholesky_flops() / 1e9}
            g.close()
        except L.GtsamAmdError as e:
            out = {"rank": args.gloo_rank, "ok": False, "error": str(e)}
        print("HOSTPROFILE " + json.dumps(out), flush=True)
    finally:
        dist.destroy_process_group()


def run_gloo(workload, world=2, claim_shards=0, timeout=300, late_callback=False):
    """Launch `world` ranks of gloo_child under the stub; returns their records (rank order)."""
    import socket
    build_stub()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env["LD_PRELOAD"] = STUB
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), workload, "--gloo-rank", str(r), "--world", str(world),
                               "--port", str(port), "--claim-shards", str(claim_shards)] + (["--late-callback"] if late_callback else []), env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    recs = []
    for pr in procs:
        so, se = pr.communicate(timeout=timeout)
        if pr.returncode != 0:
            raise RuntimeError(f"gloo rank failed:\n{so[-2000:]}\n{se[-4000:]}")
        recs += [json.loads(line[len("HOSTPROFILE "):]) for line in so.splitlines() if line.startswith("HOSTPROFILE ")]
    return sorted(recs, key=lambda r: r["rank"])


def run_snippet(code, timeout=300, env_extra=None):
    """Run a Python snippet in a child process under the stub (host code of the library only); the snippet prints one
    line `RESULT <json>`; returns the parsed object."""
    build_stub()
    env = dict(os.environ); env["LD_PRELOAD"] = STUB; env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"snippet failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
    for line in r.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[len("RESULT "):])
    raise RuntimeError("no RESULT line:\n" + r.stdout[-2000:])


def run(workload, shards=1, reps=3, env_extra=None, quiet=False):
    """Spawn the dry-run child; returns the parsed record."""
    build_stub()
    env = dict(os.environ)
    env["LD_PRELOAD"] = STUB
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.abspath(__file__), workload, "--shards", str(shards), "--reps", str(reps), "--child"],
                       env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"host profile child failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
    if not quiet:
        sys.stderr.write(r.stderr)
    for line in r.stdout.splitlines():
        if line.startswith("HOSTPROFILE "):
            return json.loads(line[len("HOSTPROFILE "):])
    raise RuntimeError("no HOSTPROFILE line:\n" + r.stdout[-2000:])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="ladybug1723")
    ap.add_argument("--shards", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--sig-only", action="store_true")
    ap.add_argument("--gloo-rank", type=int, default=-1)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, default=0)
    ap.add_argument("--claim-shards", type=int, default=0, help="n_shards passed to the library (default: the world size)")
    ap.add_argument("--late-callback", action="store_true")
    a = ap.parse_args()
    if a.gloo_rank >= 0:
        gloo_child(a)
    elif a.child:
        child(a)
    else:
        rec = run(a.workload, a.shards, a.reps, env_extra=None if a.sig_only else {"GTG_DEBUG_TIMING": "1", "HIPSTUB_NO_HASH": "1"}, quiet=a.sig_only)
        print(json.dumps(rec, indent=1))
