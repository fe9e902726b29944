#!/usr/bin/env python3
"""bench.py -- LM iterations/sec on the BAL Ladybug-1723 shape (BASELINE.json metric), N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one LevenbergMarquardtOptimizer::iterate() (linearize once + lambda tries until a step is
accepted, nonlinear/LevenbergMarquardtOptimizer.cpp:273-308) with the reference's own benchmark protocol
for this path (timing/timeSFMBAL.h:64-95: GeneralSFMFactor<SfmCamera,Point3>, Unit(2) noise, no priors,
SetCeresDefaults, points-first Schur ordering).  Data are synthetic (the BAL file is not in the image):
gtsam_amd/datasets.py::ladybug_1723, seed 42, values resident in HBM before the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (= vector) peak; not in the local guide, AMD public figure
HBM_PEAK_GBS = 8000.0


def build_workload(name):
    from gtsam_amd import datasets as D
    from gtsam_amd.problem import bal_problem
    if name == "ladybug1723":
        return bal_problem(*D.ladybug_1723()), "BAL Ladybug problem-1723-156502 shape (synthetic, seed 42)"
    if name == "streets1723":
        return bal_problem(*D.streets_1723()), "BAL shape of the L1723 size on a street-network drive with random long-range loop closures (synthetic, seed 42; generator sensitivity, SURVEY section 7 hard part 8)"
    if name == "venice1778":
        return bal_problem(*D.venice_1778()), "BAL Venice problem-1778-993923 shape (synthetic, seed 42)"
    if name == "dubrovnik16":
        return bal_problem(*D.dubrovnik_16()), "BAL Dubrovnik-16-22106 shape (synthetic, seed 42)"
    if name == "sphere2500":
        import numpy as _np
        from tests import problems as PB
        g = dict(_np.load(os.path.join(ROOT, "tests", "golden", "sphere2500.npz")))
        return PB.sphere2500(g), "sphere2500 pose graph (reference's examples/Data/sphere2500.txt via the golden fixture), prior on pose 0, odometry-chain init"
    if name == "w20000":
        import numpy as _np
        from tests import problems as PB
        g = dict(_np.load(os.path.join(ROOT, "tests", "golden", "pose2_w20000.npz")))
        return PB.pose2_graph(g), "w20000 Pose2 pose graph (reference's examples/Data/w20000.txt via the golden fixture; BASELINE configs[0] with the absent w10000 replaced), prior on pose 0, load2D init"
    raise SystemExit(f"unknown workload {name}")


_PMC_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import bench
from gtsam_amd.optimizer import DeviceLevenbergMarquardt
from gtsam_amd.params import LevenbergMarquardtParams
(problem, values0), _ = bench.build_workload(%(workload)r)
prm = LevenbergMarquardtParams.CeresDefaults()
opt = DeviceLevenbergMarquardt(problem, values0, prm)
opt.dev.linearize()
for _ in range(2):
    opt.dev.try_lambda(prm.lambdaInitial, prm.diagonalDamping, prm.minDiagonal, prm.maxDiagonal)
opt.dev.close()
print("PMC_CHILD_DONE")
"""


def measure_traffic(workload):
    """HBM bytes per factorisation of the dominant kernel, measured in THIS run: two rocprofv3 --pmc sub-runs (FETCH_SIZE, WRITE_SIZE:
    separate passes, as MI355X_MICROARCH.md prescribes) of a child that performs two lambda tries of the bench workload with the
    dataflow factorisation in its single-kernel form (GTG_DF_SINGLE=1: counter collection serialises kernels, under which the two
    cooperating kernels of the production form cannot run; same tasks, same tiles read and written).  Counters are KB; FETCH_SIZE is
    doubled on gfx950 (the guide's correction).  Returns (bytes or None, source string, per-kernel dict)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found", {}
    sums = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="gtg_pmc_", dir="/tmp")
        env = dict(os.environ, GTG_DF_SINGLE="1", TMPDIR="/tmp", GTG_QUIET="1")
        try:
            r = subprocess.run([exe, "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, "-c", _PMC_CHILD % {"root": ROOT, "workload": workload}],
                               env=env, cwd="/tmp", capture_output=True, text=True, timeout=420)
            if "PMC_CHILD_DONE" not in r.stdout:       # (rocprofv3 may crash in its own finalisation after the database is written: not the criterion)
                return None, "pmc child failed: " + (r.stderr or r.stdout)[-200:], {}
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if not dbs:
                return None, "rocprofv3 wrote no database", {}
            con = sqlite3.connect(dbs[0])
            # (the view tools/rocprof_pmc.py reads: one row per dispatch and counter)
            rows = con.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
            con.close()
            sums[counter] = {name: (int(n), float(v)) for name, n, v in rows}
        except Exception as e:  # noqa: BLE001
            return None, f"pmc pass {counter} failed: {e}"[:300], {}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    per = {}
    for name, (n, v) in sums["FETCH_SIZE"].items():
        w = sums["WRITE_SIZE"].get(name, (n, 0.0))
        per[name.replace("(anonymous namespace)::", "").split("(")[0]] = {"launches": n, "read": 2.0 * v * 1024.0 / max(n, 1), "written": w[1] * 1024.0 / max(w[0], 1)}
    chol = [v for k, v in per.items() if "k_df_single" in k]
    if not chol:
        return None, "no k_df_single dispatch in the pmc passes", per
    return chol[0]["read"] + chol[0]["written"], ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE sub-runs (separate passes) of two lambda tries, "
                                                  "dataflow factorisation in its single-kernel form (GTG_DF_SINGLE=1), FETCH_SIZE x 2 (gfx950), KB units"), per


BAL_GENERATORS = {"ladybug1723": "ladybug_1723", "dubrovnik16": "dubrovnik_16", "venice1778": "venice_1778", "streets1723": "streets_1723"}
POSE_FIXTURES = {"sphere2500": ("sphere2500.npz", True), "w20000": ("pose2_w20000.npz", False)}


def write_workload_file(workload, path):
    """The input file of the C++ bench programs: a BAL file of the synthetic shape (gtsam_amd/datasets.py), or a g2o file made of the
    golden fixture's edges, noise models and initial poses at full precision (read back by GTSAM's own readG2o)."""
    from gtsam_amd import datasets as D
    from gtsam_amd import io as IO
    if workload in BAL_GENERATORS:
        IO.write_bal(path, *getattr(D, BAL_GENERATORS[workload])())
        return
    fixture, is3d = POSE_FIXTURES[workload]
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", fixture)))
    n = int(max(g["v1"].max(), g["v2"].max())) + 1
    d = dict(v1=g["v1"], v2=g["v2"], z=g["z"], noise_kind=g["noise_kind"], noise=g["noise"])
    IO.write_g2o(path, d, vertex_keys=np.arange(n), vertex_poses=np.asarray(g["values0"]).reshape(n, -1), full_precision=True)


def start_workload_file(workload):
    """Write the workload's input file in a child process (CPU only: the Venice shape takes the generator the better part of a minute),
    so that it is ready when its leg comes up.  -> (Popen, path)"""
    import subprocess
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"gtsam_amd_bench_{workload}_{os.getpid()}.txt")
    code = "import sys; sys.path.insert(0, %r); import bench; bench.write_workload_file(%r, %r)" % (ROOT, workload, path)
    return subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True), path


def cpp_host_leg(workload, steps, warmup, prepared=None, cpu_baseline=True):
    """The headline as north_star defines the host: a C++ program against GTSAM's own API -- GTSAM's loader, NonlinearFactorGraph / Values /
    LevenbergMarquardtParams, gtsam_amd::GpuLevenbergMarquardtOptimizer::optimize() through the C ABI; exactly `steps` LM iterations timed.
    BAL shapes: tools/cpp/bench_lm_gtsam.cpp (timing/timeSFMBAL.cpp's protocol); pose graphs: tools/cpp/bench_lm_pose3.cpp
    (examples/Pose3SLAMExample_g2o.cpp's protocol with LM, the reference's own optimizer timed beside it in the same process).
    `prepared` = (Popen, path) of start_workload_file."""
    import subprocess
    import tempfile
    pose = workload in POSE_FIXTURES
    if not pose and workload not in BAL_GENERATORS:
        return {"failed": "no C++ bench program for this workload"}
    exe = os.path.join(ROOT, "tests", "_build", "bench_lm_pose3" if pose else "bench_lm_gtsam")
    if not os.path.exists(exe):
        return {"failed": "tests/_build/%s not built (make -C gtsam_amd/host; needs GTSAM's headers = /root/reference)" % os.path.basename(exe)}
    path = prepared[1] if prepared else os.path.join(tempfile.gettempdir(), f"gtsam_amd_bench_{workload}_{os.getpid()}.txt")
    try:
        if prepared:
            _, err = prepared[0].communicate(timeout=600)
            if prepared[0].returncode != 0:
                return {"failed": "writing the input file failed: " + (err or "")[-300:]}
        else:
            write_workload_file(workload, path)
        cmd = [exe, path, "--steps", str(steps), "--warmup", str(warmup)]
        if pose:
            cmd += ["--cpu-baseline", "1" if cpu_baseline else "0"] + ([] if POSE_FIXTURES[workload][1] else ["--pose2"])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not line:
            return {"failed": ("no output; stderr: " + r.stderr[-300:])}
        j = json.loads(line[-1]); j["exit_code"] = r.returncode
        if r.returncode != 0 or j.get("steps") != steps:
            j["failed"] = "exit code %d, %s steps" % (r.returncode, j.get("steps"))
        return j
    except Exception as e:  # noqa: BLE001
        return {"failed": str(e)[:300]}
    finally:
        if os.path.exists(path):
            os.unlink(path)


def workload_record(workload, desc, cpp):
    """One entry of the line's `workloads`: the C++ host's figures of another workload, with the roofline of ITS dominant kernel (the
    factorisation: block-level flops x launches / HIP-event time of the Cholesky phase of one optimisation) and the reference beside it."""
    if "failed" in cpp and "iterations_per_s" not in cpp:
        return {"workload": desc, "failed": cpp["failed"]}
    ms = cpp.get("device_phase_ms", {}).get("cholesky", 0.0); calls = cpp.get("device_phase_calls", {}).get("cholesky", 0)
    fb, ft = cpp.get("cholesky_flops_block_level", 0.0), cpp.get("cholesky_flops_stored_tiles", 0.0)
    ach = fb * calls / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    ach_t = ft * calls / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    rec = {"workload": desc, "value": cpp["iterations_per_s"], "unit": "iterations/s", "ms_per_step": cpp["ms_per_step"], "steps": cpp["steps"],
           "lambda_tries_per_s": cpp["lambda_tries_per_s"], "lambda_tries": cpp["lambda_tries"],
           "time_to_converged_cold_s": cpp["cold_time_to_converged_s"], "time_to_converged_warm_s": cpp["warm_time_to_converged_s"],
           "converged_error": cpp["final_error"], "converged_iterations": cpp["iterations_per_optimisation"],
           "converged_inner_iterations": cpp["inner_iterations_per_optimisation"], "reduced_dim": cpp.get("reduced_dim"),
           "device_phase_ms_per_try": {k: v / max(cpp["inner_iterations_per_optimisation"], 1) for k, v in cpp.get("device_phase_ms", {}).items()},
           "roofline": {"bound": "mfma", "kernel": "tile-sparse FP64 Cholesky of the reduced system (k_df_bulk + k_df_chain), flops counted on the variable blocks",
                        "achieved": ach, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MATRIX_PEAK_TFLOPS, "traffic": None,
                        "flops_per_launch": fb, "flops_stored_tiles": ft, "ms_per_launch": ms / max(calls, 1), "launches": calls,
                        "frac_stored_tiles": ach_t / FP64_MATRIX_PEAK_TFLOPS},
           "cpu_baseline": cpp.get("cpu_baseline"), "trajectory_matches_reference": cpp.get("trajectory_matches_reference"),
           "value_source": cpp.get("program"), "cpp_host": cpp}
    if "failed" in cpp:
        rec["failed"] = cpp["failed"]
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ladybug1723")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--skip-dense-roofline", action="store_true", help="do not run the extra dense-schedule factorisations (used for clean profiles)")
    ap.add_argument("--traffic", default="auto", choices=["auto", "off"], help="auto: measure roofline.traffic in this run (two rocprofv3 --pmc sub-runs)")
    ap.add_argument("--parallelism", default="shard", choices=["shard", "speculative"],
                    help="N > 1: shard (default) = landmarks sharded, one RCCL all-reduce of the reduced camera system [S | g] per try (north_star, SURVEY 8(e)); "
                         "speculative = replicated handles, replica r tries the r-th lambda of the rejection sequence (gtsam_amd/speculative.py)")
    ap.add_argument("--extra-modes", default="auto", choices=["auto", "off"],
                    help="N > 1, auto: after the headline (shard) the same line also carries `extra_modes`: the speculative-lambda replicas, the "
                         "Venice-1778 shape sharded (BASELINE configs[5]) and the sharded PCG (implicit Schur complement) on the headline shape")
    ap.add_argument("--host", default="auto", choices=["auto", "python"], help="auto: the headline is timed in the C++ host (tools/cpp/bench_lm_gtsam.cpp) at N = 1")
    ap.add_argument("--workloads", default="auto",
                    help="N = 1, headline workload: the same line also carries `workloads` -- north_star's second headline (sphere2500, with the reference's own "
                         "optimizer as its cpu_baseline) and the Venice-1778 shape, each timed in its C++ host program.  auto = sphere2500,venice1778; off; or a comma list")
    args = ap.parse_args()
    extra_workloads = []
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.workload == "ladybug1723" and args.host == "auto" and args.workloads != "off":
        extra_workloads = ["sphere2500", "venice1778"] if args.workloads == "auto" else [w for w in args.workloads.split(",") if w]
    # (input files of the slow generators are written by child processes from the start, CPU only)
    prepared = {w: start_workload_file(w) for w in extra_workloads if w in ("venice1778",)}

    import torch
    from gtsam_amd.optimizer import DeviceLevenbergMarquardt, check_convergence
    from gtsam_amd.params import LevenbergMarquardtParams

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs the GPU"
    torch.cuda.set_device(local_rank)
    allreduce = None
    comm = None
    speculative = world > 1 and args.parallelism == "speculative"
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (a bounded collective time-out: a rank that died inside an exchange ends the job with an error instead of hanging it)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=300))
        from gtsam_amd.distributed import make_allreduce
        from gtsam_amd.speculative import TorchComm
        allreduce = make_allreduce()
        comm = TorchComm()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    (problem, values0), desc = build_workload(args.workload)
    params = LevenbergMarquardtParams.CeresDefaults()       # timing/timeSFMBAL.h:69-70
    if args.workload in ("sphere2500", "w20000"):
        params = LevenbergMarquardtParams()                  # Pose3SLAMExample_g2o protocol with legacy LM

    def fresh(problem_=None, values_=None, params_=None, mode=None):
        problem_ = problem if problem_ is None else problem_; values_ = values0 if values_ is None else values_
        params_ = params if params_ is None else params_
        mode = ("speculative" if speculative else "shard") if mode is None else mode
        if world > 1 and mode == "speculative":     # every rank holds the whole graph
            from gtsam_amd.speculative import SpeculativeLevenbergMarquardt
            return SpeculativeLevenbergMarquardt(problem_, values_, params_, device=local_rank, comm=comm)
        return DeviceLevenbergMarquardt(problem_, values_, params_, device=local_rank, shard=rank, n_shards=world,
                                        allreduce=allreduce if world > 1 else None)

    torch.cuda.synchronize()
    mem_free0 = torch.cuda.mem_get_info()[0]     # (the first handle of the process: nothing is in the library's block cache yet)
    opt = fresh()
    torch.cuda.synchronize()
    handle_bytes = mem_free0 - torch.cuda.mem_get_info()[0]      # device memory the first handle of the process takes from the driver ...
    from gtsam_amd.lib import load as _load_lib
    cached_bytes = int(_load_lib().gtg_cached_memory_bytes())   # ... of which the library keeps this much released set-up scratch for the next handle
    n_red = opt.dev.reduced_dim

    def run_iterations(o, k, values_=None, params_=None):
        """k calls of iterate(); when the run converges it restarts from the initial values (same work/iteration)."""
        values_ = values0 if values_ is None else values_; params_ = params if params_ is None else params_
        done = 0
        while done < k:
            before = o.error()
            o.iterate()
            done += 1
            if check_convergence(params_.relativeErrorTol, params_.absoluteErrorTol, params_.errorTol, before, o.error()) \
                    or o.iterations() >= params_.maxIterations:
                o.dev.set_values(values_)
                o._error = o.dev.error(); o._lambda = params_.lambdaInitial; o._factor = params_.lambdaFactor
                o._iterations = 0
        return done

    def max_over_ranks(x):
        if world == 1:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_mode(problem_, values_, params_, mode, steps, warmup):
        """One more measured configuration of an N > 1 run (same protocol: warm-up, barrier, `steps` iterations, barrier, max over ranks)."""
        o = fresh(problem_, values_, params_, mode)
        run_iterations(o, warmup, values_, params_)
        o.dev.enable_timing(True); o.dev.reset_timing()
        i0 = o.getInnerIterations()
        barrier()
        tt = time.perf_counter()
        run_iterations(o, steps, values_, params_)
        barrier()
        el = max_over_ranks(time.perf_counter() - tt)
        ph = o.dev.phase_ms()
        rec = {"value": steps / el, "unit": "iterations/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": warmup,
               "lambda_tries": int(o.getInnerIterations() - i0), "lambda_tries_per_s": (o.getInnerIterations() - i0) / el, "error_after": o.error(),
               "phase_ms_per_call": {k: (v[0] / v[1] if v[1] else 0.0) for k, v in ph.items()}}
        # the same roofline keys as the headline: the replicated factorisation of the direct solver (every rank factorises the summed system);
        # the iterative solver has no factorisation -- its time is the `solve` phase (HBM-bound products over the E slots)
        cms, ccalls = ph["cholesky"]
        if ccalls and cms > 0:
            fb, ft = o.dev.cholesky_flops_block_level(), o.dev.cholesky_flops()
            rec["roofline"] = {"bound": "mfma", "achieved": fb * ccalls / (cms * 1e-3) / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": fb * ccalls / (cms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS, "traffic": None, "flops_per_launch": fb,
                               "flops_stored_tiles": ft, "ms_per_launch": cms / ccalls, "frac_stored_tiles": ft * ccalls / (cms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS}
        else:
            rec["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                               "note": "no factorisation in this mode: see phase_ms_per_call.solve (PCG products, two passes over the E slots each)"}
        if mode == "speculative":
            rec["tries_computed_by_rank0"] = int(o.speculated); rec["tries_discarded_on_rank0"] = int(o.discarded)
        o.dev.close()
        return rec

    run_iterations(opt, args.warmup)
    opt.dev.enable_timing(True); opt.dev.reset_timing()
    inner0 = opt.getInnerIterations()
    barrier()
    t0 = time.perf_counter()
    run_iterations(opt, args.steps)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    tries = opt.getInnerIterations() - inner0
    phases = opt.dev.phase_ms()
    chol_ms, chol_calls = phases["cholesky"]
    chol_flops = opt.dev.cholesky_flops()
    chol_flops_executed = opt.dev.cholesky_flops_executed()   # after skipping structurally empty 16 x 16 operand sub-tiles (round 6)
    lin_ms, lin_calls = phases["linearize"]; asm_ms, _ = phases["assemble"]

    # MFMA kernel quality in isolation: the same factorisation with the tile schedule forced dense (no reordering,
    # every tile stored) -- the regime where the trailing update (k_syrk) dominates and the MFMA roofline applies.
    dense = None
    if world == 1 and not args.skip_dense_roofline:
        os.environ["GTG_ORDERING"] = "natural"; os.environ["GTG_DENSE_PLAN"] = "1"
        try:
            od = fresh()
            od.iterate()
            od.dev.enable_timing(True); od.dev.reset_timing()
            for _ in range(3):
                od.dev.linearize(); od.dev.try_lambda(1e-3, True)
            torch.cuda.synchronize()
            dms, dcalls = od.dev.phase_ms()["cholesky"]
            dense = {"flops_per_launch": od.dev.cholesky_flops(), "ms_per_launch": dms / max(dcalls, 1)}
            od.dev.close()
        finally:
            os.environ.pop("GTG_ORDERING", None); os.environ.pop("GTG_DENSE_PLAN", None)

    # time-to-converged-chi^2: one full optimize() from the initial values (construction -> checkConvergence), as a program pays
    # it that optimises one problem after the other: the handle of the timed loop is released first (the library keeps its big
    # device blocks for the next handle of the process -- where the driver clears memory as it hands it out, a fresh hipMalloc of
    # the 1.9 GB reduced system alone was measured at 74 ms)
    flops_block = opt.dev.cholesky_flops_block_level()
    lin_bytes = opt.dev.linearize_bytes()
    opt.dev.close()
    barrier()
    t1 = time.perf_counter()
    full = fresh()
    t_setup = time.perf_counter() - t1          # gtg_create + gtg_upload_problem (host symbolic analysis, table uploads) + initial error
    full.optimize()
    barrier()
    ttc = time.perf_counter() - t1
    full_rec = {"error": full.error(), "iterations": full.iterations(), "inner": full.getInnerIterations(), "initial_error": full.trace[0][1]}
    full.dev.close()

    # ---- N > 1: the other ways this loop can use N GPUs, measured in the same job and reported BESIDE the headline (never as `value`).
    # Every leg runs on all ranks (they contain collectives).  A leg that raises on ONE rank leaves the others inside the leg's own
    # RCCL collectives until the process group's time-out ends them, and a device collective of another size issued meanwhile would be
    # undefined behaviour -- so the ranks agree on a leg's outcome over a separate HOST (gloo) group, and after the first leg that
    # failed anywhere no further leg is started (the device communicator may be unusable): they are reported as skipped.
    extra = None
    if world > 1 and args.extra_modes == "auto":
        import datetime
        import torch.distributed as dist
        extra = {}
        esteps, ewarm = max(2, min(args.steps, 8)), max(1, min(args.warmup, 2))
        agree_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=900))

        broken = []     # a leg failed on some rank: no further legs, the headline line is printed all the same

        def leg(name, fn):
            if broken:
                extra[name] = {"failed": "skipped: " + broken[0]}
                return
            ok, rec, err = 1.0, None, None
            try:
                rec = fn()
            except Exception as e:  # noqa: BLE001
                ok, err = 0.0, f"{type(e).__name__}: {e}"[:300]
            try:
                t = torch.tensor([ok], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=agree_group)
                agreed = float(t.item()) == 1.0
            except Exception as e:  # noqa: BLE001
                agreed, err = False, err or f"the agreement over the host group after '{name}' raised {type(e).__name__}: {e}"[:300]
            if agreed:
                extra[name] = rec
            else:
                extra[name] = {"failed": err or "raised on another rank"}
                broken.append(f"the leg '{name}' failed on a rank ({extra[name]['failed'][:120]}); the device communicator is not trusted after that")

        other = "speculative" if not speculative else "shard"
        leg(other, lambda: dict(timed_mode(problem, values0, params, other, esteps, ewarm),
                                what=("replicas with a speculative lambda search (gtsam_amd/speculative.py): the sequential trajectory, one try per iteration"
                                      if other == "speculative" else "landmark shard + all-reduce of [S | g] per try")))
        if args.workload == "ladybug1723":
            def venice():
                (pv, vv), dv = build_workload("venice1778")
                return dict(timed_mode(pv, vv, LevenbergMarquardtParams.CeresDefaults(), "shard", esteps, ewarm), workload=dv,
                            what="BASELINE configs[5]: landmark shard + all-reduce of [S | g]; the shape on which the sharded landmark phases are as long as the factorisation")
            leg("venice1778_shard", venice)

        def pcg():
            pp = LevenbergMarquardtParams.CeresDefaults()
            pp.linearSolverType = "Iterative"
            return dict(timed_mode(problem, values0, pp, "shard", esteps, ewarm),
                        what="landmark shard + block-Jacobi PCG on the implicit Schur complement (reference defaults: epsilon_rel 1e-3): every product S p is "
                             "sharded, one all-reduce of the reduced vector per product -- no replicated factorisation")
        if problem.n_sfm:
            leg("pcg_shard", pcg)

    if rank == 0:
        # ALGORITHMIC flops of one factorisation = the elimination counted on the d x d variable blocks (sum over block columns of
        # f^3/3 + f^2 s + f s^2, fill included: what a supernodal code with the exact structure does); the kernels execute more
        # (whole 128x128 tiles, structural zeros inside them included): `achieved_stored_tiles` / `frac_stored_tiles`
        achieved = flops_block * chol_calls / (chol_ms * 1e-3) / 1e12 if chol_ms > 0 else 0.0
        achieved_tiles = chol_flops * chol_calls / (chol_ms * 1e-3) / 1e12 if chol_ms > 0 else 0.0
        # HBM bytes per factorisation: measured in this run by two rocprofv3 --pmc sub-runs (measure_traffic); when that is not possible
        # (no rocprofv3, --traffic off, N > 1) the figure of the committed PMC passes of the same command is quoted and labelled as such
        traffic, traffic_source, traffic_kernels = None, None, {}
        if args.traffic == "auto" and world == 1:
            traffic, traffic_source, traffic_kernels = measure_traffic(args.workload)
        if traffic is None:
            why = traffic_source
            try:
                pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_cholesky_traffic.json") and "dense" not in f)
                if pmc and args.workload == "ladybug1723" and world == 1:
                    traffic = json.load(open(os.path.join(ROOT, "profiles", pmc[-1]))).get("hbm_bytes_sparse")
                    traffic_source = "from_profiles: profiles/%s (separate rocprofv3 --pmc passes, tools/profile_round.sh), not measured in this run%s" % (pmc[-1], (" (" + why + ")") if why else "")
            except Exception:  # noqa: BLE001
                traffic = None
        # the headline: timed in the C++ host (north_star: "host stays C++ with GTSAM's ... API surface intact"); the Python mirror's
        # loop above stays in the line as `python_mirror` (it is also where the per-phase device times and the roofline come from)
        cpp = cpp_host_leg(args.workload, args.steps, args.warmup) if (world == 1 and args.host == "auto") else {"failed": "N > 1 or --host python: the sharded run is driven by torch.distributed from the Python mirror"}
        cpp_ok = "failed" not in cpp
        out = {
            "metric": "LM iterations/sec", "value": cpp["iterations_per_s"] if cpp_ok else args.steps / elapsed, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": cpp["ms_per_step"] if cpp_ok else 1e3 * elapsed / args.steps, "higher_is_better": True,
            "value_source": ("C++ host: tools/cpp/bench_lm_gtsam.cpp (GTSAM graph, Values and params; gtsam_amd::GpuLevenbergMarquardtOptimizer::optimize() through the C ABI; "
                             "exactly `steps` LM iterations over optimize() calls of optimizers constructed before the timed region)") if cpp_ok
                            else "Python mirror of the host (gtsam_amd/optimizer.py over the same C ABI): " + str(cpp.get("failed")),
            "python_mirror": {"value": args.steps / elapsed, "ms_per_step": 1e3 * elapsed / args.steps, "lambda_tries_per_s": tries / elapsed,
                              "note": "same C ABI driven from gtsam_amd/optimizer.py; a run that converges inside the timed loop restarts from the initial values (set_values + error), inside the timed region"},
            "cpp_host": cpp,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "protocol": ("timeSFMBAL: GeneralSFMFactor Unit(2) noise, no priors, Ceres LM params, Schur ordering" if problem.n_sfm
                                    else "Pose2SLAMExample_g2o with LevenbergMarquardt (legacy params), BetweenFactor<Pose2> + prior" if (problem.var_type == 3).any()
                                    else "Pose3SLAMExample_g2o with LevenbergMarquardt (legacy params), BetweenFactor<Pose3> + prior"),
                       "cameras": int((problem.var_type == 1).sum()), "points": int((problem.var_type == 2).sum()),
                       "poses": int(((problem.var_type == 0) | (problem.var_type == 3)).sum()), "between_factors": int(problem.n_between),
                       "observations": int(problem.n_sfm), "reduced_dim": int(n_red),
                       "parallelism": (f"speculative-lambda x{world} (replicas; replica r tries the r-th lambda of the rejection sequence, decisions replayed in order: the sequential trajectory)"
                                       if speculative else f"landmark-shard x{world}") if world > 1 else "single GPU",
                       "exchange": (("RCCL" if getattr(comm, "device_backend", False) else "gloo through the host (dry run)") + (": gather of 5 doubles + broadcast of the accepted values per iteration" if speculative else ": all-reduce of the reduced camera system [S | g] at block granularity + landmark part of delta per try, Hessian diagonal per linearisation")) if world > 1 else None},
            "extra_modes": extra,
            # the cross-checks a reader needs, at the top level: tries per iteration of the timed loop and the device time per iteration
            "lambda_tries": int(cpp["lambda_tries"]) if (cpp_ok and "lambda_tries" in cpp) else int(tries),
            "tries_per_iteration": (cpp["lambda_tries"] / args.steps) if (cpp_ok and "lambda_tries" in cpp) else tries / args.steps,
            "device_phase_ms_per_iteration": (cpp["device_phase_ms_one_optimisation"] / max(cpp["iterations_per_optimisation"], 1)) if (cpp_ok and cpp.get("device_phase_ms_one_optimisation"))
                                             else sum(v[0] for v in phases.values()) / args.steps,
            "device_phase_ms_source": "C++ host: HIP events around every phase of one more optimize() (not the timed region), divided by its iterations" if (cpp_ok and cpp.get("device_phase_ms_one_optimisation"))
                                      else "Python mirror: HIP events around every phase of the timed loop",
            "lambda_tries_per_s": cpp["lambda_tries_per_s"] if cpp_ok else tries / elapsed,
            # construction -> checkConvergence.  `time_to_converged_s` is the COLD figure when the C++ leg ran (first optimizer of a fresh
            # process: code-object load, first device allocations); warm = a later optimizer of the same process
            "time_to_converged_s": cpp["cold_time_to_converged_s"] if cpp_ok else ttc,
            "time_to_converged_warm_s": cpp["warm_time_to_converged_s"] if cpp_ok else ttc,
            "time_to_converged_python_mirror_warm_s": ttc, "time_to_converged_setup_s": t_setup, "device_memory_per_handle_bytes": int(handle_bytes - cached_bytes), "device_memory_cached_scratch_bytes": int(cached_bytes),
            "device_memory_note": "per handle = what the first handle took from the driver minus the released set-up scratch the library's block cache keeps for the next handle (GTG_ALLOC_CACHE_MB, default 2048; gtg_release_cached_memory() returns it)", "converged_error": full_rec["error"], "converged_iterations": full_rec["iterations"],
            "converged_inner_iterations": full_rec["inner"], "initial_error": full_rec["initial_error"],
            "phase_ms_per_call": {k: (v[0] / v[1] if v[1] else 0.0) for k, v in phases.items()},
            "roofline": {"bound": "mfma", "kernel": "tile-sparse FP64 Cholesky of the reduced camera system after RCM reordering: dataflow schedule, k_df_bulk + k_df_chain, one factorisation = one launch of each (flops_per_launch = the elimination counted at the granularity of the variable blocks, fill included = the algorithmic count `frac` is quoted on; flops_stored_tiles = the stored 128x128 tiles taken as full; flops_executed = what the kernels execute: the contraction products with a structurally empty 16x16 operand sub-tile are skipped)",
                         "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_other_kernels_bytes_per_launch": {k: v for k, v in traffic_kernels.items() if "k_df_single" not in k} or None,
                         "flops_per_launch": flops_block, "ms_per_launch": chol_ms / max(chol_calls, 1),
                         "flops_block_level": flops_block, "flops_stored_tiles": chol_flops, "flops_executed": chol_flops_executed,
                         "frac_executed": (chol_flops_executed * chol_calls / (chol_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS) if chol_ms > 0 else 0.0,
                         "achieved_stored_tiles": achieved_tiles, "frac_stored_tiles": achieved_tiles / FP64_MATRIX_PEAK_TFLOPS,
                         "flops_dense_n3_over_3": float(n_red) ** 3 / 3.0},
            "roofline_dense_kernel": None if dense is None else {
                "bound": "mfma", "kernel": "same Cholesky with the tile schedule forced dense (n^3/3 flops): k_syrk-dominated regime",
                "achieved": dense["flops_per_launch"] / (dense["ms_per_launch"] * 1e-3) / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": dense["flops_per_launch"] / (dense["ms_per_launch"] * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                "flops_per_launch": dense["flops_per_launch"], "ms_per_launch": dense["ms_per_launch"],
                "mfma_f64_microbench_ceiling_tflops": 70.2},
            # linearize -> Hessian blocks without stored Jacobians (fused.h): SURVEY 8(d)'s fused byte count (factor tables read, camera / landmark
            # blocks written, the off-diagonal blocks W = Jc^T Jp written once: here in their lambda-dependent form E by the point elimination of
            # the iteration's first try) over the device time of those three phases, one call each
            "roofline_linearize": (lambda t_ms: {"bound": "hbm", "achieved": lin_bytes / max(t_ms * 1e-3, 1e-12) / 1e9,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": lin_bytes / max(t_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
                                   "bytes_per_launch": lin_bytes, "ms_per_launch": t_ms,
                                   "phases": "linearize + assemble (camera-sorted and landmark-sorted recomputing passes) + one point elimination (writes the E blocks)",
                                   "note": "after the fusion these passes are gather- and latency-bound, not HBM-bound: they move 52 B per factor from HBM and recompute the record from the L2-resident camera table"})(
                                       lin_ms / max(lin_calls, 1) + asm_ms / max(lin_calls, 1) + phases["point_eliminate"][0] / max(phases["point_eliminate"][1], 1)),
        }
        # CPU baseline: the REAL reference (oracle/_ref = GTSAM built from /root/reference) on this host, one
        # LM iteration of the same problem made of the reference's own calls (1 thread: no TBB headers in the image)
        cpu = None
        if args.cpu_baseline != "off" and world == 1:
            try:
                from oracle import ref
                if ref.available():
                    g = ref.RefGraph(problem)
                    rc, ms, ref_res = g.iteration_phases(values0, params.lambdaInitial, params.diagonalDamping,
                                                         1 if problem.n_sfm else 0, with_results=True)
                    # parity of THIS run at THIS size: the device's first lambda try of the same problem against the numbers of
                    # the reference iteration just timed (what LM's accept / reject decision is made from)
                    pd = fresh()
                    e0 = pd.error()
                    pd.dev.linearize()
                    prc, pout = pd.dev.try_lambda(params.lambdaInitial, params.diagonalDamping, params.minDiagonal, params.maxDiagonal)
                    dn = float(np.linalg.norm(pd.dev.delta())); di = float(np.abs(pd.dev.delta()).max())
                    pd.dev.close()
                    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-300)   # noqa: E731
                    rho_ref = (ref_res[0] - ref_res[3]) / (ref_res[1] - ref_res[2]) if rc == 0 and ref_res[1] != ref_res[2] else float("nan")
                    rho_dev = (e0 - pout[2]) / (pout[0] - pout[1]) if prc == 0 and pout[0] != pout[1] else float("nan")
                    out["parity_vs_reference"] = {
                        "what": "first lambda try (lambda = %g, %s damping) of the same problem: this device run against the reference iteration of the cpu_baseline leg"
                                % (params.lambdaInitial, "diagonal" if params.diagonalDamping else "identity"),
                        "status_reference": int(rc), "status_device": int(prc),
                        "error_rel": rel(e0, ref_res[0]), "linear_error_at_delta_rel": rel(pout[1], ref_res[2]),
                        "trial_error_rel": rel(pout[2], ref_res[3]), "delta_norm2_rel": rel(dn, ref_res[4]), "delta_norminf_rel": rel(di, ref_res[5]),
                        "model_fidelity_reference": rho_ref, "model_fidelity_device": rho_dev,
                        "same_decision": bool((rho_ref > params.minModelFidelity) == (rho_dev > params.minModelFidelity)),
                        "tolerances": {"error": 1e-9, "delta": 1e-7, "trial_error": 1e-6},
                        "within_tolerance": bool(rc == prc == 0 and rel(e0, ref_res[0]) <= 1e-9 and rel(dn, ref_res[4]) <= 1e-7
                                                 and rel(di, ref_res[5]) <= 1e-7 and rel(pout[2], ref_res[3]) <= 1e-6)}
                    cpu = {"value": 1e3 / ms[7], "unit": "iterations/s", "cores": 1, "kind": "reference",
                           "sample": "1 LM iteration (linearize, hessianDiagonal, damp, eliminateMultifrontal+solve with the Schur ordering, "
                                     "2x linear error, retract, nonlinear error) of the SAME problem with gtsam built from /root/reference "
                                     "(-O3 -mavx2 -mfma, no TBB)",
                           "phase_ms": dict(zip(["linearize", "hessianDiagonal", "damp", "eliminate_solve", "linear_error_x2", "retract", "error", "total"],
                                                [float(x) for x in ms])),
                           "cores_used": 1, "host_cpus": os.cpu_count(),
                           "cores_note": "the reference build has no TBB (headers absent from the image): 1 thread is what gtsam itself uses here. ~95 % of the iteration is the "
                                         "elimination of the reduced camera system, a serial chain of dense fronts (eliminateMultifrontal's root clique), so more threads would not change "
                                         "the figure: see `assisted`, where linearize and the landmark eliminations are split over std::threads by the harness"}
                    # harness-ASSISTED multi-thread variant (the library is built without TBB -- no headers in the
                    # image -- so the split is made in the harness, oracle/ref_harness.cpp ref_graph_iteration_mt: linearize and the
                    # landmark eliminations of the Schur ordering on `threads` std::threads, the camera system on one).  It is NOT
                    # GTSAM's own figure and never `value`: it rides along as `assisted`
                    try:
                        nth = max(1, min(8, os.cpu_count() or 1))
                        rc_mt, ms_mt, res_mt = g.iteration_mt(values0, params.lambdaInitial, params.diagonalDamping, nth)
                        cpu["assisted"] = {
                            "threads": nth, "value": 1e3 / ms_mt[4], "unit": "iterations/s", "status": int(rc_mt),
                            "what": "same iteration, linearize + per-landmark-group eliminatePartialSequential split over std::threads in the "
                                    "harness (what TBB would run in parallel), remaining camera system eliminated on one thread",
                            "phase_ms": dict(zip(["linearize", "eliminate_landmarks", "eliminate_solve_cameras", "back_substitute", "total"],
                                                 [float(x) for x in ms_mt])),
                            "delta_norm2_rel_vs_1_thread": rel(res_mt[4], ref_res[4])}
                    except Exception as e:  # noqa: BLE001
                        cpu["assisted"] = {"value": None, "failed": str(e)}
                else:
                    from oracle import gtsam_oracle as O
                    from gtsam_amd import datasets as D
                    from gtsam_amd.problem import bal_problem
                    ps, vs = bal_problem(*D.synthetic_bal(12, 300, seed=1, n_loops=1))
                    tt = time.perf_counter(); O.solve_damped(ps, vs, 1e-4, True); dt = time.perf_counter() - tt
                    cpu = {"value": 1.0 / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
                           "sample": "numpy restatement on a 12-camera / 258-point sub-problem (oracle/_ref not present)"}
            except Exception as e:  # noqa: BLE001
                cpu = {"value": None, "unit": "iterations/s", "cores": 1, "kind": "reference", "sample": f"failed: {e}"}
        out["cpu_baseline"] = cpu
        # north_star's other workloads, measured the same way in the same run (C++ host, own roofline; sphere2500 with the reference's own
        # optimizer on the same graph in the same process as its cpu_baseline)
        if extra_workloads:
            out["workloads"] = {}
            for w in extra_workloads:
                try:
                    wdesc = {"sphere2500": "sphere2500 pose graph (the reference's examples/Data/sphere2500.txt through the golden fixture, written as g2o at full precision and read by GTSAM's readG2o): Pose3SLAMExample_g2o protocol with LevenbergMarquardt (legacy params), prior on pose 0, odometry-chain init",
                             "venice1778": "BAL Venice problem-1778-993923 shape (synthetic, seed 42): timeSFMBAL protocol"}.get(w, w)
                    wsteps = args.steps if w in POSE_FIXTURES else max(2, min(args.steps, 10))
                    cppw = cpp_host_leg(w, wsteps, min(args.warmup, 2), prepared=prepared.get(w), cpu_baseline=args.cpu_baseline != "off")
                    out["workloads"][w] = workload_record(w, wdesc, cppw)
                    if w == "venice1778" and "failed" not in out["workloads"][w]:
                        out["workloads"][w]["cpu_baseline"] = {"value": None, "kind": "reference", "sample": "not run: one reference iteration of this shape takes minutes on one thread (L1723, a seventh of the observations: 33 s)"}
                except Exception as e:  # noqa: BLE001
                    out["workloads"][w] = {"failed": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001  (a communicator an extra leg broke: the line is out, nothing left to do)
            pass


if __name__ == "__main__":
    main()
