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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ladybug1723")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--skip-dense-roofline", action="store_true", help="do not run the extra dense-schedule factorisations (used for clean profiles)")
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        from gtsam_amd.distributed import make_allreduce
        allreduce = make_allreduce()

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

    def fresh():
        return DeviceLevenbergMarquardt(problem, values0, params, device=local_rank, shard=rank, n_shards=world,
                                        allreduce=allreduce)

    opt = fresh()
    n_red = opt.dev.reduced_dim

    def run_iterations(o, k):
        """k calls of iterate(); when the run converges it restarts from the initial values (same work/iteration)."""
        done = 0
        while done < k:
            before = o.error()
            o.iterate()
            done += 1
            if check_convergence(params.relativeErrorTol, params.absoluteErrorTol, params.errorTol, before, o.error()) \
                    or o.iterations() >= params.maxIterations:
                o.dev.set_values(values0)
                o._error = o.dev.error(); o._lambda = params.lambdaInitial; o._factor = params.lambdaFactor
                o._iterations = 0
        return done

    run_iterations(opt, args.warmup)
    opt.dev.enable_timing(True); opt.dev.reset_timing()
    inner0 = opt.getInnerIterations()
    barrier()
    t0 = time.perf_counter()
    run_iterations(opt, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tries = opt.getInnerIterations() - inner0
    phases = opt.dev.phase_ms()
    chol_ms, chol_calls = phases["cholesky"]
    chol_flops = opt.dev.cholesky_flops()
    lin_ms, lin_calls = phases["linearize"]; asm_ms, _ = phases["assemble"]

    # MFMA kernel quality in isolation: the same factorisation with the tile schedule forced dense (no reordering,
    # every tile stored) -- the regime where the trailing update (k_syrk) dominates and the MFMA roofline applies.
    dense = None
    if world == 1 and not args.skip_dense_roofline:
        os.environ["GTG_NO_REORDER"] = "1"; os.environ["GTG_DENSE_PLAN"] = "1"
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
            os.environ.pop("GTG_NO_REORDER", None); os.environ.pop("GTG_DENSE_PLAN", None)

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

    if rank == 0:
        # ALGORITHMIC flops of one factorisation = the elimination counted on the d x d variable blocks (sum over block columns of
        # f^3/3 + f^2 s + f s^2, fill included: what a supernodal code with the exact structure does); the kernels execute more
        # (whole 128x128 tiles, structural zeros inside them included): `achieved_stored_tiles` / `frac_stored_tiles`
        achieved = flops_block * chol_calls / (chol_ms * 1e-3) / 1e12 if chol_ms > 0 else 0.0
        achieved_tiles = chol_flops * chol_calls / (chol_ms * 1e-3) / 1e12 if chol_ms > 0 else 0.0
        # HBM bytes per factorisation: NOT measured in this run (PMC collection serialises kernels and needs its own rocprofv3
        # passes); the figure is read from the committed PMC passes of this same command and labelled as such
        traffic, traffic_source = None, None
        try:
            pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_cholesky_traffic.json") and "dense" not in f)
            if pmc and args.workload == "ladybug1723" and world == 1:
                traffic = json.load(open(os.path.join(ROOT, "profiles", pmc[-1]))).get("hbm_bytes_sparse")
                traffic_source = "from_profiles: profiles/%s (separate rocprofv3 --pmc passes, tools/profile_round.sh), not measured in this run" % pmc[-1]
        except Exception:  # noqa: BLE001
            traffic = None
        out = {
            "metric": "LM iterations/sec", "value": args.steps / elapsed, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "protocol": ("timeSFMBAL: GeneralSFMFactor Unit(2) noise, no priors, Ceres LM params, Schur ordering" if problem.n_sfm
                                    else "Pose2SLAMExample_g2o with LevenbergMarquardt (legacy params), BetweenFactor<Pose2> + prior" if (problem.var_type == 3).any()
                                    else "Pose3SLAMExample_g2o with LevenbergMarquardt (legacy params), BetweenFactor<Pose3> + prior"),
                       "cameras": int((problem.var_type == 1).sum()), "points": int((problem.var_type == 2).sum()),
                       "poses": int(((problem.var_type == 0) | (problem.var_type == 3)).sum()), "between_factors": int(problem.n_between),
                       "observations": int(problem.n_sfm), "reduced_dim": int(n_red),
                       "parallelism": f"landmark-shard x{world}" if world > 1 else "single GPU"},
            "lambda_tries_per_s": tries / elapsed,
            "time_to_converged_s": ttc, "time_to_converged_setup_s": t_setup, "converged_error": full.error(), "converged_iterations": full.iterations(),
            "converged_inner_iterations": full.getInnerIterations(), "initial_error": full.trace[0][1],
            "phase_ms_per_call": {k: (v[0] / v[1] if v[1] else 0.0) for k, v in phases.items()},
            "roofline": {"bound": "mfma", "kernel": "tile-sparse FP64 Cholesky of the reduced camera system after RCM reordering: dataflow schedule, k_df_bulk + k_df_chain, one factorisation = one launch of each (flops_per_launch = the elimination counted at the granularity of the variable blocks, fill included = the algorithmic count `frac` is quoted on; flops_stored_tiles = what the kernels execute over the stored 128x128 tiles)",
                         "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "flops_per_launch": flops_block, "ms_per_launch": chol_ms / max(chol_calls, 1),
                         "flops_block_level": flops_block, "flops_stored_tiles": chol_flops,
                         "achieved_stored_tiles": achieved_tiles, "frac_stored_tiles": achieved_tiles / FP64_MATRIX_PEAK_TFLOPS,
                         "flops_dense_n3_over_3": float(n_red) ** 3 / 3.0},
            "roofline_dense_kernel": None if dense is None else {
                "bound": "mfma", "kernel": "same Cholesky with the tile schedule forced dense (n^3/3 flops): k_syrk-dominated regime",
                "achieved": dense["flops_per_launch"] / (dense["ms_per_launch"] * 1e-3) / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": dense["flops_per_launch"] / (dense["ms_per_launch"] * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                "flops_per_launch": dense["flops_per_launch"], "ms_per_launch": dense["ms_per_launch"],
                "mfma_f64_microbench_ceiling_tflops": 70.2},
            "roofline_linearize": {"bound": "hbm", "achieved": lin_bytes * lin_calls / max((lin_ms + asm_ms) * 1e-3, 1e-12) / 1e9,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": lin_bytes * lin_calls / max((lin_ms + asm_ms) * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
                                   "bytes_per_launch": lin_bytes},
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
                           "host_cpus": os.cpu_count()}
                    # multi-thread variant (the library is built without TBB -- no headers in the image -- so the split is made in
                    # the harness, oracle/ref_harness.cpp ref_graph_iteration_mt: linearize and the landmark eliminations of the
                    # Schur ordering on `threads` std::threads, the camera system on one); reported next to the 1-thread figure,
                    # whichever is faster is `value`
                    try:
                        nth = max(1, min(8, os.cpu_count() or 1))
                        rc_mt, ms_mt, res_mt = g.iteration_mt(values0, params.lambdaInitial, params.diagonalDamping, nth)
                        cpu["multi_thread"] = {
                            "threads": nth, "value": 1e3 / ms_mt[4], "unit": "iterations/s", "status": int(rc_mt),
                            "what": "same iteration, linearize + per-landmark-group eliminatePartialSequential split over std::threads in the "
                                    "harness (what TBB would run in parallel), remaining camera system eliminated on one thread",
                            "phase_ms": dict(zip(["linearize", "eliminate_landmarks", "eliminate_solve_cameras", "back_substitute", "total"],
                                                 [float(x) for x in ms_mt])),
                            "delta_norm2_rel_vs_1_thread": rel(res_mt[4], ref_res[4])}
                        if rc_mt == 0 and cpu["multi_thread"]["value"] > cpu["value"]:
                            cpu["value_1_thread"] = cpu["value"]; cpu["value"] = cpu["multi_thread"]["value"]; cpu["cores"] = nth
                    except Exception as e:  # noqa: BLE001
                        cpu["multi_thread"] = {"value": None, "failed": str(e)}
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
        # The C++ host north_star asks for, measured in this run: the reference's own benchmark program of the path
        # (timing/timeSFMBAL.cpp with the optimizer's type name changed, tools/cpp/time_sfm_bal_gpu.cpp) on the same problem
        # written as a BAL file -- GTSAM's loader, GTSAM's graph, GpuLevenbergMarquardtOptimizer (extraction + upload +
        # optimize() through the C ABI).  Second construction / optimisation of the process (the first also pays the first
        # use of the device); time_to_converged = construction -> checkConvergence, the metric's definition.
        out["cpp_host"] = None
        exe = os.path.join(ROOT, "tests", "_build", "time_sfm_bal_gpu")
        if world == 1 and args.workload in ("ladybug1723", "dubrovnik16", "venice1778", "streets1723") and os.path.exists(exe) and args.cpu_baseline != "off":
            try:
                import subprocess
                import tempfile
                from gtsam_amd import datasets as D
                from gtsam_amd import io as IO
                gen = {"ladybug1723": D.ladybug_1723, "dubrovnik16": D.dubrovnik_16, "venice1778": D.venice_1778, "streets1723": D.streets_1723}[args.workload]
                path = os.path.join(tempfile.gettempdir(), f"gtsam_amd_bench_{args.workload}.txt")
                IO.write_bal(path, *gen())
                r = subprocess.run([exe, path], capture_output=True, text=True, timeout=600)
                os.unlink(path)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
                j = json.loads(line)
                out["cpp_host"] = {
                    "program": "tools/cpp/time_sfm_bal_gpu.cpp (= timing/timeSFMBAL.cpp, optimizer type changed), GTSAM built from /root/reference",
                    "construct_ms": j["second_run_construct_ms"], "optimize_ms": j["second_run_optimize_ms"],
                    "time_to_converged_s": (j["second_run_construct_ms"] + j["second_run_optimize_ms"]) * 1e-3,
                    "iterations": j["iterations"], "inner_iterations": j["inner_iterations"],
                    "iterations_per_s_optimize_only": j["second_run_iterations_per_s_optimize_only"],
                    "device_phase_ms": j["second_run_device_phase_ms"],
                    "first_run_construct_ms": j["construct_ms"], "first_run_optimize_ms": j["optimize_ms"],
                    "final_error": j["final_error"], "final_error_recomputed_by_gtsam_on_host": j["final_error_recomputed_on_host"],
                    "exit_code": r.returncode}
            except Exception as e:  # noqa: BLE001
                out["cpp_host"] = {"failed": str(e)[:300]}
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
