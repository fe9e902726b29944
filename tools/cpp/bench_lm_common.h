// tools/cpp/bench_lm_common.h -- the timing protocol the C++ bench programs share (bench_lm_gtsam.cpp: BAL graphs, timing/timeSFMBAL.cpp's
// protocol; bench_lm_pose3.cpp: pose graphs, examples/Pose3SLAMExample_g2o.cpp's protocol with LM).
//
// A step is one LM iteration (LevenbergMarquardtOptimizer::iterate: linearize once + lambda tries until a step is accepted) as optimize()
// runs it.  An optimisation converges after a handful of iterations, so K steps are K iterations spread over as many optimize() calls as
// it takes: optimizers are CONSTRUCTED before the timed region (graph extracted, tables and values resident in HBM), the last one is
// capped by maxIterations so that the timed region holds exactly K iterations.  What is timed is optimize() as a user calls it --
// including the refresh of the returned gtsam::Values at its end.  Warm-up: W iterations the same way.  Also measured: construction ->
// converged of the FIRST optimizer of the process (cold: code-object load, first device allocations) and of a later one (warm), and the
// HIP-event phase times of one more optimisation (outside the timed region).
#pragma once

#include <GpuLevenbergMarquardtOptimizer.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

namespace benchlm {

typedef std::chrono::high_resolution_clock Clock;
inline double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

struct Result {
  size_t done = 0, optimizeCalls = 0, itsPerRun = 0;
  int tries = 0, innerPerRun = 0;
  double elapsedMs = 0, e0 = 0, eFinal = 0, hostError = 0;
  double coldConstruct = 0, coldOptimize = 0, warmConstruct = 0, warmOptimize = 0, deviceMs = 0;
  std::vector<double> phase;          // HIP-event ms per phase (gtg_phase_name order), summed over one optimisation
  std::vector<long long> phaseCalls;  // launches of each phase in that optimisation
  double flopsBlock = 0, flopsTiles = 0; long long reducedDim = 0;   // the factorisation's flop counts (gtg_cholesky_flops*) and the reduced system's size
  std::vector<double> errorTrace;     // error after every outer iteration of the phase-timing optimisation (iterationHook)
  bool ok(int steps) const { return done == (size_t)steps && std::abs(hostError - eFinal) <= 1e-9 * std::abs(hostError); }
};

inline Result run(const gtsam::NonlinearFactorGraph& graph, const gtsam::Values& initial, const gtsam::LevenbergMarquardtParams& params,
                  int steps, int warmup) {
  using gtsam_amd::GpuLevenbergMarquardtOptimizer;
  Result r;
  // ---- cold: the first optimizer of the process, construction -> converged (the metric's time-to-converged)
  {
    const auto c0 = Clock::now();
    GpuLevenbergMarquardtOptimizer lm(graph, initial, params);
    const auto c1 = Clock::now();
    r.e0 = lm.error();
    const gtsam::Values& result = lm.optimize();   // (the reference optimize() returns: copying it into a Values of the caller's is the caller's 16 ms)
    const auto c2 = Clock::now();
    r.coldConstruct = ms(c0, c1); r.coldOptimize = ms(c1, c2);
    r.itsPerRun = lm.iterations(); r.innerPerRun = lm.getInnerIterations(); r.eFinal = lm.error();
    r.hostError = graph.error(result);   // the reference's own evaluation of the returned Values
  }
  if (r.itsPerRun == 0) return r;
  // ---- warm: the same once more (what a program that optimises one problem after the other pays)
  {
    const auto w0 = Clock::now();
    GpuLevenbergMarquardtOptimizer lm(graph, initial, params);
    const auto w1 = Clock::now();
    lm.optimize();
    r.warmConstruct = ms(w0, w1); r.warmOptimize = ms(w1, Clock::now());
  }
  // ---- warm-up and timed region: optimizers built up front, optimize() calls timed
  auto plan = [&](int n) {   // maxIterations of the optimize() calls that make n iterations
    std::vector<int> caps;
    while (n > 0) { const int k = std::min<int>(n, (int)r.itsPerRun); caps.push_back(k); n -= k; }
    return caps;
  };
  auto build = [&](const std::vector<int>& caps) {
    std::vector<std::unique_ptr<GpuLevenbergMarquardtOptimizer>> v;
    for (int k : caps) { gtsam::LevenbergMarquardtParams p = params; p.maxIterations = k; v.emplace_back(new GpuLevenbergMarquardtOptimizer(graph, initial, p)); }
    return v;
  };
  { auto w = build(plan(warmup)); for (auto& o : w) o->optimize(); }
  auto timed = build(plan(steps));
  (void)hipDeviceSynchronize();
  const auto t0 = Clock::now();
  for (auto& o : timed) o->optimize();
  (void)hipDeviceSynchronize();
  const auto t1 = Clock::now();
  for (auto& o : timed) { r.done += o->iterations(); r.tries += o->getInnerIterations(); }
  r.optimizeCalls = timed.size();
  r.elapsedMs = ms(t0, t1);
  timed.clear();
  // device time of the phases of one more optimisation (events around every phase: not part of the timed region); this run also
  // records the error after every outer iteration (the hook makes optimize() refresh the host Values per iteration: not timed either)
  {
    gtsam::LevenbergMarquardtParams p = params;
    p.iterationHook = [&r](size_t, double, double after) { r.errorTrace.push_back(after); };
    GpuLevenbergMarquardtOptimizer lm(graph, initial, p);
    lm.enablePhaseTiming(true);
    lm.optimize();
    r.phase = lm.phaseMilliseconds(); r.phaseCalls = lm.phaseCalls();
    for (double v : r.phase) r.deviceMs += v;
    r.flopsBlock = gtg_cholesky_flops_block_level(lm.handle()); r.flopsTiles = gtg_cholesky_flops(lm.handle());
    r.reducedDim = (long long)gtg_reduced_dim(lm.handle());
  }
  return r;
}

// the keys both programs print (no braces: the caller adds its own keys around them)
inline std::string json(const Result& r, int steps, int warmup) {
  char buf[2048];
  std::snprintf(buf, sizeof buf,
                "\"steps\": %zu, \"steps_requested\": %d, \"warmup\": %d, \"elapsed_ms\": %.4f, "
                "\"ms_per_step\": %.5f, \"iterations_per_s\": %.4f, \"lambda_tries\": %d, \"lambda_tries_per_s\": %.4f, \"optimize_calls\": %zu, "
                "\"iterations_per_optimisation\": %zu, \"inner_iterations_per_optimisation\": %d, \"initial_error\": %.12g, \"final_error\": %.12g, "
                "\"final_error_recomputed_by_gtsam_on_host\": %.12g, \"cold_construct_ms\": %.2f, \"cold_optimize_ms\": %.2f, \"cold_time_to_converged_s\": %.5f, "
                "\"warm_construct_ms\": %.2f, \"warm_optimize_ms\": %.2f, \"warm_time_to_converged_s\": %.5f, \"device_phase_ms_one_optimisation\": %.3f",
                r.done, steps, warmup, r.elapsedMs, r.elapsedMs / std::max<size_t>(r.done, 1), 1e3 * r.done / std::max(r.elapsedMs, 1e-9), r.tries,
                1e3 * r.tries / std::max(r.elapsedMs, 1e-9), r.optimizeCalls, r.itsPerRun, r.innerPerRun, r.e0, r.eFinal, r.hostError, r.coldConstruct,
                r.coldOptimize, (r.coldConstruct + r.coldOptimize) * 1e-3, r.warmConstruct, r.warmOptimize, (r.warmConstruct + r.warmOptimize) * 1e-3, r.deviceMs);
  std::string s(buf);
  s += ", \"device_phase_ms\": {";
  for (size_t k = 0; k < r.phase.size(); k++) {
    std::snprintf(buf, sizeof buf, "%s\"%s\": %.4f", k ? ", " : "", gtg_phase_name((int)k), r.phase[k]);
    s += buf;
  }
  s += "}, \"device_phase_calls\": {";
  for (size_t k = 0; k < r.phaseCalls.size(); k++) {
    std::snprintf(buf, sizeof buf, "%s\"%s\": %lld", k ? ", " : "", gtg_phase_name((int)k), r.phaseCalls[k]);
    s += buf;
  }
  std::snprintf(buf, sizeof buf, "}, \"cholesky_flops_block_level\": %.6g, \"cholesky_flops_stored_tiles\": %.6g, \"reduced_dim\": %lld, \"error_trace\": [",
                r.flopsBlock, r.flopsTiles, r.reducedDim);
  s += buf;
  for (size_t k = 0; k < r.errorTrace.size(); k++) { std::snprintf(buf, sizeof buf, "%s%.12g", k ? ", " : "", r.errorTrace[k]); s += buf; }
  s += "]";
  return s;
}

}  // namespace benchlm
