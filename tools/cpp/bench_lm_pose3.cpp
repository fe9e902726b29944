// tools/cpp/bench_lm_pose3.cpp -- north_star's second headline (sphere2500) measured like the first: a C++ program against GTSAM's own API
// with gtsam_amd::GpuLevenbergMarquardtOptimizer in the place of gtsam::LevenbergMarquardtOptimizer, and -- in the same process, on
// the same graph -- gtsam::LevenbergMarquardtOptimizer itself as the CPU baseline.
// Protocol: examples/Pose3SLAMExample_g2o.cpp:36-56 (readG2o, prior Diagonal::Variances(1e-6 x3, 1e-4 x3) on the first key) with
// LevenbergMarquardtParams in the place of GaussNewtonParams (BASELINE.json configs[3]); `--pose2`: examples/Pose2SLAMExample_g2o.cpp
// :46-67 (load2D through readG2o, prior Variances(1e-6, 1e-6, 1e-8)).  A file without VERTEX lines (the reference's sphere2500.txt)
// is initialised along the odometry chain: every pose from its predecessor and the edge i -> i + 1.
//
//   bench_lm_pose3 <g2oFile> [--pose2] [--steps K] [--warmup W] [--cpu-baseline 0|1]
//
// Timing protocol and JSON keys: bench_lm_common.h.  Prints ONE JSON line.
#include "bench_lm_common.h"

#include <gtsam/geometry/Pose2.h>
#include <gtsam/geometry/Pose3.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/dataset.h>

#include <cstring>
#include <thread>

using namespace gtsam;

template <class POSE>
static void chainInit(const NonlinearFactorGraph& graph, Values* initial) {
  // poses the file does not list: the first key of the first edge at the origin, then key2 = key1 * measured wherever key1 is known
  // (edges in file order; the odometry edges i -> i + 1 come first in the TORO files)
  for (const auto& f : graph) {
    auto b = std::dynamic_pointer_cast<BetweenFactor<POSE>>(f);
    if (!b) continue;
    if (initial->empty()) initial->insert(b->key1(), POSE());
    if (initial->exists(b->key1()) && !initial->exists(b->key2())) initial->insert(b->key2(), initial->at<POSE>(b->key1()) * b->measured());
  }
}

int main(int argc, char* argv[]) {
  if (argc < 2) { std::printf("usage: bench_lm_pose3 <g2oFile> [--pose2] [--steps K] [--warmup W] [--cpu-baseline 0|1]\n"); return 2; }
  int steps = 20, warmup = 5, cpuBaseline = 1; bool is3D = true;
  for (int a = 2; a < argc; a++) {
    if (!std::strcmp(argv[a], "--pose2")) is3D = false;
    if (a + 1 < argc && !std::strcmp(argv[a], "--steps")) steps = std::atoi(argv[a + 1]);
    if (a + 1 < argc && !std::strcmp(argv[a], "--warmup")) warmup = std::atoi(argv[a + 1]);
    if (a + 1 < argc && !std::strcmp(argv[a], "--cpu-baseline")) cpuBaseline = std::atoi(argv[a + 1]);
  }
  NonlinearFactorGraph::shared_ptr graph;
  Values::shared_ptr initial;
  std::tie(graph, initial) = readG2o(argv[1], is3D);
  const size_t listed = initial->size();
  if (is3D) chainInit<Pose3>(*graph, initial.get()); else chainInit<Pose2>(*graph, initial.get());
  const size_t nBetween = graph->size();
  const Key firstKey = initial->keys().front();
  if (is3D) graph->addPrior(firstKey, Pose3(), noiseModel::Diagonal::Variances((Vector(6) << 1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4).finished()));
  else graph->addPrior(firstKey, Pose2(), noiseModel::Diagonal::Variances(Vector3(1e-6, 1e-6, 1e-8)));
  const LevenbergMarquardtParams params;   // the legacy defaults, as `LevenbergMarquardtOptimizer optimizer(graph, initial)` uses them

  const benchlm::Result r = benchlm::run(*graph, *initial, params, steps, warmup);
  if (r.itsPerRun == 0) { std::printf("{\"failed\": \"the optimisation made no iteration\"}\n"); return 1; }

  // ---- the CPU baseline: the reference's own optimizer on the same graph, same initial values, same params, this process, this host
  std::string cpu = "null";
  bool sameTrace = true;
  if (cpuBaseline) {
    std::vector<double> refTrace;
    LevenbergMarquardtParams p = params;
    p.iterationHook = [&refTrace](size_t, double, double after) { refTrace.push_back(after); };
    const auto t0 = benchlm::Clock::now();
    LevenbergMarquardtOptimizer lm(*graph, *initial, p);
    const auto t1 = benchlm::Clock::now();
    lm.optimize();
    const auto t2 = benchlm::Clock::now();
    double worst = 0.0;
    sameTrace = refTrace.size() == r.errorTrace.size();
    for (size_t k = 0; sameTrace && k < refTrace.size(); k++) worst = std::max(worst, std::abs(refTrace[k] - r.errorTrace[k]) / std::max(std::abs(refTrace[k]), 1e-300));
    char buf[1024];
    std::snprintf(buf, sizeof buf,
                  "{\"value\": %.5f, \"unit\": \"iterations/s\", \"cores\": 1, \"kind\": \"reference\", \"lambda_tries_per_s\": %.5f, \"construct_ms\": %.2f, \"optimize_ms\": %.2f, "
                  "\"time_to_converged_s\": %.5f, \"iterations\": %zu, \"inner_iterations\": %d, \"final_error\": %.12g, \"host_cpus\": %u, "
                  "\"same_outer_iterations_as_device\": %s, \"error_trace_max_rel_diff_vs_device\": %.3e, "
                  "\"sample\": \"the whole optimisation: gtsam::LevenbergMarquardtOptimizer (built from /root/reference, -O3 -mavx2 -mfma, no TBB: 1 thread) "
                  "constructed and optimize()d on the SAME graph / Values / params in this process; value = outer iterations / optimize() seconds\"}",
                  1e3 * lm.iterations() / benchlm::ms(t1, t2), 1e3 * lm.getInnerIterations() / benchlm::ms(t1, t2), benchlm::ms(t0, t1), benchlm::ms(t1, t2),
                  benchlm::ms(t0, t2) * 1e-3, lm.iterations(), lm.getInnerIterations(), lm.error(), std::thread::hardware_concurrency(),
                  sameTrace ? "true" : "false", sameTrace ? worst : -1.0);
    cpu = buf;
    sameTrace = sameTrace && worst <= 1e-6 && lm.iterations() == r.itsPerRun && lm.getInnerIterations() == r.innerPerRun;
  }
  std::printf("{\"program\": \"tools/cpp/bench_lm_pose3.cpp: readG2o + prior + gtsam_amd::GpuLevenbergMarquardtOptimizer::optimize(), C++ host end to end\", "
              "\"poses\": %zu, \"poses_listed_in_file\": %zu, \"between_factors\": %zu, \"factors\": %zu, %s, \"cpu_baseline\": %s, \"trajectory_matches_reference\": %s}\n",
              initial->size(), listed, nBetween, graph->size(), benchlm::json(r, steps, warmup).c_str(), cpu.c_str(),
              cpuBaseline ? (sameTrace ? "true" : "false") : "null");
  return r.ok(steps) ? 0 : 1;
}
