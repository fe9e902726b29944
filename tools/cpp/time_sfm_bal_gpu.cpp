// tools/cpp/time_sfm_bal_gpu.cpp -- the reference's own benchmark of this path, timing/timeSFMBAL.cpp + timeSFMBAL.h:64-95, with one
// type name changed: the graph is built by GTSAM from a BAL file (SfmData::FromBalFile, GeneralSFMFactor<SfmCamera, Point3>, Unit(2)
// noise, no priors, SetCeresDefaults, points-first Schur ordering) and optimised by gtsam_amd::GpuLevenbergMarquardtOptimizer.
// Host C++ end to end -- this is what a GTSAM user's program pays: walking the NonlinearFactorGraph once (extraction into the SoA
// tables), the host symbolic analysis + uploads, then optimize() on the device.  Prints ONE JSON line.
//
//   time_sfm_bal_gpu <BALfile> [--cpu-iterations N]      (N > 0: also time N iterate() calls of gtsam::LevenbergMarquardtOptimizer)
#include <GpuLevenbergMarquardtOptimizer.h>
#include <gtsam/geometry/Cal3Bundler.h>
#include <gtsam/geometry/PinholeCamera.h>
#include <gtsam/inference/Symbol.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <gtsam/sfm/SfmData.h>
#include <gtsam/slam/GeneralSFMFactor.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

using namespace gtsam;
using symbol_shorthand::C;
using symbol_shorthand::P;
typedef PinholeCamera<Cal3Bundler> Camera;
typedef GeneralSFMFactor<Camera, Point3> SfmFactor;

static double ms(std::chrono::high_resolution_clock::time_point a, std::chrono::high_resolution_clock::time_point b) {
  return std::chrono::duration<double, std::milli>(b - a).count();
}

int main(int argc, char* argv[]) {
  if (argc < 2) { std::printf("usage: time_sfm_bal_gpu <BALfile> [--cpu-iterations N]\n"); return 2; }
  int cpuIterations = 0;
  for (int a = 2; a + 1 < argc; a++) if (!std::strcmp(argv[a], "--cpu-iterations")) cpuIterations = std::atoi(argv[a + 1]);
  const auto t0 = std::chrono::high_resolution_clock::now();
  const SfmData db = SfmData::FromBalFile(argv[1]);
  const auto t1 = std::chrono::high_resolution_clock::now();
  // Build graph using conventional GeneralSFMFactor (timing/timeSFMBAL.cpp:33-55)
  const SharedNoiseModel noise = noiseModel::Unit::Create(2);
  NonlinearFactorGraph graph;
  for (size_t j = 0; j < db.numberTracks(); j++)
    for (const SfmMeasurement& m : db.tracks[j].measurements) graph.emplace_shared<SfmFactor>(m.second, noise, C(m.first), P(j));
  Values initial;
  size_t i = 0, j = 0;
  for (const SfmCamera& camera : db.cameras) initial.insert(C(i++), camera);
  for (const SfmTrack& track : db.tracks) initial.insert(P(j++), track.p);
  // timing/timeSFMBAL.h:64-95
  LevenbergMarquardtParams params;
  LevenbergMarquardtParams::SetCeresDefaults(&params);
  Ordering ordering;
  for (size_t jj = 0; jj < db.numberTracks(); jj++) ordering.push_back(P(jj));
  for (size_t ii = 0; ii < db.numberCameras(); ii++) ordering.push_back(C(ii));
  params.setOrdering(ordering);
  const auto t2 = std::chrono::high_resolution_clock::now();
  // (the first optimizer lives in its own scope, as in a program that optimises one problem after the other: its device buffers are
  // released before the second one is constructed)
  std::chrono::high_resolution_clock::time_point t3, t4;
  double e0, lmError, hostError; size_t lmIterations; int lmInner;
  {
    gtsam_amd::GpuLevenbergMarquardtOptimizer lm(graph, initial, params);   // extraction, analysis, upload, initial error (device)
    t3 = std::chrono::high_resolution_clock::now();
    e0 = lm.error();
    const Values result = lm.optimize();
    t4 = std::chrono::high_resolution_clock::now();
    hostError = graph.error(result);
    lmError = lm.error(); lmIterations = lm.iterations(); lmInner = lm.getInnerIterations();
  }
  // the same once more in this process: the first run above also paid for the first use of the device (code object load, stream
  // and event creation, first-touch of the host scratch, device memory from the driver) -- a program that optimises more than once
  // pays the second figure
  const auto w0 = std::chrono::high_resolution_clock::now();
  gtsam_amd::GpuLevenbergMarquardtOptimizer lm2(graph, initial, params);
  const auto w1 = std::chrono::high_resolution_clock::now();
  lm2.enablePhaseTiming(true);
  lm2.optimize();
  const auto w2 = std::chrono::high_resolution_clock::now();
  double deviceMs = 0.0;
  for (double v : lm2.phaseMilliseconds()) deviceMs += v;
  double cpuMsPerIteration = 0.0; double cpuError = 0.0;
  if (cpuIterations > 0) {
    LevenbergMarquardtOptimizer ref(graph, initial, params);
    const auto c0 = std::chrono::high_resolution_clock::now();
    for (int k = 0; k < cpuIterations; k++) ref.iterate();
    cpuMsPerIteration = ms(c0, std::chrono::high_resolution_clock::now()) / cpuIterations;
    cpuError = ref.error();
  }
  std::printf("{\"program\": \"timeSFMBAL through GpuLevenbergMarquardtOptimizer (C++ host end to end)\", \"cameras\": %zu, \"points\": %zu, "
              "\"factors\": %zu, \"read_bal_ms\": %.1f, \"build_graph_ms\": %.1f, \"construct_ms\": %.1f, \"optimize_ms\": %.1f, "
              "\"iterations\": %zu, \"inner_iterations\": %d, \"ms_per_iteration\": %.3f, \"iterations_per_s_optimize_only\": %.2f, "
              "\"iterations_per_s_with_construction\": %.2f, \"initial_error\": %.9g, \"final_error\": %.12g, \"final_error_recomputed_on_host\": %.12g, "
              "\"second_run_construct_ms\": %.1f, \"second_run_optimize_ms\": %.1f, \"second_run_iterations_per_s_optimize_only\": %.2f, "
              "\"second_run_iterations_per_s_with_construction\": %.2f, \"second_run_device_phase_ms\": %.1f, \"cpu_reference_ms_per_iteration\": %.1f, \"cpu_reference_iterations\": %d, \"cpu_reference_error_after\": %.9g}\n",
              db.numberCameras(), db.numberTracks(), graph.size(), ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), lmIterations,
              lmInner, ms(t3, t4) / std::max<size_t>(lmIterations, 1), 1e3 * lmIterations / ms(t3, t4),
              1e3 * lmIterations / ms(t2, t4), e0, lmError, hostError, ms(w0, w1), ms(w1, w2), 1e3 * lm2.iterations() / ms(w1, w2),
              1e3 * lm2.iterations() / ms(w0, w2), deviceMs, cpuMsPerIteration, cpuIterations, cpuError);
  return std::abs(hostError - lmError) <= 1e-9 * std::abs(hostError) ? 0 : 1;
}
