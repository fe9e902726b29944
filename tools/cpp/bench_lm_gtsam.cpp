// tools/cpp/bench_lm_gtsam.cpp -- the headline measurement of bench.py, taken where north_star puts the host: a C++ program against
// GTSAM's own API (NonlinearFactorGraph / Values / LevenbergMarquardtParams), with gtsam_amd::GpuLevenbergMarquardtOptimizer in the
// place of gtsam::LevenbergMarquardtOptimizer.  Protocol of the reference's benchmark of this path (timing/timeSFMBAL.cpp:33-55,
// timing/timeSFMBAL.h:64-95): GeneralSFMFactor<SfmCamera, Point3>, Unit(2) noise, no priors, SetCeresDefaults, points-first ordering.
//
//   bench_lm_gtsam <BALfile> --steps K --warmup W
//
// Timing protocol and JSON keys: bench_lm_common.h (shared with bench_lm_pose3.cpp).  Prints ONE JSON line.
#include "bench_lm_common.h"

#include <gtsam/geometry/Cal3Bundler.h>
#include <gtsam/geometry/PinholeCamera.h>
#include <gtsam/inference/Symbol.h>
#include <gtsam/sfm/SfmData.h>
#include <gtsam/slam/GeneralSFMFactor.h>

#include <cstring>

using namespace gtsam;
using symbol_shorthand::C;
using symbol_shorthand::P;
typedef PinholeCamera<Cal3Bundler> Camera;
typedef GeneralSFMFactor<Camera, Point3> SfmFactor;

int main(int argc, char* argv[]) {
  if (argc < 2) { std::printf("usage: bench_lm_gtsam <BALfile> [--steps K] [--warmup W]\n"); return 2; }
  int steps = 20, warmup = 5;
  for (int a = 2; a + 1 < argc; a++) {
    if (!std::strcmp(argv[a], "--steps")) steps = std::atoi(argv[a + 1]);
    if (!std::strcmp(argv[a], "--warmup")) warmup = std::atoi(argv[a + 1]);
  }
  const SfmData db = SfmData::FromBalFile(argv[1]);
  const SharedNoiseModel noise = noiseModel::Unit::Create(2);
  NonlinearFactorGraph graph;
  for (size_t j = 0; j < db.numberTracks(); j++)
    for (const SfmMeasurement& m : db.tracks[j].measurements) graph.emplace_shared<SfmFactor>(m.second, noise, C(m.first), P(j));
  Values initial;
  size_t i = 0, j = 0;
  for (const SfmCamera& camera : db.cameras) initial.insert(C(i++), camera);
  for (const SfmTrack& track : db.tracks) initial.insert(P(j++), track.p);
  LevenbergMarquardtParams params;
  LevenbergMarquardtParams::SetCeresDefaults(&params);
  Ordering ordering;
  for (size_t jj = 0; jj < db.numberTracks(); jj++) ordering.push_back(P(jj));
  for (size_t ii = 0; ii < db.numberCameras(); ii++) ordering.push_back(C(ii));
  params.setOrdering(ordering);

  const benchlm::Result r = benchlm::run(graph, initial, params, steps, warmup);
  if (r.itsPerRun == 0) { std::printf("{\"failed\": \"the optimisation made no iteration\"}\n"); return 1; }
  std::printf("{\"program\": \"tools/cpp/bench_lm_gtsam.cpp: GTSAM graph + gtsam_amd::GpuLevenbergMarquardtOptimizer::optimize(), C++ host end to end\", "
              "\"cameras\": %zu, \"points\": %zu, \"factors\": %zu, %s}\n",
              db.numberCameras(), db.numberTracks(), graph.size(), benchlm::json(r, steps, warmup).c_str());
  return r.ok(steps) ? 0 : 1;
}
