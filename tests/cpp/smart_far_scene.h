// tests/cpp/smart_far_scene.h -- TEST INFRASTRUCTURE.  A smart-factor graph on which the degeneracy modes that replace a failed
// track by a POINT AT INFINITY (SmartProjectionFactor.h:356-371, :419-427) make sense: cameras on a narrow arc (every viewing
// direction in front of every camera, or the reference throws a CheiralityException), a cloud wide enough across the optical axes
// for the distortion coefficients to be observable, and a quarter of the landmarks thousands of units behind it -- those fail the
// landmark-distance threshold (FAR_POINT); a few tracks keep a single measurement (DEGENERATE).  Priors on the end cameras.
#pragma once
#include <gtsam/geometry/Cal3Bundler.h>
#include <gtsam/geometry/PinholeCamera.h>
#include <gtsam/inference/Symbol.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>
#include <gtsam/slam/SmartProjectionFactor.h>

#include <random>

inline void smartFarScene(gtsam::LinearizationMode lin, gtsam::DegeneracyMode deg, gtsam::NonlinearFactorGraph* graph, gtsam::Values* initial) {
  using namespace gtsam;
  typedef PinholeCamera<Cal3Bundler> Camera;
  std::mt19937 rng(1234);
  std::normal_distribution<double> N(0.0, 1.0);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  const int nc = 9, np = 140, nfar = 24;
  std::vector<Camera> cams; std::vector<Point3> pts;
  for (int i = 0; i < nc; i++) {
    const double a = 0.125 * i - 0.5;
    cams.emplace_back(Pose3(Rot3::RzRyRx(0.02 * N(rng), -a, 0.02 * N(rng)), Point3(8 * std::sin(a), 0.3 * N(rng), -8 * std::cos(a))),
                      Cal3Bundler(500 + 30 * i, 2e-2 * N(rng), 2e-3 * N(rng), 0, 0));
  }
  for (int j = 0; j < np - nfar; j++) pts.emplace_back(4.0 * U(rng), 3.0 * U(rng), 2.0 * U(rng));
  for (int j = 0; j < nfar; j++) { const double d = 2750 + 1250 * U(rng); pts.emplace_back(0.15 * d * U(rng), 0.12 * d * U(rng), d); }
  SmartProjectionParams sp(lin, deg);
  sp.setLandmarkDistanceThreshold(100.0);
  auto noise = noiseModel::Isotropic::Sigma(2, 0.8);
  for (int j = 0; j < np; j++) {
    auto f = std::make_shared<SmartProjectionFactor<Camera>>(noise, sp);
    int used = 0;
    for (int i = 0; i < nc; i++) {
      if ((i + 3 * j) % 4 == 0 && used >= 2) continue;
      if (j % 11 == 5 && used >= 1) continue;                       // a track with a single measurement: degenerate by definition
      const auto zs = cams[i].projectSafe(pts[j]);
      if (!zs.second) continue;
      f->add(zs.first + Point2(0.5 * N(rng), 0.5 * N(rng)), Symbol('c', i)); used++;
    }
    graph->push_back(f);
  }
  for (int i = 0; i < nc; i++)
    initial->insert(Symbol('c', i), cams[i].retract((Vector(9) << 0.002 * N(rng), 0.002 * N(rng), 0.002 * N(rng), 0.01 * N(rng), 0.01 * N(rng), 0.01 * N(rng), N(rng), 0, 0).finished()));
  // priors on the two end cameras fix the gauge: without them the end game of the legacy parameters (identity damping down to
  // lambda = 1e-8 on a system with seven flat directions) is decided by the rounding of the solve, in the reference as well
  for (int i : {0, nc - 1})
    graph->addPrior(Symbol('c', i), initial->at<Camera>(Symbol('c', i)), noiseModel::Isotropic::Sigma(9, 0.05));
}

// An orbit scene without failed tracks (cameras on a wide arc around a compact cloud, 2 pixels of measurement noise) for
// TriangulationParameters::enableEPI: every triangulation refined by the reference's LM on TriangulationFactors.
inline void smartEpiScene(gtsam::NonlinearFactorGraph* graph, gtsam::Values* initial) {
  using namespace gtsam;
  typedef PinholeCamera<Cal3Bundler> Camera;
  std::mt19937 rng(4321);
  std::normal_distribution<double> N(0.0, 1.0);
  const int nc = 9, np = 80;
  std::vector<Camera> cams; std::vector<Point3> pts;
  for (int i = 0; i < nc; i++) {
    const double a = 0.22 * i - 0.9;
    cams.emplace_back(Pose3(Rot3::RzRyRx(0.02 * N(rng), -a, 0.02 * N(rng)), Point3(8 * std::sin(a), 0.3 * N(rng), -8 * std::cos(a))),
                      Cal3Bundler(500 + 30 * i, 2e-2 * N(rng), 2e-3 * N(rng), 0, 0));
  }
  for (int j = 0; j < np; j++) pts.emplace_back(1.2 * N(rng), 0.9 * N(rng), 1.2 * N(rng));
  SmartProjectionParams sp;
  sp.setEnableEPI(true);
  auto noise = noiseModel::Isotropic::Sigma(2, 2.0);
  for (int j = 0; j < np; j++) {
    auto f = std::make_shared<SmartProjectionFactor<Camera>>(noise, sp);
    int used = 0;
    for (int i = 0; i < nc; i++) {
      if ((i + 3 * j) % 4 == 0 && used >= 2) continue;
      const auto zs = cams[i].projectSafe(pts[j]);
      if (!zs.second) continue;
      f->add(zs.first + Point2(2.0 * N(rng), 2.0 * N(rng)), Symbol('c', i)); used++;
    }
    graph->push_back(f);
  }
  for (int i = 0; i < nc; i++)
    initial->insert(Symbol('c', i), cams[i].retract((Vector(9) << 0.01 * N(rng), 0.01 * N(rng), 0.01 * N(rng), 0.05 * N(rng), 0.05 * N(rng), 0.05 * N(rng), N(rng), 0, 0).finished()));
}
