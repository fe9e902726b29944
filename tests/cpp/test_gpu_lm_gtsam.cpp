// tests/cpp/test_gpu_lm_gtsam.cpp -- the drop-in claim, in the reference's own language: the same
// NonlinearFactorGraph / Values / LevenbergMarquardtParams are optimised by gtsam::LevenbergMarquardtOptimizer
// (CPU, the reference) and by gtsam_amd::GpuLevenbergMarquardtOptimizer (HIP); iteration counts, errors, lambda
// and the optimised Values must agree.  Reads like examples/SFMExample_bal.cpp / Pose3SLAMExample_g2o.cpp.
// Built by gtsam_amd/host/Makefile (needs GTSAM headers: build container only); run on the GPU box by
// tests/test_gpu_gtsam_shim.py.
#include <GpuLevenbergMarquardtOptimizer.h>
#include <gtsam/geometry/Cal3Bundler.h>
#include <gtsam/geometry/PinholeCamera.h>
#include <gtsam/geometry/Pose2.h>
#include <gtsam/inference/Symbol.h>
#include <gtsam/linear/PCGSolver.h>
#include <gtsam/linear/Preconditioner.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/GeneralSFMFactor.h>
#include <gtsam/slam/ProjectionFactor.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <array>
#include <random>
#include <sstream>
#include <unistd.h>

using namespace gtsam;
using symbol_shorthand::C;
using symbol_shorthand::P;
using symbol_shorthand::X;
typedef PinholeCamera<Cal3Bundler> SfmCamera;
typedef GeneralSFMFactor<SfmCamera, Point3> MyFactor;

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { failures++; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static double valuesDiff(const Values& a, const Values& b) {
  double worst = 0;
  for (const auto& kv : a) worst = std::max(worst, kv.value.localCoordinates_(b.at(kv.key)).cwiseAbs().maxCoeff());
  return worst;
}

// NonlinearOptimizerParams::Iterative with a block-Jacobi PCGSolver, converged tightly: the reference runs CG on the
// full damped system, the GPU path on its Schur complement -- both then take the step of the direct solvers.
static LevenbergMarquardtParams iterativeParams(LevenbergMarquardtParams params) {
  auto pcg = std::make_shared<PCGSolverParameters>(std::make_shared<BlockJacobiPreconditionerParameters>());
  pcg->maxIterations = 5000; pcg->epsilon_rel = 1e-13; pcg->epsilon_abs = 1e-26;
  params.linearSolverType = NonlinearOptimizerParams::Iterative;
  params.iterativeParams = pcg;
  return params;
}

static void compare(const char* name, const NonlinearFactorGraph& graph, const Values& initial, const LevenbergMarquardtParams& params,
                    double tol) {
  auto t0 = std::chrono::high_resolution_clock::now();
  LevenbergMarquardtOptimizer cpu(graph, initial, params);
  const Values rc = cpu.optimize();
  auto t1 = std::chrono::high_resolution_clock::now();
  gtsam_amd::GpuLevenbergMarquardtOptimizer gpu(graph, initial, params);
  const double e0 = gpu.error();
  const Values rg = gpu.optimize();
  auto t2 = std::chrono::high_resolution_clock::now();
  std::printf("%-22s init %.9g | cpu: it %zu inner %d err %.12g lambda %.3g (%.1f ms) | gpu: it %zu inner %d err %.12g lambda %.3g (%.1f ms)\n",
              name, e0, cpu.iterations(), cpu.getInnerIterations(), cpu.error(), cpu.lambda(),
              std::chrono::duration<double, std::milli>(t1 - t0).count(), gpu.iterations(), gpu.getInnerIterations(), gpu.error(),
              gpu.lambda(), std::chrono::duration<double, std::milli>(t2 - t1).count());
  EXPECT(std::abs(e0 - graph.error(initial)) <= 1e-9 * std::abs(e0), "initial error %.15g vs %.15g", e0, graph.error(initial));
  EXPECT(cpu.iterations() == gpu.iterations(), "iterations %zu vs %zu", cpu.iterations(), gpu.iterations());
  EXPECT(cpu.getInnerIterations() == gpu.getInnerIterations(), "inner iterations");
  EXPECT(std::abs(cpu.error() - gpu.error()) <= tol * std::abs(cpu.error()) + 1e-12, "final error %.15g vs %.15g", cpu.error(), gpu.error());
  EXPECT(std::abs(cpu.lambda() - gpu.lambda()) <= 1e-3 * cpu.lambda(), "lambda %.12g vs %.12g", cpu.lambda(), gpu.lambda());  // Ceres policy: lambda is the PRODUCT over all iterations of a cubic in the model fidelity, so 1e-8 differences in the errors of a noisy, slowly converging problem show up at 1e-4 here
  EXPECT(std::abs(graph.error(rg) - gpu.error()) <= 1e-9 * std::abs(gpu.error()) + 1e-12, "values()/error() out of sync");
  EXPECT(valuesDiff(rc, rg) <= 1e-5, "optimised values differ by %.3g", valuesDiff(rc, rg));
  // iterate() one step at a time keeps the host state current
  gtsam_amd::GpuLevenbergMarquardtOptimizer step(graph, initial, params);
  step.iterate();
  EXPECT(std::abs(graph.error(step.values()) - step.error()) <= 1e-9 * step.error() + 1e-12, "iterate(): state not synced");
  EXPECT(step.iterations() == 1, "iterate(): iterations()");
  // LevenbergMarquardtParams::logFile: both optimizers append the same rows (the seconds column aside)
  auto logRows = [](const std::string& path) {
    std::vector<std::array<double, 5>> rows; std::ifstream is(path); std::string line;
    while (std::getline(is, line)) { std::array<double, 5> r{}; char c; std::istringstream ss(line); ss >> r[0] >> c >> r[1] >> c >> r[2] >> c >> r[3] >> c >> r[4]; rows.push_back(r); }
    return rows;
  };
  const std::string base = "/tmp/gtsam_amd_log_" + std::to_string((long)getpid());
  LevenbergMarquardtParams lp = params;
  lp.logFile = base + "_cpu.csv"; std::remove(lp.logFile.c_str());
  { LevenbergMarquardtOptimizer o(graph, initial, lp); o.optimize(); }
  const auto rowsCpu = logRows(lp.logFile); std::remove(lp.logFile.c_str());
  lp.logFile = base + "_gpu.csv"; std::remove(lp.logFile.c_str());
  { gtsam_amd::GpuLevenbergMarquardtOptimizer o(graph, initial, lp); o.optimize(); }
  const auto rowsGpu = logRows(lp.logFile); std::remove(lp.logFile.c_str());
  EXPECT(!rowsCpu.empty() && rowsCpu.size() == rowsGpu.size(), "logFile rows %zu vs %zu", rowsCpu.size(), rowsGpu.size());
  for (size_t i = 0; i < std::min(rowsCpu.size(), rowsGpu.size()); i++) {
    EXPECT(rowsCpu[i][0] == rowsGpu[i][0] && rowsCpu[i][4] == rowsGpu[i][4], "logFile row %zu: counters", i);
    // the intermediate iterates of the noisy, slowly converging BAL case differ at the 1e-3 level between the two
    // implementations (measured: 4.5e-4; both end at the same minimum, checked above to `tol`), so the rows are compared
    // at the precision that is stable along the whole trajectory; the counters must be identical
    EXPECT(std::abs(rowsCpu[i][2] - rowsGpu[i][2]) <= 1e-2 * std::abs(rowsCpu[i][2]) + 1e-12, "logFile row %zu: error %g vs %g", i, rowsCpu[i][2], rowsGpu[i][2]);
    EXPECT(std::abs(rowsCpu[i][3] - rowsGpu[i][3]) <= 5e-2 * std::abs(rowsCpu[i][3]), "logFile row %zu: lambda %g vs %g", i, rowsCpu[i][3], rowsGpu[i][3]);
  }
}

int main() {
  std::mt19937 rng(7);
  std::normal_distribution<double> N(0.0, 1.0);
  {  // ---- bundle adjustment: SfmCamera + Point3, as examples/SFMExample_bal.cpp builds it ------------------------
    NonlinearFactorGraph graph; Values initial;
    const int nc = 6, np = 60;
    std::vector<SfmCamera> cams; std::vector<Point3> pts;
    for (int i = 0; i < nc; i++)
      cams.emplace_back(Pose3(Rot3::RzRyRx(0.05 * N(rng), 0.3 * i - 0.8, 0.05 * N(rng)), Point3(1.5 * i - 4, 0.2 * N(rng), -8)), Cal3Bundler(600 + 20 * i, 1e-3, 1e-5, 0, 0));
    for (int j = 0; j < np; j++) pts.emplace_back(3 * N(rng), 2 * N(rng), 4 + N(rng));
    auto noise = noiseModel::Isotropic::Sigma(2, 1.0);
    for (int j = 0; j < np; j++)
      for (int i = 0; i < nc; i++) {
        if ((i + j) % 3 == 0) continue;
        const auto zs = cams[i].projectSafe(pts[j]);
        if (!zs.second) continue;
        const Point2 z = zs.first + Point2(0.5 * N(rng), 0.5 * N(rng));
        graph.emplace_shared<MyFactor>(z, noise, C(i), P(j));
      }
    graph.addPrior(C(0), cams[0], noiseModel::Isotropic::Sigma(9, 0.1));
    graph.addPrior(P(0), pts[0], noiseModel::Isotropic::Sigma(3, 0.1));
    for (int i = 0; i < nc; i++) initial.insert(C(i), cams[i].retract((Vector(9) << 0.01 * N(rng), 0.01 * N(rng), 0.01 * N(rng), 0.05 * N(rng), 0.05 * N(rng), 0.05 * N(rng), N(rng), 0, 0).finished()));
    for (int j = 0; j < np; j++) initial.insert(P(j), Point3(pts[j] + Point3(0.05 * N(rng), 0.05 * N(rng), 0.05 * N(rng))));
    compare("BAL legacy/COLAMD", graph, initial, LevenbergMarquardtParams(), 1e-6);
    compare("BAL legacy/Iterative PCG", graph, initial, iterativeParams(LevenbergMarquardtParams()), 1e-6);
    LevenbergMarquardtParams ceres; LevenbergMarquardtParams::SetCeresDefaults(&ceres);
    Ordering ordering;   // Schur ordering of timing/timeSFMBAL.h:74-83
    for (int j = 0; j < np; j++) ordering.push_back(P(j));
    for (int i = 0; i < nc; i++) ordering.push_back(C(i));
    ceres.setOrdering(ordering);
    compare("BAL ceres/Schur", graph, initial, ceres, 1e-6);
  }
  {  // ---- Pose3 pose graph, as examples/Pose3SLAMExample_g2o.cpp (with LM) ----------------------------------------
    NonlinearFactorGraph graph; Values initial;
    const int n = 40;
    std::vector<Pose3> truth;
    for (int i = 0; i < n; i++) truth.emplace_back(Rot3::RzRyRx(0.1 * std::sin(0.3 * i), 0.15 * i, 0.05 * std::cos(0.2 * i)), Point3(3 * std::cos(0.3 * i), 3 * std::sin(0.3 * i), 0.1 * i));
    auto odo = noiseModel::Diagonal::Sigmas((Vector(6) << 0.05, 0.05, 0.05, 0.1, 0.1, 0.1).finished());
    Matrix6 info = Matrix6::Identity() * 50; info(0, 1) = info(1, 0) = 5; info(3, 5) = info(5, 3) = -7;
    auto loop = noiseModel::Gaussian::Information(info);
    auto addEdge = [&](int a, int b, const SharedNoiseModel& nm) {
      const Pose3 z = truth[a].between(truth[b]).retract((Vector(6) << 0.02 * N(rng), 0.02 * N(rng), 0.02 * N(rng), 0.05 * N(rng), 0.05 * N(rng), 0.05 * N(rng)).finished());
      graph.emplace_shared<BetweenFactor<Pose3>>(X(a), X(b), z, nm);
    };
    for (int i = 0; i + 1 < n; i++) addEdge(i, i + 1, odo);
    for (int i = 0; i + 10 < n; i += 3) addEdge(i, i + 10, loop);
    graph.addPrior(X(0), truth[0], noiseModel::Diagonal::Variances((Vector(6) << 1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4).finished()));
    for (int i = 0; i < n; i++) initial.insert(X(i), truth[i].retract((Vector(6) << 0.1 * N(rng), 0.1 * N(rng), 0.1 * N(rng), 0.3 * N(rng), 0.3 * N(rng), 0.3 * N(rng)).finished()));
    compare("Pose3 graph legacy", graph, initial, LevenbergMarquardtParams(), 1e-6);
    compare("Pose3 graph Iterative PCG", graph, initial, iterativeParams(LevenbergMarquardtParams()), 1e-6);
    // same graph with outlier loop closures and noiseModel::Robust on the loops (Huber) and the odometry (Cauchy)
    NonlinearFactorGraph robust;
    auto rodo = noiseModel::Robust::Create(noiseModel::mEstimator::Cauchy::Create(2.0), odo);
    auto rloop = noiseModel::Robust::Create(noiseModel::mEstimator::Huber::Create(1.345), loop);
    for (const auto& f : graph) {
      auto b = std::dynamic_pointer_cast<BetweenFactor<Pose3>>(f);
      if (!b) { robust.push_back(f); continue; }
      const bool isLoop = b->noiseModel().get() == loop.get();
      robust.emplace_shared<BetweenFactor<Pose3>>(b->key1(), b->key2(), b->measured(), isLoop ? SharedNoiseModel(rloop) : SharedNoiseModel(rodo));
    }
    robust.emplace_shared<BetweenFactor<Pose3>>(X(2), X(30), Pose3(Rot3::RzRyRx(0.5, -0.4, 1.0), Point3(4, -3, 2)), rloop);   // gross outliers
    robust.emplace_shared<BetweenFactor<Pose3>>(X(7), X(21), Pose3(Rot3::RzRyRx(-1.0, 0.2, 0.3), Point3(-5, 1, 1)), rloop);
    compare("Pose3 graph robust (Huber + Cauchy)", robust, initial, LevenbergMarquardtParams(), 1e-6);
  }
  {  // ---- Pose2 pose graph, as examples/Pose2SLAMExample_g2o.cpp (with LM) on a Manhattan-like loop ----------------------
    NonlinearFactorGraph graph; Values initial;
    const int n = 60;
    std::vector<Pose2> truth;
    for (int i = 0; i < n; i++) truth.emplace_back(4 * std::cos(0.25 * i) + 0.05 * i, 4 * std::sin(0.25 * i), 0.25 * i + M_PI / 2);
    auto odo = noiseModel::Diagonal::Sigmas(Vector3(0.05, 0.05, 0.02));
    Matrix3 info = Matrix3::Identity() * 30; info(0, 1) = info(1, 0) = 3; info(2, 2) = 200;
    auto loop = noiseModel::Gaussian::Information(info);
    auto addEdge = [&](int a, int b, const SharedNoiseModel& nm) {
      const Pose2 z = truth[a].between(truth[b]).retract(Vector3(0.03 * N(rng), 0.03 * N(rng), 0.01 * N(rng)));
      graph.emplace_shared<BetweenFactor<Pose2>>(X(a), X(b), z, nm);
    };
    for (int i = 0; i + 1 < n; i++) addEdge(i, i + 1, odo);
    for (int i = 0; i + 25 < n; i += 2) addEdge(i, i + 25, loop);
    graph.addPrior(X(0), truth[0], noiseModel::Diagonal::Variances(Vector3(1e-6, 1e-6, 1e-8)));
    for (int i = 0; i < n; i++) initial.insert(X(i), truth[i].retract(Vector3(0.2 * N(rng), 0.2 * N(rng), 0.1 * N(rng))));
    compare("Pose2 graph legacy", graph, initial, LevenbergMarquardtParams(), 1e-6);
  }
  {  // ---- unsupported content is a hard error, not a silent fallback -------------------------------------------------
    NonlinearFactorGraph graph; Values initial;
    initial.insert(X(0), Pose3()); initial.insert(X(1), Pose3());
    graph.emplace_shared<BetweenFactor<Pose3>>(X(0), X(1), Pose3(), noiseModel::Robust::Create(noiseModel::mEstimator::DCS::Create(1.0), noiseModel::Unit::Create(6)));
    bool threw = false;
    try { gtsam_amd::GpuLevenbergMarquardtOptimizer bad(graph, initial); } catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw, "m-estimators outside the supported six must be rejected");
    NonlinearFactorGraph g2;
    g2.emplace_shared<BetweenFactor<Pose3>>(X(0), X(1), Pose3(), noiseModel::Constrained::All(6));
    threw = false;
    try { gtsam_amd::GpuLevenbergMarquardtOptimizer bad(g2, initial); } catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw, "constrained noise model must be rejected");
  }
  std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
  return failures ? 1 : 0;
}
