// tests/cpp/test_shim_extractor.cpp -- the extractor of the GTSAM-side shim (NonlinearFactorGraph / Values -> the SoA tables
// of gtg_problem), on the CPU.  Runs under tools/hipstub (LD_PRELOAD; dry-run HIP runtime that hashes every host-to-device
// copy): graphs are built the way the reference's examples build them, from the reference's shipped files through the
// reference's own loaders, handed to gtsam_amd::GpuLevenbergMarquardtOptimizer, and the records of everything the library
// uploaded are printed.  tests/test_shim_extractor_cpu.py compares them with the records of the Python mirror's extractor
// for the same graphs: equal records = identical device tables (variable order, factor tables, shared noise rows, packed
// values, and the whole symbolic analysis behind them).  Also: content outside the path is rejected with the reference's
// exception types before anything reaches the device.
#include <GpuLevenbergMarquardtOptimizer.h>
#include <gtsam/geometry/Cal3Bundler.h>
#include <gtsam/geometry/PinholeCamera.h>
#include <gtsam/geometry/Pose2.h>
#include <gtsam/inference/Symbol.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <gtsam/sfm/SfmData.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/GeneralSFMFactor.h>
#include <gtsam/slam/dataset.h>

#include <cstdio>
#include <string>

using namespace gtsam;
using symbol_shorthand::C;
using symbol_shorthand::P;
typedef PinholeCamera<Cal3Bundler> SfmCamera;
typedef GeneralSFMFactor<SfmCamera, Point3> MyFactor;

extern "C" {
int hipstub_h2d_count(void) __attribute__((weak));
void hipstub_h2d_record(int i, long long* n, unsigned long long* h) __attribute__((weak));
void hipstub_reset(void) __attribute__((weak));
}

// all-reduce stand-ins for shards that are set up one after the other in this process (under the stub a "device" pointer is
// host memory): as if every shard contributed the same buffer -- true for the layout check of the upload -- or nothing
static int lockstepSum(void* ptr, int64_t n, void*, void* user) { double* p = static_cast<double*>(ptr); for (int64_t i = 0; i < n; i++) p[i] *= *static_cast<int*>(user); return 0; }
static int noSum(void*, int64_t, void*, void*) { return 0; }

static void report(const char* name) {
  std::printf("CASE %s", name);
  for (int i = 0; i < hipstub_h2d_count(); i++) { long long n; unsigned long long h; hipstub_h2d_record(i, &n, &h); std::printf(" %lld:%llu", n, h); }
  std::printf("\n");
}

int main(int argc, char** argv) {
  if (!hipstub_h2d_count) { std::printf("run under LD_PRELOAD=tools/hipstub/libhipstub.so\n"); return 2; }
  const std::string data = argc > 1 ? argv[1] : "/root/reference/examples/Data/";
  int failures = 0;
  {  // examples/SFMExample_bal.cpp:43-76 on dubrovnik-3-7-pre
    SfmData mydata = SfmData::FromBalFile(data + "dubrovnik-3-7-pre.txt");
    NonlinearFactorGraph graph;
    auto noise = noiseModel::Isotropic::Sigma(2, 1.0);
    size_t j = 0;
    for (const SfmTrack& track : mydata.tracks) {
      for (const auto& m : track.measurements) graph.emplace_shared<MyFactor>(m.second, noise, C(m.first), P(j));
      j += 1;
    }
    graph.addPrior(C(0), mydata.cameras[0], noiseModel::Isotropic::Sigma(9, 0.1));
    graph.addPrior(P(0), mydata.tracks[0].p, noiseModel::Isotropic::Sigma(3, 0.1));
    Values initial;
    size_t i = 0; j = 0;
    for (const SfmCamera& camera : mydata.cameras) initial.insert(C(i++), camera);
    for (const SfmTrack& track : mydata.tracks) initial.insert(P(j++), track.p);
    hipstub_reset();
    gtsam_amd::GpuLevenbergMarquardtOptimizer lm(graph, initial);
    report("sfmexample_bal_dubrovnik_3_7");
    // (this program runs under tools/hipstub, where kernels do not run: the initial error, which the optimizer takes from the device
    // since round 3, is checked against graph.error(initial) by the GPU test tests/cpp/test_gpu_lm_gtsam.cpp; here: the state exists,
    // holds the caller's values, and the copy of the graph is complete)
    if (lm.values().size() != initial.size() || !lm.values().equals(initial, 1e-12)) { failures++; std::printf("FAIL initial values\n"); }
    // the same graph as shard s of 2 (ShardSpec): set-up incl. the layout verification through the callback
    int world = 2;
    for (int sh = 0; sh < world; sh++) {
      try { gtsam_amd::GpuLevenbergMarquardtOptimizer part(graph, initial, LevenbergMarquardtParams(), 0, gtsam_amd::ShardSpec{sh, world, &lockstepSum, &world}); }
      catch (const std::exception& e) { failures++; std::printf("FAIL sharded construction %d: %s\n", sh, e.what()); }
    }
    bool threw = false;   // an all-reduce that does not span the shards is detected at construction
    try { gtsam_amd::GpuLevenbergMarquardtOptimizer part(graph, initial, LevenbergMarquardtParams(), 0, gtsam_amd::ShardSpec{0, world, &noSum, nullptr}); }
    catch (const std::runtime_error&) { threw = true; }
    if (!threw) { failures++; std::printf("FAIL a communicator that does not span n_shards must be rejected\n"); }
    threw = false;
    try { gtsam_amd::GpuLevenbergMarquardtOptimizer part(graph, initial, LevenbergMarquardtParams(), 0, gtsam_amd::ShardSpec{0, world, nullptr, nullptr}); }
    catch (const std::invalid_argument&) { threw = true; }
    if (!threw) { failures++; std::printf("FAIL n_shards > 1 without a callback must be rejected\n"); }
  }
  {  // examples/Pose2SLAMExample_g2o.cpp:46-67 protocol on w100.graph (load2D creates one noise model object per edge)
    auto gv = load2D(data + "w100.graph");
    NonlinearFactorGraph graph = *gv.first;
    graph.addPrior(0, Pose2(), noiseModel::Diagonal::Variances(Vector3(1e-6, 1e-6, 1e-8)));
    hipstub_reset();
    gtsam_amd::GpuLevenbergMarquardtOptimizer lm(graph, *gv.second);
    report("pose2slam_w100");
  }
  {  // Pose3SLAMExample_g2o.cpp protocol on pose3example.txt (EDGE_SE3:QUAT, full information matrices)
    auto gv = readG2o(data + "pose3example.txt", true);
    NonlinearFactorGraph graph = *gv.first;
    graph.addPrior(0, gv.second->at<Pose3>(0), noiseModel::Diagonal::Variances((Vector(6) << 1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4).finished()));
    hipstub_reset();
    gtsam_amd::GpuLevenbergMarquardtOptimizer lm(graph, *gv.second);
    report("pose3slam_pose3example");
  }
  {  // rejected before anything is uploaded: the reference's exception types
    NonlinearFactorGraph graph; Values initial;
    initial.insert(0, Pose3()); initial.insert(1, Pose3());
    graph.emplace_shared<BetweenFactor<Pose3>>(0, 7, Pose3(), noiseModel::Unit::Create(6));       // key 7 is not in Values
    bool ok = false;
    try { gtsam_amd::GpuLevenbergMarquardtOptimizer bad(graph, initial); } catch (const ValuesKeyDoesNotExist&) { ok = true; } catch (...) {}
    if (!ok) { failures++; std::printf("FAIL unknown key must raise ValuesKeyDoesNotExist\n"); }
    NonlinearFactorGraph g2; g2.emplace_shared<BetweenFactor<Pose3>>(0, 1, Pose3(), noiseModel::Constrained::All(6));
    ok = false;
    try { gtsam_amd::GpuLevenbergMarquardtOptimizer bad(g2, initial); } catch (const std::invalid_argument&) { ok = true; } catch (...) {}
    if (!ok) { failures++; std::printf("FAIL constrained noise must raise invalid_argument\n"); }
    Values v3; v3.insert(0, Rot3()); 
    ok = false;
    try { gtsam_amd::GpuLevenbergMarquardtOptimizer bad(NonlinearFactorGraph(), v3); } catch (const std::invalid_argument&) { ok = true; } catch (...) {}
    if (!ok) { failures++; std::printf("FAIL unsupported value type must raise invalid_argument\n"); }
  }
  std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
  return failures ? 1 : 0;
}
