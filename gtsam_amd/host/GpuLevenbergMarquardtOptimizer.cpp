// GpuLevenbergMarquardtOptimizer.cpp -- extractor (NonlinearFactorGraph/Values -> SoA gtg_problem) and the
// host LM state machine around the C ABI.  See the header for the reference anchors.
#include "GpuLevenbergMarquardtOptimizer.h"

#include <gtsam/geometry/Cal3Bundler.h>
#include <gtsam/geometry/Cal3DS2.h>
#include <gtsam/geometry/Cal3_S2.h>
#include <gtsam/geometry/PinholeCamera.h>
#include <gtsam/geometry/Pose2.h>
#include <gtsam/geometry/Pose3.h>
#include <gtsam/linear/JacobianFactor.h>
#include <gtsam/linear/NoiseModel.h>
#include <gtsam/linear/linearExceptions.h>
#include <gtsam/linear/PCGSolver.h>
#include <gtsam/linear/Preconditioner.h>
#include <gtsam/nonlinear/PriorFactor.h>
#include <gtsam/nonlinear/internal/LevenbergMarquardtState.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/GeneralSFMFactor.h>
#include <gtsam/slam/ProjectionFactor.h>
#include <gtsam/slam/SmartProjectionFactor.h>

#include <gtsam/config.h>

// The compile-time options of the linked GTSAM change the mathematics of this path (gtsam/config.h.in:31-90).  The device code
// implements the DEFAULT-flag semantics; refuse to build the shim against a GTSAM configured otherwise.
#if defined(GTSAM_USE_QUATERNIONS)
#error "gtsam_amd: built for Rot3 as a rotation matrix (GTSAM_USE_QUATERNIONS off): Rot3 retract / Logmap differ with quaternions"
#endif
#if !defined(GTSAM_POSE3_EXPMAP) || !defined(GTSAM_ROT3_EXPMAP)
#error "gtsam_amd: the device retracts Pose3 / Rot3 with the full exponential map (GTSAM_POSE3_EXPMAP and GTSAM_ROT3_EXPMAP on)"
#endif
#if defined(GTSAM_SLOW_BUT_CORRECT_BETWEENFACTOR)
#error "gtsam_amd: BetweenFactor Jacobians are not multiplied by dLog on the device (GTSAM_SLOW_BUT_CORRECT_BETWEENFACTOR off)"
#endif
#if defined(GTSAM_SLOW_BUT_CORRECT_EXPMAP)
#error "gtsam_amd: Pose2 uses the first-order chart on the device (GTSAM_SLOW_BUT_CORRECT_EXPMAP off)"
#endif
#if !defined(GTSAM_THROW_CHEIRALITY_EXCEPTION)
#error "gtsam_amd: a point behind the camera zeroes the factor, as the reference does when it throws CheiralityException (flag on)"
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <algorithm>
#include <map>
#include <new>
#include <stdexcept>
#include <thread>

using namespace gtsam;

namespace gtsam_amd {

typedef PinholeCamera<Cal3Bundler> SfmCamera;
typedef GeneralSFMFactor<SfmCamera, Point3> SfmFactor;
typedef GenericProjectionFactor<Pose3, Point3, Cal3_S2> ProjFactor;
typedef GenericProjectionFactor<Pose3, Point3, Cal3DS2> ProjFactorDS2;
typedef SmartProjectionFactor<SfmCamera> SmartFactor;   // the factor of timing/timeSFMBALsmart.cpp
// The smart factor keeps its noise model and its parameters protected and has no accessor for them; a class derived from it may
// name them, and the pointers to member it forms are ordinary pointers to members of the factor (a maintainer binding this into
// GTSAM would add two accessors instead).
struct SmartAccess : SmartFactor {
  static SharedIsotropic SmartFactorBase<SfmCamera>::* noise() { return &SmartAccess::noiseModel_; }
  static SmartProjectionParams SmartFactor::* params() { return &SmartAccess::params_; }
};
typedef internal::LevenbergMarquardtState State;

// The States THIS class publishes have a dynamic type of their own: that -- not an address, which the allocator hands out again -- is
// how a State somebody else installed is told from one of ours (the base class's decreaseLambda() makes plain LevenbergMarquardtStates).
struct GpuState : State {
  using State::State;
  // Both constructors of LevenbergMarquardtState deep-copy the Values they are given (the Values&& one passes its argument on as an
  // lvalue, LevenbergMarquardtState.h:61-63): 16 ms for the 158 000 variables of the L1723 shape, per State.  So a GpuState is built on
  // an EMPTY Values, and the caller's map is then MOVED into the member's storage: the empty member is destroyed and a new Values is
  // constructed in its place from an rvalue (Values(Values&&), Values.h:118: the map's nodes change hands, O(1)).  `values` is declared
  // const in NonlinearOptimizerState, so it cannot be assigned or swapped -- but ending the lifetime of a const member SUBOBJECT of a heap
  // object and creating a new object of the same type in its storage is storage reuse, not modification of a const object ([basic.life]:
  // the restriction on re-creating const objects covers complete objects of static / thread / automatic storage; since C++20 the new
  // member is "transparently replaceable" under its old name, and implementations treat earlier dialects alike -- it is what
  // std::vector<T>::emplace does for a T with a const member).  Rounds 3-5 swapped through a const_cast instead, which IS a
  // modification of a const object.
  static std::unique_ptr<GpuState> make(Values&& v, double error, double lambda, double factor, unsigned iterations, unsigned inner) {
    std::unique_ptr<GpuState> s(new GpuState(Values(), error, lambda, factor, iterations, inner));
    void* at = const_cast<void*>(static_cast<const void*>(&s->values));
    s->values.~Values();
    ::new (at) Values(std::move(v));     // (Values' move constructor moves a std::map: it does not throw)
    return s;
  }
};

struct GpuLevenbergMarquardtOptimizer::Impl {
  gtg_handle h = nullptr;
  std::vector<Key> keys;                 // variable id -> Key (Values order: sorted)
  int32_t idOf(Key k) const {            // Key -> variable id: binary search over the sorted, contiguous keys
    auto it = std::lower_bound(keys.begin(), keys.end(), k);
    if (it == keys.end() || *it != k) throw ValuesKeyDoesNotExist("GpuLevenbergMarquardtOptimizer", k);
    return (int32_t)(it - keys.begin());
  }
  std::vector<int32_t> var_type;
  std::vector<int64_t> val_off;
  std::vector<double> packed;            // host copy of the packed values
  Values scratch;                        // the deep copy of the caller's Values (made beside the extraction); swapped into the State at the end of init
  std::vector<Value*> slots;             // variable id -> the GenericValue object of that variable inside the State's Values (heap objects
                                         // owned by the map's nodes: they stay where they are when the map is swapped into the next State)
  const Values* published = nullptr;     // the Values object (inside the State THIS class published last) whose nodes `slots` point into:
                                         // checked before every write through `slots` (adoptStateIfForeign)
  std::vector<std::pair<int32_t, int64_t>> fac_map;   // factor of graph_ -> (GTG_FAC_*, index in that type's table); (-1, 0): null
  std::vector<int64_t> dim_off;          // variable id -> offset in the tangent vector (delta)
  bool keep_linearization = false;       // iterate(): download the records right after gtg_linearize
  GaussianFactorGraph::shared_ptr linearization;
  bool host_values_stale = false;
  // device-side copies of the LM state while optimize() keeps Values on the GPU
  double error = 0, lambda = 0, factor = 0;
  size_t iterations = 0; int inner = 0;
  std::chrono::high_resolution_clock::time_point start = std::chrono::high_resolution_clock::now();   // logFile's seconds column
  ~Impl() { if (h) gtg_destroy(h); }
};

namespace {
void check(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + gtg_last_error());
}
void packPose(const Pose3& T, double* p) {
  const Matrix3 R = T.rotation().matrix();
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) p[3 * i + j] = R(i, j);
  p[9] = T.x(); p[10] = T.y(); p[11] = T.z();
}
Pose3 unpackPose(const double* p) {
  Matrix3 R;
  R << p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8];
  return Pose3(Rot3(R), Point3(p[9], p[10], p[11]));
}
void packCamera(const SfmCamera& c, double* p) {
  packPose(c.pose(), p);
  p[12] = c.calibration().fx(); p[13] = c.calibration().k1(); p[14] = c.calibration().k2();
  p[15] = c.calibration().px(); p[16] = c.calibration().py();
}

struct NoiseTable {
  std::vector<int32_t> kind, dim, rkind; std::vector<int64_t> off; std::vector<double> data, rparam;
  std::map<const noiseModel::Base*, int32_t> seen;        // by object: the usual case of one shared_ptr on many factors
  std::map<std::vector<double>, int32_t> seen_value;      // by value: loaders create one model object per factor (load2D / load3D)
  // noiseModel::Robust (NoiseModel.h:670-760) -> (GTG_ROBUST_*, parameter) beside the base model's row
  static void estimator(const noiseModel::Robust& rb, int32_t* rk, double* k) {
    namespace me = noiseModel::mEstimator;
    const auto& e = rb.robust();
    if (!e) throw std::invalid_argument("Robust noise model without an m-estimator");
    if (std::dynamic_pointer_cast<me::Null>(e)) { *rk = GTG_ROBUST_NONE; *k = 0.0; return; }
    if (e->reweightScheme() != me::Base::Block)
      throw std::invalid_argument("m-estimators with the Scalar re-weighting scheme are outside the GPU path");
    if (auto p = std::dynamic_pointer_cast<me::Fair>(e)) { *rk = GTG_ROBUST_FAIR; *k = p->modelParameter(); }
    else if (auto p = std::dynamic_pointer_cast<me::Huber>(e)) { *rk = GTG_ROBUST_HUBER; *k = p->modelParameter(); }
    else if (auto p = std::dynamic_pointer_cast<me::Cauchy>(e)) { *rk = GTG_ROBUST_CAUCHY; *k = p->modelParameter(); }
    else if (auto p = std::dynamic_pointer_cast<me::Tukey>(e)) { *rk = GTG_ROBUST_TUKEY; *k = p->modelParameter(); }
    else if (auto p = std::dynamic_pointer_cast<me::Welsch>(e)) { *rk = GTG_ROBUST_WELSCH; *k = p->modelParameter(); }
    else if (auto p = std::dynamic_pointer_cast<me::GemanMcClure>(e)) { *rk = GTG_ROBUST_GEMANMCCLURE; *k = p->modelParameter(); }
    else if (auto p = std::dynamic_pointer_cast<me::DCS>(e)) { *rk = GTG_ROBUST_DCS; *k = p->modelParameter(); }
    else if (auto p = std::dynamic_pointer_cast<me::L2WithDeadZone>(e)) { *rk = GTG_ROBUST_L2WITHDEADZONE; *k = p->modelParameter(); }
    // the asymmetric estimators differ from their symmetric namesakes for NEGATIVE distances only (LossFunctions.cpp:433-500); what a
    // Robust noise model hands them is a norm (NoiseModel.h:716-718, NoiseModel.cpp:697-730), so on this path they are Tukey / Cauchy
    else if (auto p = std::dynamic_pointer_cast<me::AsymmetricTukey>(e)) { *rk = GTG_ROBUST_TUKEY; *k = p->modelParameter(); }
    else if (auto p = std::dynamic_pointer_cast<me::AsymmetricCauchy>(e)) { *rk = GTG_ROBUST_CAUCHY; *k = p->modelParameter(); }
    else throw std::invalid_argument("m-estimator outside the GPU path (supported: Fair, Huber, Cauchy, Tukey, Welsch, GemanMcClure, DCS, L2WithDeadZone, "
                                     "AsymmetricTukey, AsymmetricCauchy; not: Custom)");
  }
  int32_t add(const SharedNoiseModel& outer, size_t expect_dim) {
    if (!outer) throw std::invalid_argument("factor without a noise model is not supported");
    auto it = seen.find(outer.get());
    if (it != seen.end()) return it->second;
    if (outer->dim() != expect_dim) throw std::invalid_argument("NoiseModelFactor: NoiseModel has wrong dimension");
    SharedNoiseModel nm = outer;
    int32_t rk = GTG_ROBUST_NONE; double rp = 0.0;
    if (auto rb = std::dynamic_pointer_cast<noiseModel::Robust>(outer)) { estimator(*rb, &rk, &rp); nm = rb->noise(); }
    if (nm->isConstrained() || std::dynamic_pointer_cast<noiseModel::Robust>(nm))
      throw std::invalid_argument("Constrained / nested Robust noise models are outside the GPU path");
    int32_t k; std::vector<double> params;
    if (nm->isUnit()) k = GTG_NOISE_UNIT;
    else if (auto iso = std::dynamic_pointer_cast<noiseModel::Isotropic>(nm)) { k = GTG_NOISE_ISOTROPIC; params.push_back(iso->sigma()); }
    else if (auto dg = std::dynamic_pointer_cast<noiseModel::Diagonal>(nm)) { k = GTG_NOISE_DIAGONAL; for (size_t i = 0; i < dg->dim(); i++) params.push_back(dg->sigma(i)); }
    else if (auto ga = std::dynamic_pointer_cast<noiseModel::Gaussian>(nm)) {
      k = GTG_NOISE_GAUSSIAN;
      const Matrix R = ga->R();
      for (int i = 0; i < R.rows(); i++) for (int j = 0; j < R.cols(); j++) params.push_back(R(i, j));
    } else throw std::invalid_argument("unsupported noise model type");
    // identical models share a row of the table (first-occurrence order), however many objects they are
    std::vector<double> key{(double)k, (double)nm->dim(), (double)rk, rp};
    key.insert(key.end(), params.begin(), params.end());
    auto byValue = seen_value.find(key);
    if (byValue != seen_value.end()) { seen[outer.get()] = byValue->second; return byValue->second; }
    const int32_t idx = (int32_t)kind.size();
    off.push_back((int64_t)data.size()); dim.push_back((int32_t)nm->dim());
    rkind.push_back(rk); rparam.push_back(rp);
    kind.push_back(k); data.insert(data.end(), params.begin(), params.end());
    seen_value[key] = idx;
    seen[outer.get()] = idx;
    return idx;
  }
};
}  // namespace

// Construction.  The reference's constructor (LevenbergMarquardtOptimizer.cpp:47-66, NonlinearOptimizer.cpp:41-43) copies the graph
// and the Values and evaluates graph.error(initialValues) on the host -- 0.11 s for the L1723 shape, one thread (NonlinearFactorGraph.cpp
// :170-179 is serial even with TBB).  Here the base classes are constructed on an EMPTY graph (so that neither that pass nor COLAMD
// runs: the ordering handed over is the caller's, or a trivial one -- the device path has its own elimination structure), then
// graph_ and state_ (both protected, NonlinearOptimizer.h:78-80) are set: the graph by shared_ptr copies, the state from the
// caller's Values and the error the DEVICE computed for them (k_error, <= 1e-9 of the reference's, tests/test_gpu_parity.py).
static LevenbergMarquardtParams withOrdering(const LevenbergMarquardtParams& params, const Values& initial, const Ordering* ordering) {
  LevenbergMarquardtParams p = params;
  if (ordering) p.ordering = *ordering;
  else if (!p.ordering) p.ordering = Ordering(initial.keys());
  return p;
}

GpuLevenbergMarquardtOptimizer::GpuLevenbergMarquardtOptimizer(const NonlinearFactorGraph& graph, const Values& initial,
                                                               const LevenbergMarquardtParams& params, int device,
                                                               const ShardSpec& shards)
    : LevenbergMarquardtOptimizer(NonlinearFactorGraph(), Values(), withOrdering(params, initial, nullptr)), impl_(new Impl) {
  init(graph, initial, device, shards);
}

GpuLevenbergMarquardtOptimizer::GpuLevenbergMarquardtOptimizer(const NonlinearFactorGraph& graph, const Values& initial,
                                                               const Ordering& ordering,
                                                               const LevenbergMarquardtParams& params, int device,
                                                               const ShardSpec& shards)
    : LevenbergMarquardtOptimizer(NonlinearFactorGraph(), Values(), withOrdering(params, initial, &ordering)), impl_(new Impl) {
  init(graph, initial, device, shards);
}

GpuLevenbergMarquardtOptimizer::~GpuLevenbergMarquardtOptimizer() = default;

namespace {
// What one host thread extracts from its range of the graph: the rows of every factor table in graph order, the noise models and
// calibrations as (run length, object) -- rows of the shared tables are handed out afterwards, sequentially, in first-occurrence
// order, so that the tables are those of a single-threaded pass whatever the thread count.
struct Extract {
  std::vector<int32_t> sfm_cam, sfm_pt, pj_pose, pj_pt, pj_sen, bt_1, bt_2, pr_var, sm_cam;
  std::vector<double> sfm_z, pj_z, sensor, bt_z, pr_data, sm_z, sm_prm;
  std::vector<int64_t> pr_off, sm_ptr;
  std::vector<std::pair<int32_t, int64_t>> fac_map;                       // (type, index in THIS chunk's table of that type)
  typedef std::vector<std::pair<int64_t, SharedNoiseModel>> Runs;
  Runs sfm_nz, pj_nz, bt_nz, pr_nz, sm_nz;
  std::vector<int32_t> bt_dim, pr_dim;                                      // expected noise dimension per between / prior factor
  std::vector<std::pair<const void*, int>> pj_cal;                          // calibration object per projection factor, 1 = Cal3DS2
  std::exception_ptr err; size_t err_at = 0;
  static void run(Runs& r, const SharedNoiseModel& nm) { if (!r.empty() && r.back().second.get() == nm.get()) r.back().first++; else r.emplace_back(1, nm); }
};
}  // namespace

void GpuLevenbergMarquardtOptimizer::init(const NonlinearFactorGraph& graph, const Values& initial, int device, const ShardSpec& shards) {
  Impl& m = *impl_;
  const bool timing = std::getenv("GTG_DEBUG_TIMING") != nullptr;   // host-side breakdown of the construction on stderr
  auto tprev = std::chrono::high_resolution_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::high_resolution_clock::now();
    std::fprintf(stderr, "[gtsam_amd shim ] %-46s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tprev).count());
    tprev = now;
  };
  // Two copies the optimizer owes its base class run beside the extraction, which reads the CALLER's graph meanwhile:
  //  - graph_ (NonlinearOptimizer.h:78): one shared-pointer copy per factor -- 18 ms for the 0.68 M factors of the L1723 shape;
  //  - a deep copy of the caller's Values (one heap object per variable).  It stays in the Impl: values() needs a Values object
  //    inside a State, and BOTH constructors of LevenbergMarquardtState deep-copy what they are given (the Values&& one passes
  //    its argument on as an lvalue, LevenbergMarquardtState.h:61-63), so the cheapest way to a current State is to overwrite the
  //    payloads of this copy in place (syncValuesToHost) and let the constructor copy it.
  //  - `slots` (variable id -> the GenericValue object of the copy): one walk over the copy's map, made by the same thread.
  // Two threads: the two copies are as long as each other (18 + 16 ms on the L1723 shape) and as the whole of the library's set-up beside them.
  std::exception_ptr copyErr, copyErr2;
  // (graph_: the copy used to run on a thread of its own -- a second pass of cache misses over the 0.68 M factors beside the extraction's,
  // and with the deep copy of the Values the last thing the constructor waited for.  The extraction threads make it now, each factor
  // while it is in the thread's cache anyway.)
  std::thread copier([&] { try { graph_.resize(graph.size()); } catch (...) { copyErr = std::current_exception(); } });
  std::thread copier2([&] {
    try {
      m.scratch = initial;
      m.slots.clear(); m.slots.reserve(m.scratch.size());
      for (const auto& kv : m.scratch) m.slots.push_back(const_cast<Value*>(&kv.value));   // (the GenericValue objects are non-const heap objects owned by the map's nodes)
    } catch (...) { copyErr2 = std::current_exception(); }
  });
  struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } joinCopier{copier};
  Join joinCopier2{copier2};
  // The one-time work of the process -- HIP runtime start, code-object load, function objects of every kernel -- starts on a helper
  // thread now and runs under the host passes below (gtg_prewarm is idempotent: later constructions return at once).
  std::thread prewarmer([device] { gtg_prewarm(device); });
  Join joinPrewarmer{prewarmer};

  // ---- variables: Values order (sorted by Key, Values.h:74-79).  The walk over the map that collects keys and value pointers is pointer
  // chasing (1.5 - 5 ms for the 158 000 variables of the L1723 shape, beside the two copier threads); it is cut into key ranges walked
  // side by side: the range boundaries are lower_bound()s of keys interpolated inside every symbol's index range (Symbol keys: character
  // in the top byte, Key.h / Symbol.h; plain integer keys are one such range), so the pieces are equal where the indices are dense and
  // merely unequal where they are not.  Classification (a chain of dynamic_casts) runs on the extraction's threads below, each thread its
  // range; the values are packed on threads while the factor tables are merged.
  lap("(threads started)");
  const size_t nvars = initial.size();
  m.keys.reserve(nvars);
  std::vector<const Value*> vptr; vptr.reserve(nvars);
  {
    const char* walk_env = std::getenv("GTG_VALUES_WALKERS");   // (tests: the split walk on small graphs)
    const size_t walkers = nvars < 2 ? 1 : walk_env ? (size_t)std::max(1, std::atoi(walk_env)) : nvars >= 32768 ? 4 : 1;
    std::vector<Values::deref_iterator> cut;           // walker t takes [cut[t], cut[t + 1])
    cut.push_back(initial.begin());
    if (walkers > 1) {
      struct Seg { Key first, last; };
      std::vector<Seg> segs;                            // the keys of one symbol character each
      for (auto it = initial.begin(); it != initial.end();) {
        const Key first = (*it).key, top = first >> 56;
        auto next = top == 0xFF ? initial.end() : initial.lower_bound((top + 1) << 56);
        auto last = next; --last.it_;
        segs.push_back(Seg{first, (*last).key});
        it = next;
      }
      long double total = 0;
      for (const Seg& g : segs) total += (long double)(g.last - g.first) + 1;
      size_t gi = 0; long double before = 0;
      for (size_t t = 1; t < walkers; t++) {
        const long double want = total * t / walkers;
        while (gi + 1 < segs.size() && before + (long double)(segs[gi].last - segs[gi].first) + 1 <= want) { before += (long double)(segs[gi].last - segs[gi].first) + 1; gi++; }
        const Key k = segs[gi].first + (Key)std::min<long double>(want - before, (long double)(segs[gi].last - segs[gi].first));
        cut.push_back(initial.lower_bound(k));
      }
    }
    cut.push_back(initial.end());
    lap("(variables: key ranges)");
    const size_t pieces = cut.size() - 1;
    std::vector<std::vector<Key>> pk(pieces);
    std::vector<std::vector<const Value*>> pv(pieces);
    auto walk = [&](size_t t) {
      pk[t].reserve(nvars / pieces + 16); pv[t].reserve(nvars / pieces + 16);
      for (auto it = cut[t]; it != cut[t + 1]; ++it) { const auto kv = *it; pk[t].push_back(kv.key); pv[t].push_back(&kv.value); }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < pieces; t++) pool.emplace_back(walk, t);
    walk(0);
    for (auto& th : pool) th.join();
    lap("(variables: walk)");
    for (size_t t = 0; t < pieces; t++) { m.keys.insert(m.keys.end(), pk[t].begin(), pk[t].end()); vptr.insert(vptr.end(), pv[t].begin(), pv[t].end()); }
    if (m.keys.size() != nvars) throw std::logic_error("GpuLevenbergMarquardtOptimizer: the walk over the Values lost variables");
  }
  m.var_type.assign(nvars, -1);
  // Key -> variable id: the keys are sorted, so a binary search over the contiguous array (a std::map of 158 000 keys cost 0.2 s of
  // pointer chasing for the 1.35 M lookups of the L1723 shape)
  auto idOf = [&](Key k) { return m.idOf(k); };
  lap("variables: keys + value pointers");

  // ---- factors: dynamic_cast to the supported types (anything else is a hard error), on host threads ------------------
  const size_t nfac = graph.size();
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const char* thr_env = std::getenv("GTG_HOST_THREADS");
  const size_t grain = std::getenv("GTG_EXTRACT_GRAIN") ? std::max(1, std::atoi(std::getenv("GTG_EXTRACT_GRAIN"))) : 4096;   // factors per thread at least (tests: 1)
  const size_t nthreads = std::max<size_t>(1, std::min<size_t>({(size_t)(thr_env ? std::max(1, std::atoi(thr_env)) : (int)std::min(hw, 16u)), nfac / grain + 1}));
  std::vector<Extract> part(nthreads);
  auto work = [&](size_t ti) {
    Extract& x = part[ti];
    const size_t b = nfac * ti / nthreads, e = nfac * (ti + 1) / nthreads;
    x.fac_map.reserve(e - b);
    x.sm_ptr.push_back(0);
    // (a chunk is nearly always one kind of factor: room for that kind up front -- vectors that grow by doubling go through mmap / munmap
    // above 128 KB, and 16 threads doing that held up the thread that deep-copies the Values: it was the last thing the constructor
    // waited for, 7 ms with 32 extraction threads, 0.3 ms with 16 -- round 6, L1723 shape)
    if (b < e && graph.begin()[b]) {
      const NonlinearFactor* f0 = graph.begin()[b].get();
      if (dynamic_cast<const SfmFactor*>(f0)) { x.sfm_cam.reserve(e - b); x.sfm_pt.reserve(e - b); x.sfm_z.reserve(2 * (e - b)); }
      else if (dynamic_cast<const ProjFactor*>(f0) || dynamic_cast<const ProjFactorDS2*>(f0)) { x.pj_pose.reserve(e - b); x.pj_pt.reserve(e - b); x.pj_sen.reserve(e - b); x.pj_z.reserve(2 * (e - b)); x.pj_cal.reserve(e - b); }
      else if (dynamic_cast<const BetweenFactor<Pose3>*>(f0) || dynamic_cast<const BetweenFactor<Pose2>*>(f0)) { x.bt_1.reserve(e - b); x.bt_2.reserve(e - b); x.bt_z.reserve(12 * (e - b)); x.bt_dim.reserve(e - b); }
    }
    for (size_t v = nvars * ti / nthreads; v < nvars * (ti + 1) / nthreads; v++) {   // this thread's variables: their types (-1: unsupported, reported below)
      const Value* val = vptr[v];
      m.var_type[v] = dynamic_cast<const GenericValue<Point3>*>(val) ? GTG_VAR_POINT3 : dynamic_cast<const GenericValue<SfmCamera>*>(val) ? GTG_VAR_SFM_CAMERA :
                      dynamic_cast<const GenericValue<Pose3>*>(val) ? GTG_VAR_POSE3 : dynamic_cast<const GenericValue<Pose2>*>(val) ? GTG_VAR_POSE2 : -1;
    }
    size_t i = b;
    try {
      for (; i < e; i++) {
        const auto& f = (graph_.at(i) = graph.begin()[i]);   // (graph[i] returns a COPY of the pointer: two more atomic operations per factor)
        if (!f) { x.fac_map.emplace_back(-1, 0); continue; }
        if (auto s = dynamic_cast<const SfmFactor*>(f.get())) {
          x.fac_map.emplace_back(GTG_FAC_GENERAL_SFM, (int64_t)x.sfm_cam.size());
          x.sfm_cam.push_back(idOf(s->key1())); x.sfm_pt.push_back(idOf(s->key2()));
          x.sfm_z.push_back(s->measured().x()); x.sfm_z.push_back(s->measured().y());
          Extract::run(x.sfm_nz, s->noiseModel());
        } else if (auto p = dynamic_cast<const ProjFactor*>(f.get())) {
          x.fac_map.emplace_back(GTG_FAC_PROJECTION, (int64_t)x.pj_pose.size());
          if (p->throwCheirality()) throw std::invalid_argument("GenericProjectionFactor with throwCheirality is not supported");
          x.pj_pose.push_back(idOf(p->key1())); x.pj_pt.push_back(idOf(p->key2()));
          x.pj_z.push_back(p->measured().x()); x.pj_z.push_back(p->measured().y());
          Extract::run(x.pj_nz, p->noiseModel());
          x.pj_cal.emplace_back(p->calibration().get(), 0);
          if (p->body_P_sensor()) { x.pj_sen.push_back((int32_t)(x.sensor.size() / 12)); x.sensor.resize(x.sensor.size() + 12); packPose(*p->body_P_sensor(), x.sensor.data() + x.sensor.size() - 12); }
          else x.pj_sen.push_back(-1);
        } else if (auto pd = dynamic_cast<const ProjFactorDS2*>(f.get())) {   // the same factor with a Cal3DS2 calibration (section 8(f) #3)
          x.fac_map.emplace_back(GTG_FAC_PROJECTION, (int64_t)x.pj_pose.size());
          if (pd->throwCheirality()) throw std::invalid_argument("GenericProjectionFactor with throwCheirality is not supported");
          x.pj_pose.push_back(idOf(pd->key1())); x.pj_pt.push_back(idOf(pd->key2()));
          x.pj_z.push_back(pd->measured().x()); x.pj_z.push_back(pd->measured().y());
          Extract::run(x.pj_nz, pd->noiseModel());
          x.pj_cal.emplace_back(pd->calibration().get(), 1);
          if (pd->body_P_sensor()) { x.pj_sen.push_back((int32_t)(x.sensor.size() / 12)); x.sensor.resize(x.sensor.size() + 12); packPose(*pd->body_P_sensor(), x.sensor.data() + x.sensor.size() - 12); }
          else x.pj_sen.push_back(-1);
        } else if (auto sf = dynamic_cast<const SmartFactor*>(f.get())) {
          x.fac_map.emplace_back(-2, (int64_t)x.sm_prm.size() / 8);   // (no Jacobian record: its linearisation is a Hessian factor)
          const SmartProjectionParams& sp = (*sf).*SmartAccess::params();
          const TriangulationParameters& tp = sp.triangulation;
          // HESSIAN, JACOBIAN_Q and JACOBIAN_SVD are the same normal equations (the device never forms the factor itself); an
          // IMPLICIT_SCHUR factor cannot be eliminated by the reference's direct solvers either.  (throwCheirality / verboseCheirality
          // are read by the pose-only smart factors, not by SmartProjectionFactor<CAMERA>.)
          if (sp.linearizationMode == IMPLICIT_SCHUR) throw std::invalid_argument("SmartProjectionFactor: the IMPLICIT_SCHUR linearisation is not supported");
          if (tp.useLOST) throw std::invalid_argument("SmartProjectionFactor: useLOST is not supported");
          if (tp.enableEPI && tp.noiseModel) throw std::invalid_argument("SmartProjectionFactor: enableEPI with a noise model in the triangulation parameters is not supported");
          const SharedIsotropic& iso = (*sf).*SmartAccess::noise();
          Extract::run(x.sm_nz, iso);
          const auto& zs = sf->measured();
          if (zs.size() != sf->keys().size() || zs.empty()) throw std::invalid_argument("SmartProjectionFactor: measurements and keys do not match");
          for (size_t k = 0; k < zs.size(); k++) { x.sm_cam.push_back(idOf(sf->keys()[k])); x.sm_z.push_back(zs[k].x()); x.sm_z.push_back(zs[k].y()); }
          x.sm_ptr.push_back((int64_t)x.sm_cam.size());
          x.sm_prm.insert(x.sm_prm.end(), {tp.rankTolerance, tp.landmarkDistanceThreshold, tp.dynamicOutlierRejectionThreshold, sp.retriangulationThreshold,
                                           sp.degeneracyMode == ZERO_ON_DEGENERACY ? 1.0 : (sp.degeneracyMode == HANDLE_INFINITY ? 2.0 : 0.0),
                                           sp.linearizationMode == JACOBIAN_Q ? 2.0 : (sp.linearizationMode == JACOBIAN_SVD ? 3.0 : 0.0), tp.enableEPI ? 1.0 : 0.0, 0.0});
        } else if (auto bb = dynamic_cast<const BetweenFactor<Pose3>*>(f.get())) {
          x.fac_map.emplace_back(GTG_FAC_BETWEEN_POSE3, (int64_t)x.bt_1.size());
          x.bt_1.push_back(idOf(bb->key1())); x.bt_2.push_back(idOf(bb->key2()));
          x.bt_z.resize(x.bt_z.size() + 12); packPose(bb->measured(), x.bt_z.data() + x.bt_z.size() - 12);
          Extract::run(x.bt_nz, bb->noiseModel()); x.bt_dim.push_back(6);
        } else if (auto b2 = dynamic_cast<const BetweenFactor<Pose2>*>(f.get())) {
          // same table as BetweenFactor<Pose3>: the factor's type follows from its variables', (x, y, theta) in the first 3 doubles
          x.fac_map.emplace_back(GTG_FAC_BETWEEN_POSE3, (int64_t)x.bt_1.size());
          x.bt_1.push_back(idOf(b2->key1())); x.bt_2.push_back(idOf(b2->key2()));
          const Pose2& z = b2->measured();
          x.bt_z.insert(x.bt_z.end(), {z.x(), z.y(), z.theta(), 0, 0, 0, 0, 0, 0, 0, 0, 0});
          Extract::run(x.bt_nz, b2->noiseModel()); x.bt_dim.push_back(3);
        } else if (auto q2 = dynamic_cast<const PriorFactor<Pose2>*>(f.get())) {
          x.fac_map.emplace_back(GTG_FAC_PRIOR, (int64_t)x.pr_var.size());
          x.pr_var.push_back(idOf(q2->key())); x.pr_off.push_back((int64_t)x.pr_data.size());
          x.pr_data.insert(x.pr_data.end(), {q2->prior().x(), q2->prior().y(), q2->prior().theta()});
          Extract::run(x.pr_nz, q2->noiseModel()); x.pr_dim.push_back(3);
        } else if (auto pp = dynamic_cast<const PriorFactor<Pose3>*>(f.get())) {
          x.fac_map.emplace_back(GTG_FAC_PRIOR, (int64_t)x.pr_var.size());
          x.pr_var.push_back(idOf(pp->key())); x.pr_off.push_back((int64_t)x.pr_data.size());
          x.pr_data.resize(x.pr_data.size() + 12); packPose(pp->prior(), x.pr_data.data() + x.pr_data.size() - 12);
          Extract::run(x.pr_nz, pp->noiseModel()); x.pr_dim.push_back(6);
        } else if (auto pc = dynamic_cast<const PriorFactor<SfmCamera>*>(f.get())) {
          x.fac_map.emplace_back(GTG_FAC_PRIOR, (int64_t)x.pr_var.size());
          x.pr_var.push_back(idOf(pc->key())); x.pr_off.push_back((int64_t)x.pr_data.size());
          x.pr_data.resize(x.pr_data.size() + 17); packCamera(pc->prior(), x.pr_data.data() + x.pr_data.size() - 17);
          Extract::run(x.pr_nz, pc->noiseModel()); x.pr_dim.push_back(9);
        } else if (auto p3 = dynamic_cast<const PriorFactor<Point3>*>(f.get())) {
          x.fac_map.emplace_back(GTG_FAC_PRIOR, (int64_t)x.pr_var.size());
          x.pr_var.push_back(idOf(p3->key())); x.pr_off.push_back((int64_t)x.pr_data.size());
          x.pr_data.insert(x.pr_data.end(), {p3->prior().x(), p3->prior().y(), p3->prior().z()});
          Extract::run(x.pr_nz, p3->noiseModel()); x.pr_dim.push_back(3);
        } else {
          throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: factor type outside the GPU hot path "
                                      "(supported: GeneralSFMFactor<SfmCamera,Point3>, GenericProjectionFactor<Pose3,Point3,Cal3_S2|Cal3DS2>, "
                                      "SmartProjectionFactor<SfmCamera>, BetweenFactor<Pose3|Pose2>, PriorFactor<Pose3|Pose2|SfmCamera|Point3>)");
        }
      }
    } catch (...) { x.err = std::current_exception(); x.err_at = i; }
  };
  copier.join();
  if (copyErr) std::rethrow_exception(copyErr);
  {
    std::vector<std::thread> pool;
    for (size_t ti = 1; ti < nthreads; ti++) pool.emplace_back(work, ti);
    work(0);
    for (auto& t : pool) t.join();
  }
  for (size_t v = 0; v < nvars; v++)
    if (m.var_type[v] < 0) throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: unsupported value type for key " + DefaultKeyFormatter(m.keys[v]));
  for (const Extract& x : part) if (x.err) std::rethrow_exception(x.err);   // the first offending factor in graph order
  lap("factors: extraction, variables: classification (host threads)");
  // offsets of the packed values / of the tangent vector, then the packing itself on threads beside the merge of the factor tables
  m.val_off.resize(nvars + 1); m.dim_off.resize(nvars + 1);
  m.val_off[0] = 0; m.dim_off[0] = 0;
  for (size_t v = 0; v < nvars; v++) {
    const int32_t t = m.var_type[v];
    m.val_off[v + 1] = m.val_off[v] + (t == GTG_VAR_POSE3 ? 12 : t == GTG_VAR_SFM_CAMERA ? 17 : 3);
    m.dim_off[v + 1] = m.dim_off[v] + (t == GTG_VAR_POSE3 ? 6 : t == GTG_VAR_SFM_CAMERA ? 9 : 3);
  }
  m.packed.resize((size_t)m.val_off[nvars]);
  auto pack = [&](size_t b, size_t e) {
    for (size_t v = b; v < e; v++) {
      double* p = m.packed.data() + m.val_off[v];
      const int32_t t = m.var_type[v];
      if (t == GTG_VAR_POINT3) { const Point3& q = static_cast<const GenericValue<Point3>*>(vptr[v])->value(); p[0] = q.x(); p[1] = q.y(); p[2] = q.z(); }
      else if (t == GTG_VAR_SFM_CAMERA) packCamera(static_cast<const GenericValue<SfmCamera>*>(vptr[v])->value(), p);
      else if (t == GTG_VAR_POSE3) packPose(static_cast<const GenericValue<Pose3>*>(vptr[v])->value(), p);
      else { const Pose2& q = static_cast<const GenericValue<Pose2>*>(vptr[v])->value(); p[0] = q.x(); p[1] = q.y(); p[2] = q.theta(); }
    }
  };
  const size_t npack = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(nthreads, 8), nvars / 8192 + 1));
  std::vector<std::thread> packers;
  for (size_t ti = 0; ti < npack; ti++) packers.emplace_back(pack, nvars * ti / npack, nvars * (ti + 1) / npack);
  struct JoinAll { std::vector<std::thread>& v; ~JoinAll() { for (auto& t : v) if (t.joinable()) t.join(); } } joinPackers{packers};

  // ---- merge in graph order: concatenate the tables, hand out the rows of the noise / calibration tables -----------------
  NoiseTable nt;
  // (the GeneralSFM tables and the factor map -- 30 MB on the L1723 shape -- are filled by threads, one chunk each, after this loop has
  // handed out the offsets and the noise rows in graph order: plain buffers, not value-initialised)
  struct RawI { std::unique_ptr<int32_t[]> p; size_t n = 0; void alloc(size_t k) { p.reset(new int32_t[k ? k : 1]); n = k; } int32_t* data() { return p.get(); } size_t size() const { return n; } };
  struct RawD { std::unique_ptr<double[]> p; size_t n = 0; void alloc(size_t k) { p.reset(new double[k ? k : 1]); n = k; } double* data() { return p.get(); } size_t size() const { return n; } };
  RawI sfm_cam, sfm_pt, sfm_nz;
  RawD sfm_z;
  std::vector<int32_t> pj_pose, pj_pt, pj_nz, pj_cal, pj_sen, bt_1, bt_2, bt_nz, pr_var, pr_nz;
  std::vector<double> pj_z, calib, sensor, bt_z, pr_data;
  std::vector<int64_t> pr_off;
  std::map<const void*, int32_t> calib_id;   // shared calibration objects (Cal3_S2 or Cal3DS2) -> row of the calibration table
  std::vector<double> calib_dist;            // k1 k2 p1 p2 per row (zero for a Cal3_S2)
  std::vector<int64_t> sm_ptr(1, 0);          // smart factors: one track each
  std::vector<int32_t> sm_cam, sm_nz;
  std::vector<double> sm_z, sm_prm;
  bool any_distortion = false;
  {
    size_t n_sfm = 0, n_pj = 0, n_bt = 0, n_pr = 0, n_smc = 0, n_sm = 0;
    for (const Extract& x : part) { n_sfm += x.sfm_cam.size(); n_pj += x.pj_pose.size(); n_bt += x.bt_1.size(); n_pr += x.pr_var.size(); n_smc += x.sm_cam.size(); n_sm += x.sm_prm.size() / 8; }
    sfm_cam.alloc(n_sfm); sfm_pt.alloc(n_sfm); sfm_nz.alloc(n_sfm); sfm_z.alloc(2 * n_sfm);
    pj_pose.reserve(n_pj); pj_pt.reserve(n_pj); pj_nz.reserve(n_pj); pj_cal.reserve(n_pj); pj_sen.reserve(n_pj); pj_z.reserve(2 * n_pj);
    bt_1.reserve(n_bt); bt_2.reserve(n_bt); bt_nz.reserve(n_bt); bt_z.reserve(12 * n_bt);
    m.fac_map.clear(); m.fac_map.resize(nfac);
  }
  auto cat = [](auto& dst, const auto& src) { dst.insert(dst.end(), src.begin(), src.end()); };
  struct Place { int64_t o_sfm, o_pj, o_bt, o_pr, o_sm; std::vector<std::pair<int64_t, int32_t>> sfm_rows; };   // per chunk: offsets of its tables, (count, noise row) of its GeneralSFM runs
  std::vector<Place> place(part.size());
  int64_t o_sfm_next = 0;
  for (size_t pi = 0; pi < part.size(); pi++) {
    const Extract& x = part[pi];
    const int64_t o_sfm = o_sfm_next, o_pj = (int64_t)pj_pose.size(), o_bt = (int64_t)bt_1.size(), o_pr = (int64_t)pr_var.size(),
                  o_sm = (int64_t)sm_nz.size(), o_sen = (int64_t)(sensor.size() / 12), o_prd = (int64_t)pr_data.size(), o_smc = (int64_t)sm_cam.size();
    place[pi].o_sfm = o_sfm; place[pi].o_pj = o_pj; place[pi].o_bt = o_bt; place[pi].o_pr = o_pr; place[pi].o_sm = o_sm;
    o_sfm_next += (int64_t)x.sfm_cam.size();
    for (const auto& r : x.sfm_nz) place[pi].sfm_rows.emplace_back(r.first, nt.add(r.second, 2));
    cat(pj_pose, x.pj_pose); cat(pj_pt, x.pj_pt); cat(pj_z, x.pj_z);
    for (const auto& r : x.pj_nz) pj_nz.insert(pj_nz.end(), (size_t)r.first, nt.add(r.second, 2));
    for (int32_t sidx : x.pj_sen) pj_sen.push_back(sidx < 0 ? -1 : (int32_t)(sidx + o_sen));
    cat(sensor, x.sensor);
    for (const auto& kc : x.pj_cal) {
      auto it = calib_id.find(kc.first);
      if (it == calib_id.end()) {
        it = calib_id.emplace(kc.first, (int32_t)(calib.size() / 5)).first;
        if (kc.second) {
          const Cal3DS2* K = static_cast<const Cal3DS2*>(kc.first);
          calib.insert(calib.end(), {K->fx(), K->fy(), K->skew(), K->px(), K->py()});
          calib_dist.insert(calib_dist.end(), {K->k1(), K->k2(), K->p1(), K->p2()});
          any_distortion = true;
        } else {
          const Cal3_S2* K = static_cast<const Cal3_S2*>(kc.first);
          calib.insert(calib.end(), {K->fx(), K->fy(), K->skew(), K->px(), K->py()});
          calib_dist.insert(calib_dist.end(), {0.0, 0.0, 0.0, 0.0});
        }
      }
      pj_cal.push_back(it->second);
    }
    cat(bt_1, x.bt_1); cat(bt_2, x.bt_2); cat(bt_z, x.bt_z);
    { size_t at = 0; for (const auto& r : x.bt_nz) { bt_nz.insert(bt_nz.end(), (size_t)r.first, nt.add(r.second, (size_t)x.bt_dim[at])); at += (size_t)r.first; } }
    cat(pr_var, x.pr_var); cat(pr_data, x.pr_data);
    for (int64_t o : x.pr_off) pr_off.push_back(o + o_prd);
    { size_t at = 0; for (const auto& r : x.pr_nz) { pr_nz.insert(pr_nz.end(), (size_t)r.first, nt.add(r.second, (size_t)x.pr_dim[at])); at += (size_t)r.first; } }
    cat(sm_cam, x.sm_cam); cat(sm_z, x.sm_z); cat(sm_prm, x.sm_prm);
    for (size_t k = 1; k < x.sm_ptr.size(); k++) sm_ptr.push_back(x.sm_ptr[k] + o_smc);
    for (const auto& r : x.sm_nz) sm_nz.insert(sm_nz.end(), (size_t)r.first, nt.add(r.second, 2));
  }
  {
    auto fill = [&](size_t pi) {
      const Extract& x = part[pi];
      const Place& pl = place[pi];
      const size_t k = x.sfm_cam.size();
      if (k) {
        std::memcpy(sfm_cam.data() + pl.o_sfm, x.sfm_cam.data(), 4 * k); std::memcpy(sfm_pt.data() + pl.o_sfm, x.sfm_pt.data(), 4 * k);
        std::memcpy(sfm_z.data() + 2 * pl.o_sfm, x.sfm_z.data(), 16 * k);
        int32_t* nz = sfm_nz.data() + pl.o_sfm;
        for (const auto& r : pl.sfm_rows) { std::fill(nz, nz + r.first, r.second); nz += r.first; }
      }
      auto* fm_out = m.fac_map.data() + nfac * pi / nthreads;     // (the chunk's factors: graph indices [nfac pi / nthreads, nfac (pi + 1) / nthreads))
      for (const auto& fm : x.fac_map)
        *fm_out++ = {fm.first, fm.second + (fm.first == GTG_FAC_GENERAL_SFM ? pl.o_sfm : fm.first == GTG_FAC_PROJECTION ? pl.o_pj :
                                             fm.first == GTG_FAC_BETWEEN_POSE3 ? pl.o_bt : fm.first == GTG_FAC_PRIOR ? pl.o_pr : fm.first == -2 ? pl.o_sm : 0)};
    };
    std::vector<std::thread> fillers;
    for (size_t pi = 1; pi < part.size(); pi++) fillers.emplace_back(fill, pi);
    fill(0);
    for (auto& t : fillers) t.join();
  }
  lap("factors: merge, noise / calibration tables");
  gtg_problem pb{};
  pb.n_vars = (int32_t)m.keys.size(); pb.var_type = m.var_type.data();
  pb.n_noise = (int32_t)nt.kind.size(); pb.noise_kind = nt.kind.data(); pb.noise_dim = nt.dim.data();
  pb.noise_off = nt.off.data(); pb.noise_data = nt.data.data();
  pb.noise_robust = nt.rkind.data(); pb.noise_robust_param = nt.rparam.data();
  pb.n_sfm = (int64_t)sfm_cam.size(); pb.sfm_cam = sfm_cam.data(); pb.sfm_point = sfm_pt.data(); pb.sfm_z = sfm_z.data(); pb.sfm_noise = sfm_nz.data();
  pb.n_proj = (int64_t)pj_pose.size(); pb.proj_pose = pj_pose.data(); pb.proj_point = pj_pt.data(); pb.proj_z = pj_z.data();
  pb.proj_noise = pj_nz.data(); pb.proj_calib = pj_cal.data(); pb.proj_sensor = pj_sen.data();
  pb.n_calib = (int32_t)(calib.size() / 5); pb.calib = calib.data(); pb.calib_distortion = any_distortion ? calib_dist.data() : nullptr; pb.n_sensor = (int32_t)(sensor.size() / 12); pb.sensor = sensor.data();
  pb.n_between = (int64_t)bt_1.size(); pb.between_v1 = bt_1.data(); pb.between_v2 = bt_2.data(); pb.between_z = bt_z.data(); pb.between_noise = bt_nz.data();
  pb.n_smart = (int64_t)sm_nz.size(); pb.smart_ptr = sm_ptr.data(); pb.smart_cam = sm_cam.data(); pb.smart_z = sm_z.data();
  pb.smart_noise = sm_nz.data(); pb.smart_params = sm_prm.data();
  pb.n_prior = (int64_t)pr_var.size(); pb.prior_var = pr_var.data(); pb.prior_off = pr_off.data(); pb.prior_data = pr_data.data(); pb.prior_noise = pr_nz.data();

  if (shards.n_shards < 1 || shards.shard < 0 || shards.shard >= shards.n_shards) throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: bad ShardSpec");
  if (shards.n_shards > 1 && !shards.allreduce) throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: n_shards > 1 needs an all-reduce callback");
  for (auto& t : packers) t.join();
  lap("variables: packed (threads, beside the merge)");
  // (the prewarm thread is NOT waited for here: gtg_create below blocks inside the runtime until it has started, and the upload's first
  // launches either find their kernels loaded or load them themselves, while the helper goes on to the per-try kernels)
  check(gtg_create(&m.h, device), "gtg_create");
  if (shards.allreduce) check(gtg_set_allreduce(m.h, shards.allreduce, shards.user), "gtg_set_allreduce");   // before the upload: it verifies the layout across the shards
  check(gtg_upload_problem(m.h, &pb, shards.shard, shards.n_shards), "gtg_upload_problem");
  check(gtg_set_values(m.h, m.packed.data(), (int64_t)m.packed.size()), "gtg_set_values");
  lap("library: create + upload + set_values");
  // State(initialValues, graph.error(initialValues), lambdaInitial, lambdaFactor) -- LevenbergMarquardtOptimizer.cpp:47-53 -- with
  // the error evaluated on the device
  double e0 = 0.0;
  check(gtg_error(m.h, &e0), "gtg_error");
  lap("device: initial error");
  copier2.join();
  lap("wait for the copies of the graph and the Values");
  if (copyErr) std::rethrow_exception(copyErr);
  if (copyErr2) std::rethrow_exception(copyErr2);
  // The State: the deep copy made beside the extraction moves into a GpuState (see GpuState::make: no second copy).  `slots` points
  // into the nodes of that Values' map -- heap objects that stay where they are when the map moves on into the next State -- and every
  // use checks first that state_ is still a State of this class holding that map (adoptStateIfForeign): a State published by anybody
  // else -- the inherited public tryLambda() does that -- is adopted, not written over.
  {
    if (m.slots.size() != nvars) throw std::logic_error("GpuLevenbergMarquardtOptimizer: the copy of the Values lost variables");
    std::unique_ptr<GpuState> fresh = GpuState::make(std::move(m.scratch), e0, params_.lambdaInitial, params_.lambdaFactor, 0, 0);   // (`slots`, collected by the copier thread, points into the nodes that move here)
    state_ = std::move(fresh);
    m.published = &state_->values;
  }
  const State* s = static_cast<const State*>(state_.get());
  m.error = s->error; m.lambda = s->lambda; m.factor = s->currentFactor; m.iterations = s->iterations; m.inner = s->totalNumberInnerIterations;
  lap("state");
  prewarmer.join();
  lap("wait for the prewarm thread");
}

// state_ was replaced or changed by code outside this class since this class last published it.  LevenbergMarquardtOptimizer::tryLambda is
// public and non-virtual: it solves (through the overridden solve()), retracts and evaluates on the CPU, and then EITHER installs a new plain
// LevenbergMarquardtState with a deep copy of ITS new values (accepted step, LM.cpp:246-249) OR raises lambda IN PLACE on the State it
// found (rejected step, LM.cpp:262-268).  The device follows the host in both cases:
//  - a State that is not a GpuState, or a GpuState that does not hold the map `slots` points into: values re-packed and uploaded, `slots`
//    rebuilt on its nodes (identity = the dynamic type; an address can be handed out again by the allocator between two foreign States);
//  - in every case the scalars (error, lambda, factor, both counters) are taken from the State: an in-place increaseLambda() on our own
//    State changes nothing else.
void GpuLevenbergMarquardtOptimizer::adoptStateIfForeign() const {
  Impl& m = *impl_;
  const State* st = static_cast<const State*>(state_.get());
  const bool ours = dynamic_cast<const GpuState*>(state_.get()) != nullptr && m.published == &state_->values && state_->values.size() == m.slots.size();
  if (!ours) {
    const Values& v = state_->values;
    if (v.size() != m.keys.size()) throw std::logic_error("GpuLevenbergMarquardtOptimizer: the optimizer's State holds other variables than the graph was uploaded with");
    m.slots.clear(); m.slots.reserve(m.keys.size());
    size_t id = 0;
    for (const auto& kv : v) {
      if (kv.key != m.keys[id]) throw std::logic_error("GpuLevenbergMarquardtOptimizer: the optimizer's State holds other variables than the graph was uploaded with");
      double* p = m.packed.data() + m.val_off[id];
      const int32_t t = m.var_type[id];
      if (t == GTG_VAR_POINT3) { const Point3& q = kv.value.cast<Point3>(); p[0] = q.x(); p[1] = q.y(); p[2] = q.z(); }
      else if (t == GTG_VAR_SFM_CAMERA) packCamera(kv.value.cast<SfmCamera>(), p);
      else if (t == GTG_VAR_POSE3) packPose(kv.value.cast<Pose3>(), p);
      else { const Pose2& q = kv.value.cast<Pose2>(); p[0] = q.x(); p[1] = q.y(); p[2] = q.theta(); }
      m.slots.push_back(const_cast<Value*>(&kv.value));
      id++;
    }
    check(gtg_set_values(m.h, m.packed.data(), (int64_t)m.packed.size()), "gtg_set_values");
    m.host_values_stale = false;
    m.published = &state_->values;
  }
  m.error = st->error; m.lambda = st->lambda; m.factor = st->currentFactor; m.iterations = st->iterations; m.inner = st->totalNumberInnerIterations;
}

void GpuLevenbergMarquardtOptimizer::syncValuesToHost(bool force) {
  Impl& m = *impl_;
  if (!m.host_values_stale && !force) return;
  if (m.published != &state_->values)     // (the public entry points adopt a foreign State before they touch the device: cannot happen through them)
    throw std::logic_error("GpuLevenbergMarquardtOptimizer: the optimizer's State was replaced while the values were on the device");
  if (m.host_values_stale) {
    check(gtg_get_values(m.h, m.packed.data(), (int64_t)m.packed.size()), "gtg_get_values");
    // overwrite the payloads of the State's Values in place (m.slots: the GenericValue objects by variable id), host threads
    const size_t nv = m.slots.size();
    auto work = [&](size_t b, size_t e) {
      for (size_t v = b; v < e; v++) {
        const double* p = m.packed.data() + m.val_off[v];
        Value& val = *m.slots[v];
        if (m.var_type[v] == GTG_VAR_POINT3) static_cast<GenericValue<Point3>&>(val).value() = Point3(p[0], p[1], p[2]);
        else if (m.var_type[v] == GTG_VAR_SFM_CAMERA) static_cast<GenericValue<SfmCamera>&>(val).value() = SfmCamera(unpackPose(p), Cal3Bundler(p[12], p[13], p[14], p[15], p[16]));
        else if (m.var_type[v] == GTG_VAR_POSE3) static_cast<GenericValue<Pose3>&>(val).value() = unpackPose(p);
        else static_cast<GenericValue<Pose2>&>(val).value() = Pose2(p[0], p[1], p[2]);
      }
    };
    const char* thr_env = std::getenv("GTG_HOST_THREADS");
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>({(size_t)(thr_env ? std::max(1, std::atoi(thr_env)) : (int)std::min(std::max(1u, std::thread::hardware_concurrency()), 32u)), nv / 8192 + 1}));   // (8 until round 6: the 1.8 M payloads of the Venice shape took 12 % of its optimize())
    std::vector<std::thread> pool;
    for (size_t ti = 1; ti < nthreads; ti++) pool.emplace_back(work, nv * ti / nthreads, nv * (ti + 1) / nthreads);
    work(0, nv / nthreads);
    for (auto& t : pool) t.join();
  }
  // the next State, with the device's error / lambda / counters
  std::unique_ptr<GpuState> fresh;
  if (dynamic_cast<const GpuState*>(state_.get())) {
    // ours: it takes the map of the current one over (O(1); the nodes -- and with them `slots` -- move on).  The object in the current
    // State's `values` storage is the one GpuState::make created there by placement new -- an object of type Values, not const Values --
    // so it may be moved from, through a pointer to that object (std::launder).
    Values* current = std::launder(static_cast<Values*>(const_cast<void*>(static_cast<const void*>(&state_->values))));
    fresh = GpuState::make(std::move(*current), m.error, m.lambda, m.factor, (unsigned)m.iterations, (unsigned)m.inner);
  } else {
    // a State the base class published and this class adopted: its `values` IS a const member -- deep copy (once; from here on the
    // States are ours again), `slots` follows the copy's nodes
    fresh = GpuState::make(Values(state_->values), m.error, m.lambda, m.factor, (unsigned)m.iterations, (unsigned)m.inner);
    m.slots.clear();
    for (const auto& kv : fresh->values) m.slots.push_back(const_cast<Value*>(&kv.value));
  }
  state_ = std::move(fresh);
  m.published = &state_->values;
  m.host_values_stale = false;
}

// LevenbergMarquardtOptimizer::tryLambda (LM.cpp:121-270) with solve / error / retract on the device.
bool GpuLevenbergMarquardtOptimizer::tryLambdaDevice() {
  Impl& m = *impl_;
  using std::cout; using std::endl;
  const bool verbose = params_.verbosityLM >= LevenbergMarquardtParams::TRYLAMBDA;
  const auto tryStart = std::chrono::high_resolution_clock::now();
  if (verbose) cout << "trying lambda = " << m.lambda << endl;
  if (params_.verbosityLM >= LevenbergMarquardtParams::DAMPED) cout << "building damped system with lambda " << m.lambda << endl;
  double out[4] = {0, 0, 0, 0};
  int rc;
  if (params_.isIterative()) {
    // NonlinearOptimizer::solve, Iterative branch (NonlinearOptimizer.cpp:154-172): PCGSolverParameters only, and the
    // device solver is block-Jacobi PCG on the implicit Schur complement (a Dummy preconditioner or a SubgraphSolver is
    // outside the GPU path).
    if (!params_.iterativeParams) throw std::runtime_error("NonlinearOptimizer::solve: cg parameter has to be assigned ...");
    auto pcg = std::dynamic_pointer_cast<PCGSolverParameters>(params_.iterativeParams);
    if (!pcg) throw std::runtime_error("GpuLevenbergMarquardtOptimizer: only PCGSolverParameters are handled by the GPU path");
    if (!std::dynamic_pointer_cast<BlockJacobiPreconditionerParameters>(pcg->preconditioner))
      throw std::runtime_error("GpuLevenbergMarquardtOptimizer: the GPU PCG solver is block-Jacobi preconditioned "
                               "(set PCGSolverParameters::preconditioner to BlockJacobiPreconditionerParameters)");
    const double cg[4] = {(double)pcg->maxIterations, (double)pcg->minIterations, pcg->epsilon_rel, pcg->epsilon_abs};
    int32_t cg_iterations = 0;
    rc = gtg_try_lambda_pcg(m.h, m.lambda, params_.diagonalDamping, params_.minDiagonal, params_.maxDiagonal, cg, out, &cg_iterations);
    check(rc, "gtg_try_lambda_pcg");
  } else {
    rc = gtg_try_lambda(m.h, m.lambda, params_.diagonalDamping, params_.minDiagonal, params_.maxDiagonal, out);
    check(rc, "gtg_try_lambda");
  }
  bool step_is_successful = false, stopSearchingLambda = false;
  double modelFidelity = 0.0, newError = std::numeric_limits<double>::infinity(), costChange = 0.0;
  const bool systemSolvedSuccessfully = rc != GTG_INDETERMINATE;   // else: IndeterminantLinearSystemException path, LM.cpp:158-160
  if (systemSolvedSuccessfully) {
    if (verbose) cout << "linear delta norm = " << out[3] << endl;
    const double oldLinearizedError = out[0], newlinearizedError = out[1];
    const double linearizedCostChange = oldLinearizedError - newlinearizedError;
    if (verbose) cout << "newlinearizedError = " << newlinearizedError << "  linearizedCostChange = " << linearizedCostChange << endl;
    if (linearizedCostChange >= 0) {
      newError = out[2];
      if (verbose) cout << "calculating error:" << endl << "old error (" << m.error << ") new (tentative) error (" << newError << ")" << endl;
      costChange = m.error - newError;
      if (linearizedCostChange > std::numeric_limits<double>::epsilon() * oldLinearizedError) {
        modelFidelity = costChange / linearizedCostChange;
        step_is_successful = modelFidelity > params_.minModelFidelity;
        if (verbose) cout << "modelFidelity: " << modelFidelity << endl;
      }
      const double minAbsoluteTolerance = params_.relativeErrorTol * m.error;
      if (std::abs(costChange) < minAbsoluteTolerance) {
        if (verbose)
          cout << "abs(costChange)=" << std::abs(costChange) << "  minAbsoluteTolerance=" << minAbsoluteTolerance
               << " (relativeErrorTol=" << params_.relativeErrorTol << ")" << endl;
        stopSearchingLambda = true;
      }
    }
  }
  if (params_.verbosityLM == LevenbergMarquardtParams::SUMMARY) {   // LM.cpp:223-242
    const double iterationTime = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - tryStart).count() / 1e6;
    if (m.iterations == 0) cout << "iter      cost      cost_change    lambda  success iter_time" << endl;
    cout << std::setw(4) << m.iterations << " " << std::setw(12) << newError << " " << std::setw(12) << std::setprecision(2)
         << costChange << " " << std::setw(10) << std::setprecision(2) << m.lambda << " " << std::setw(6)
         << systemSolvedSuccessfully << " " << std::setw(10) << std::setprecision(2) << iterationTime << endl;
  }
  if (step_is_successful) {   // decreaseLambda, LevenbergMarquardtState.h:81-94
    double newLambda = m.lambda, newFactor = m.factor;
    if (params_.useFixedLambdaFactor) newLambda /= m.factor;
    else { newLambda *= std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * modelFidelity - 1.0, 3)); newFactor = 2.0 * m.factor; }
    m.lambda = std::max(params_.lambdaLowerBound, newLambda); m.factor = newFactor;
    check(gtg_accept(m.h), "gtg_accept");
    m.error = newError; m.iterations += 1; m.inner += 1; m.host_values_stale = true;
    return true;
  } else if (!stopSearchingLambda) {   // increaseLambda, LevenbergMarquardtState.h:70-76
    if (verbose) cout << "increasing lambda" << endl;
    m.lambda *= m.factor; m.inner += 1;
    if (!params_.useFixedLambdaFactor) m.factor *= 2.0;
    if (m.lambda >= params_.lambdaUpperBound) {
      if (params_.verbosity >= NonlinearOptimizerParams::TERMINATION || params_.verbosityLM == LevenbergMarquardtParams::SUMMARY)
        cout << "Warning:  Levenberg-Marquardt giving up because cannot decrease error with maximum lambda" << endl;
      return true;
    }
    return false;
  }
  if (verbose) cout << "Levenberg-Marquardt: stopping as relative cost reduction is small" << endl;
  return true;
}

// writeLogFile, LM.cpp:101-118: inner iterations, seconds, error, lambda, outer iterations
void GpuLevenbergMarquardtOptimizer::writeLogFileDevice(double currentError) {
  const Impl& m = *impl_;
  if (params_.logFile.empty()) return;
  std::ofstream os(params_.logFile.c_str(), std::ios::app);
  const double timeSpent = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - m.start).count() / 1e6;
  os << m.inner << "," << timeSpent << "," << currentError << "," << m.lambda << "," << m.iterations << std::endl;
}

// LevenbergMarquardtOptimizer::iterate (LM.cpp:273-308) on the device state
void GpuLevenbergMarquardtOptimizer::iterateDevice() {
  Impl& m = *impl_;
  if (params_.verbosityLM >= LevenbergMarquardtParams::DAMPED) std::cout << "linearizing = " << std::endl;
  check(gtg_linearize(m.h), "gtg_linearize");
  if (m.keep_linearization) m.linearization = downloadLinearization();   // iterate()'s return value: the graph linearised here
  if (m.inner == 0) {   // write initial error
    writeLogFileDevice(m.error);
    if (params_.verbosityLM == LevenbergMarquardtParams::SUMMARY)
      std::cout << "Initial error: " << m.error << ", values: " << m.keys.size() << std::endl;
  }
  while (!tryLambdaDevice()) writeLogFileDevice(m.error);
}

GaussianFactorGraph::shared_ptr GpuLevenbergMarquardtOptimizer::iterate() {
  Impl& m = *impl_;
  adoptStateIfForeign();
  m.keep_linearization = true;   // what the reference returns: `linear`, the graph linearised at the values the iteration
  iterateDevice();               // started from (LevenbergMarquardtOptimizer.cpp:277,307)
  m.keep_linearization = false;
  syncValuesToHost(true);   // iterate() is a public entry point: values()/error()/lambda() must be current
  GaussianFactorGraph::shared_ptr out = m.linearization;
  m.linearization.reset();
  return out;
}

// The device's whitened records [A1 | A2 | b] (gtg_get_jacobians) as JacobianFactors without a noise model (the records are
// whitened and, for Robust models, re-weighted -- what NoiseModelFactor::linearize returns for unconstrained models,
// NonlinearFactor.cpp:150-182), one per factor of graph_, in its order.
GaussianFactorGraph::shared_ptr GpuLevenbergMarquardtOptimizer::downloadLinearization() const {
  const Impl& m = *impl_;
  static const int64_t width[4] = {26, 20, 78, 90};
  std::vector<double> rec[4];
  int64_t count[4] = {0, 0, 0, 0};
  for (const auto& tf : m.fac_map) if (tf.first >= 0) count[tf.first]++;   // (-1: null factor, -2: smart factor -- its linearisation is a
                                                                            // Hessian factor the device never forms: left empty here)
  for (int t = 0; t < 4; t++) {
    if (!count[t]) continue;
    rec[t].resize((size_t)(count[t] * width[t]));
    check(gtg_get_jacobians(m.h, t, rec[t].data(), (int64_t)rec[t].size()), "gtg_get_jacobians");
  }
  auto out = std::make_shared<GaussianFactorGraph>();
  out->reserve(graph_.size());
  typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> RowMat;
  for (size_t i = 0; i < graph_.size(); i++) {
    const auto tf = m.fac_map[i];
    if (tf.first < 0) { out->push_back(GaussianFactor::shared_ptr()); continue; }
    const double* r = rec[tf.first].data() + tf.second * width[tf.first];
    const KeyVector& keys = graph_[i]->keys();
    if (tf.first == GTG_FAC_GENERAL_SFM) {
      out->emplace_shared<JacobianFactor>(keys[0], Matrix(Eigen::Map<const RowMat>(r, 2, 9)), keys[1], Matrix(Eigen::Map<const RowMat>(r + 18, 2, 3)),
                                          Vector(Eigen::Map<const Vector>(r + 24, 2)));
    } else if (tf.first == GTG_FAC_PROJECTION) {
      out->emplace_shared<JacobianFactor>(keys[0], Matrix(Eigen::Map<const RowMat>(r, 2, 6)), keys[1], Matrix(Eigen::Map<const RowMat>(r + 12, 2, 3)),
                                          Vector(Eigen::Map<const Vector>(r + 18, 2)));
    } else if (tf.first == GTG_FAC_BETWEEN_POSE3) {
      const int d = (m.var_type[m.idOf(keys[0])] == GTG_VAR_POSE2) ? 3 : 6;   // Pose2: 3x3 blocks inside the same record
      out->emplace_shared<JacobianFactor>(keys[0], Matrix(Eigen::Map<const RowMat>(r, d, d)), keys[1], Matrix(Eigen::Map<const RowMat>(r + 36, d, d)),
                                          Vector(Eigen::Map<const Vector>(r + 72, d)));
    } else {
      const int32_t vt = m.var_type[m.idOf(keys[0])];
      const int d = vt == GTG_VAR_POSE3 ? 6 : vt == GTG_VAR_SFM_CAMERA ? 9 : 3;
      out->emplace_shared<JacobianFactor>(keys[0], Matrix(Eigen::Map<const RowMat>(r, d, d)), Vector(Eigen::Map<const Vector>(r + 81, d)));
    }
  }
  return out;
}

GaussianFactorGraph::shared_ptr GpuLevenbergMarquardtOptimizer::linearize() const {
  adoptStateIfForeign();
  const Impl& m = *impl_;
  if (m.host_values_stale) throw std::logic_error("GpuLevenbergMarquardtOptimizer::linearize: host values out of date");   // (cannot happen through the public entry points)
  check(gtg_linearize(m.h), "gtg_linearize");
  return downloadLinearization();
}

VectorValues GpuLevenbergMarquardtOptimizer::solve(const GaussianFactorGraph& gfg, const NonlinearOptimizerParams& params) const {
  const Impl& m = *impl_;
  // Is this buildDampedSystem's output for our graph (LevenbergMarquardtState.h:125-156)?  graph_.size() linear factors, then one
  // unary JacobianFactor per variable: A = I (or diag(sqrt hessian diagonal)), b = 0, Isotropic sigma = 1 / sqrt(lambda).
  const size_t nf = graph_.size(), nv = m.keys.size();
  double lambda = 0.0; bool diagonal = false, recognised = gfg.size() == nf + nv && nv > 0;
  for (size_t v = 0; recognised && v < nv; v++) {
    auto jf = std::dynamic_pointer_cast<JacobianFactor>(gfg[nf + v]);
    auto iso = jf ? std::dynamic_pointer_cast<noiseModel::Isotropic>(jf->get_model()) : nullptr;
    if (!jf || jf->size() != 1 || !iso || jf->getb().cwiseAbs().maxCoeff() != 0.0) { recognised = false; break; }
    const double l = 1.0 / (iso->sigma() * iso->sigma());
    if (v == 0) lambda = l; else if (std::abs(l - lambda) > 1e-12 * lambda) { recognised = false; break; }
    const Matrix A = jf->getA(jf->begin());
    if (!A.isDiagonal()) { recognised = false; break; }
    if ((A.diagonal().array() != 1.0).any()) diagonal = true;
  }
  if (!recognised) return LevenbergMarquardtOptimizer::solve(gfg, params);
  const auto* lm = dynamic_cast<const LevenbergMarquardtParams*>(&params);
  const double dmin = lm ? lm->minDiagonal : params_.minDiagonal, dmax = lm ? lm->maxDiagonal : params_.maxDiagonal;
  check(gtg_linearize(m.h), "gtg_linearize");   // the device's own linearisation at values(): what `gfg` should have been built from
  // ... which is VERIFIED, not assumed, for EVERY factor: gfg's [A | b] blocks are compared with the device's raw records of the same
  // factors (same keys, same numbers to 1e-9) on host threads, without constructing a single factor object.  A graph linearised at
  // other values, or with any factor edited by the caller, has the right SHAPE but not these numbers; it goes to the reference's CPU
  // solve like any other graph.  (Entries of smart factors -- Hessian factors the device never forms -- are skipped.)
  {
    static const int64_t width[4] = {26, 20, 78, 90};
    std::vector<double> rec[4];
    int64_t count[4] = {0, 0, 0, 0};
    for (const auto& tf : m.fac_map) if (tf.first >= 0) count[tf.first]++;
    for (int t = 0; t < 4; t++) {
      if (!count[t]) continue;
      rec[t].resize((size_t)(count[t] * width[t]));
      check(gtg_get_jacobians(m.h, t, rec[t].data(), (int64_t)rec[t].size()), "gtg_get_jacobians");
    }
    std::atomic<bool> same{true};
    auto work = [&](size_t b0, size_t e0) {
      for (size_t i = b0; i < e0 && same.load(std::memory_order_relaxed); i++) {
        const auto tf = m.fac_map[i];
        if (tf.first < 0) continue;
        const auto* theirs = dynamic_cast<const JacobianFactor*>(gfg[i].get());
        if (!theirs || theirs->get_model() || theirs->keys() != graph_[i]->keys()) { same = false; return; }
        const double* r = rec[tf.first].data() + tf.second * width[tf.first];
        // the record's layout (what downloadLinearization wraps): row-major blocks, then b
        int rows, nblk, boff[2], bcols[2], rhs;
        if (tf.first == GTG_FAC_GENERAL_SFM) { rows = 2; nblk = 2; boff[0] = 0; bcols[0] = 9; boff[1] = 18; bcols[1] = 3; rhs = 24; }
        else if (tf.first == GTG_FAC_PROJECTION) { rows = 2; nblk = 2; boff[0] = 0; bcols[0] = 6; boff[1] = 12; bcols[1] = 3; rhs = 18; }
        else if (tf.first == GTG_FAC_BETWEEN_POSE3) { const int d = (m.var_type[m.idOf(theirs->keys()[0])] == GTG_VAR_POSE2) ? 3 : 6; rows = d; nblk = 2; boff[0] = 0; bcols[0] = d; boff[1] = 36; bcols[1] = d; rhs = 72; }
        else { const int32_t vt = m.var_type[m.idOf(theirs->keys()[0])]; const int d = vt == GTG_VAR_POSE3 ? 6 : vt == GTG_VAR_SFM_CAMERA ? 9 : 3; rows = d; nblk = 1; boff[0] = 0; bcols[0] = d; boff[1] = 0; bcols[1] = 0; rhs = 81; }
        if ((int)theirs->size() != nblk || (int)theirs->rows() != rows) { same = false; return; }
        double scale = 1.0, worst = 0.0;
        for (int k = 0; k < nblk; k++) {
          const auto A = theirs->getA(theirs->begin() + k);
          if ((int)A.cols() != bcols[k]) { same = false; return; }
          for (int a = 0; a < rows; a++)
            for (int cc = 0; cc < bcols[k]; cc++) { const double x = A(a, cc); scale = std::max(scale, std::abs(x)); worst = std::max(worst, std::abs(x - r[boff[k] + a * bcols[k] + cc])); }
        }
        const auto bb = theirs->getb();
        for (int a = 0; a < rows; a++) { scale = std::max(scale, std::abs(bb(a))); worst = std::max(worst, std::abs(bb(a) - r[rhs + a])); }
        if (!(worst <= 1e-9 * scale)) { same = false; return; }
      }
    };
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>({(size_t)std::min(std::max(1u, std::thread::hardware_concurrency()), 16u), nf / 16384 + 1}));
    std::vector<std::thread> pool;
    for (size_t ti = 1; ti < nthreads; ti++) pool.emplace_back(work, nf * ti / nthreads, nf * (ti + 1) / nthreads);
    work(0, nf / nthreads);
    for (auto& t : pool) t.join();
    if (!same) return LevenbergMarquardtOptimizer::solve(gfg, params);
  }
  double out[4];
  int rc;
  if (params.isIterative()) {
    auto pcg = std::dynamic_pointer_cast<PCGSolverParameters>(params.iterativeParams);
    if (!pcg) throw std::runtime_error("GpuLevenbergMarquardtOptimizer::solve: only PCGSolverParameters are handled by the GPU path");
    const double cg[4] = {(double)pcg->maxIterations, (double)pcg->minIterations, pcg->epsilon_rel, pcg->epsilon_abs};
    int32_t its = 0;
    rc = gtg_try_lambda_pcg(m.h, lambda, diagonal, dmin, dmax, cg, out, &its);
  } else {
    rc = gtg_try_lambda(m.h, lambda, diagonal, dmin, dmax, out);
  }
  check(rc, "gtg_try_lambda");
  if (rc == GTG_INDETERMINATE) throw IndeterminantLinearSystemException(m.keys.empty() ? Key(0) : m.keys[0]);
  std::vector<double> delta((size_t)m.dim_off.back());
  check(gtg_get_delta(m.h, delta.data(), (int64_t)delta.size()), "gtg_get_delta");
  VectorValues x;
  for (size_t v = 0; v < nv; v++)
    x.insert(m.keys[v], Vector(Eigen::Map<const Vector>(delta.data() + m.dim_off[v], m.dim_off[v + 1] - m.dim_off[v])));
  return x;
}

const Values& GpuLevenbergMarquardtOptimizer::optimize() {
  Impl& m = *impl_;
  adoptStateIfForeign();
  const LevenbergMarquardtParams& p = params_;
  using std::cout; using std::endl;
  double currentError = m.error;
  if (currentError <= p.errorTol) {
    if (p.verbosity >= NonlinearOptimizerParams::ERROR) cout << "Exiting, as error = " << currentError << " < " << p.errorTol << endl;
    return values();
  }
  if (p.verbosity >= NonlinearOptimizerParams::VALUES) values().print("Initial values");
  if (p.verbosity >= NonlinearOptimizerParams::ERROR) cout << "Initial error: " << currentError << endl;
  if (m.iterations >= p.maxIterations) {
    if (p.verbosity >= NonlinearOptimizerParams::TERMINATION) cout << "iterations: " << m.iterations << " >? " << p.maxIterations << endl;
    return values();
  }
  double newError = currentError;
  const bool timing = std::getenv("GTG_DEBUG_TIMING") != nullptr;
  const auto tOpt = std::chrono::high_resolution_clock::now();
  do {   // NonlinearOptimizer::defaultOptimize, NonlinearOptimizer.cpp:86-105
    currentError = newError;
    const auto tIt = std::chrono::high_resolution_clock::now();
    iterateDevice();
    if (timing) std::fprintf(stderr, "[gtsam_amd shim ]   iteration %zu: %.2f ms\n", m.iterations, std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - tIt).count());
    newError = m.error;
    if (p.iterationHook) { syncValuesToHost(false); p.iterationHook(m.iterations, currentError, newError); }
    if (p.verbosity >= NonlinearOptimizerParams::VALUES) { syncValuesToHost(false); values().print("newValues"); }
    if (p.verbosity >= NonlinearOptimizerParams::ERROR) cout << "newError: " << newError << endl;
  } while (m.iterations < p.maxIterations &&
           !checkConvergence(p.relativeErrorTol, p.absoluteErrorTol, p.errorTol, currentError, newError, p.verbosity) &&
           std::isfinite(currentError));
  if (p.verbosity >= NonlinearOptimizerParams::TERMINATION) {
    cout << "iterations: " << m.iterations << " >? " << p.maxIterations << endl;
    if (m.iterations >= p.maxIterations) cout << "Terminating because reached maximum iterations" << endl;
  }
  const auto tSync = std::chrono::high_resolution_clock::now();
  syncValuesToHost(true);
  if (timing)
    std::fprintf(stderr, "[gtsam_amd shim ] optimize(): %zu iterations %.2f ms, values back into the State %.2f ms\n", m.iterations,
                 std::chrono::duration<double, std::milli>(tSync - tOpt).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - tSync).count());
  return values();
}

void GpuLevenbergMarquardtOptimizer::enablePhaseTiming(bool on) { gtg_enable_timing(impl_->h, on ? 1 : 0); }

std::vector<double> GpuLevenbergMarquardtOptimizer::phaseMilliseconds() const {
  std::vector<double> ms(GTG_PH_COUNT, 0.0);
  std::vector<int64_t> calls(GTG_PH_COUNT, 0);
  gtg_get_phase_ms(impl_->h, ms.data(), calls.data(), GTG_PH_COUNT);
  return ms;
}

std::vector<long long> GpuLevenbergMarquardtOptimizer::phaseCalls() const {
  std::vector<double> ms(GTG_PH_COUNT, 0.0);
  std::vector<int64_t> calls(GTG_PH_COUNT, 0);
  gtg_get_phase_ms(impl_->h, ms.data(), calls.data(), GTG_PH_COUNT);
  return std::vector<long long>(calls.begin(), calls.end());
}

gtg_handle GpuLevenbergMarquardtOptimizer::handle() const { return impl_->h; }

}  // namespace gtsam_amd
