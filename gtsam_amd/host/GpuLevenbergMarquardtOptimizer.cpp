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

#include <chrono>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <stdexcept>

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

struct GpuLevenbergMarquardtOptimizer::Impl {
  gtg_handle h = nullptr;
  std::vector<Key> keys;                 // variable id -> Key (Values order)
  std::map<Key, int32_t> id;             // Key -> variable id
  std::vector<int32_t> var_type;
  std::vector<int64_t> val_off;
  std::vector<double> packed;            // host copy of the packed values
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
    else throw std::invalid_argument("m-estimator outside the GPU path (supported: Fair, Huber, Cauchy, Tukey, Welsch, GemanMcClure)");
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

GpuLevenbergMarquardtOptimizer::GpuLevenbergMarquardtOptimizer(const NonlinearFactorGraph& graph, const Values& initial,
                                                               const LevenbergMarquardtParams& params, int device,
                                                               const ShardSpec& shards)
    : LevenbergMarquardtOptimizer(graph, initial, [&] {
        // the reference's constructor runs COLAMD when no ordering is given (LevenbergMarquardtParams.h:112-117);
        // the device path has its own elimination structure, so hand it a trivial ordering to skip that work
        if (params.ordering) return params;
        LevenbergMarquardtParams p = params; p.ordering = Ordering(initial.keys()); return p; }()),
      impl_(new Impl) { init(initial, device, shards); }

GpuLevenbergMarquardtOptimizer::GpuLevenbergMarquardtOptimizer(const NonlinearFactorGraph& graph, const Values& initial,
                                                               const Ordering& ordering,
                                                               const LevenbergMarquardtParams& params, int device,
                                                               const ShardSpec& shards)
    : LevenbergMarquardtOptimizer(graph, initial, ordering, params), impl_(new Impl) { init(initial, device, shards); }

GpuLevenbergMarquardtOptimizer::~GpuLevenbergMarquardtOptimizer() = default;

void GpuLevenbergMarquardtOptimizer::init(const Values& initial, int device, const ShardSpec& shards) {
  Impl& m = *impl_;
  // ---- variables: Values order (sorted by Key, Values.h:74-79) ---------------------------------------------
  m.val_off.push_back(0);
  for (const auto& kv : initial) {
    int32_t t;
    if (dynamic_cast<const GenericValue<Pose3>*>(&kv.value)) t = GTG_VAR_POSE3;
    else if (dynamic_cast<const GenericValue<SfmCamera>*>(&kv.value)) t = GTG_VAR_SFM_CAMERA;
    else if (dynamic_cast<const GenericValue<Point3>*>(&kv.value)) t = GTG_VAR_POINT3;
    else if (dynamic_cast<const GenericValue<Pose2>*>(&kv.value)) t = GTG_VAR_POSE2;
    else throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: unsupported value type for key " + DefaultKeyFormatter(kv.key));
    m.id[kv.key] = (int32_t)m.keys.size();
    m.keys.push_back(kv.key); m.var_type.push_back(t);
    m.val_off.push_back(m.val_off.back() + (t == GTG_VAR_POSE3 ? 12 : t == GTG_VAR_SFM_CAMERA ? 17 : 3));
  }
  m.dim_off.push_back(0);
  for (int32_t t : m.var_type) m.dim_off.push_back(m.dim_off.back() + (t == GTG_VAR_POSE3 ? 6 : t == GTG_VAR_SFM_CAMERA ? 9 : 3));
  m.packed.assign(m.val_off.back(), 0.0);
  for (size_t v = 0; v < m.keys.size(); v++) {
    double* p = m.packed.data() + m.val_off[v];
    if (m.var_type[v] == GTG_VAR_POSE3) packPose(initial.at<Pose3>(m.keys[v]), p);
    else if (m.var_type[v] == GTG_VAR_SFM_CAMERA) packCamera(initial.at<SfmCamera>(m.keys[v]), p);
    else if (m.var_type[v] == GTG_VAR_POSE2) { const Pose2 q = initial.at<Pose2>(m.keys[v]); p[0] = q.x(); p[1] = q.y(); p[2] = q.theta(); }
    else { const Point3 q = initial.at<Point3>(m.keys[v]); p[0] = q.x(); p[1] = q.y(); p[2] = q.z(); }
  }
  auto idOf = [&](Key k) { auto it = m.id.find(k); if (it == m.id.end()) throw ValuesKeyDoesNotExist("GpuLevenbergMarquardtOptimizer", k); return it->second; };

  // ---- factors: one pass, dynamic_cast to the supported types (anything else is a hard error) ------------------
  NoiseTable nt;
  std::vector<int32_t> sfm_cam, sfm_pt, sfm_nz, pj_pose, pj_pt, pj_nz, pj_cal, pj_sen, bt_1, bt_2, bt_nz, pr_var, pr_nz;
  std::vector<double> sfm_z, pj_z, calib, sensor, bt_z, pr_data;
  std::vector<int64_t> pr_off;
  std::map<const void*, int32_t> calib_id;   // shared calibration objects (Cal3_S2 or Cal3DS2) -> row of the calibration table
  std::vector<double> calib_dist;            // k1 k2 p1 p2 per row (zero for a Cal3_S2)
  std::vector<int64_t> sm_ptr(1, 0);          // smart factors: one track each
  std::vector<int32_t> sm_cam, sm_nz;
  std::vector<double> sm_z, sm_prm;
  bool any_distortion = false;
  for (const auto& f : graph_) {
    if (!f) { m.fac_map.emplace_back(-1, 0); continue; }
    if (auto s = std::dynamic_pointer_cast<SfmFactor>(f)) {
      m.fac_map.emplace_back(GTG_FAC_GENERAL_SFM, (int64_t)sfm_cam.size());
      sfm_cam.push_back(idOf(s->key1())); sfm_pt.push_back(idOf(s->key2()));
      sfm_z.push_back(s->measured().x()); sfm_z.push_back(s->measured().y());
      sfm_nz.push_back(nt.add(s->noiseModel(), 2));
    } else if (auto p = std::dynamic_pointer_cast<ProjFactor>(f)) {
      m.fac_map.emplace_back(GTG_FAC_PROJECTION, (int64_t)pj_pose.size());
      if (p->throwCheirality()) throw std::invalid_argument("GenericProjectionFactor with throwCheirality is not supported");
      pj_pose.push_back(idOf(p->key1())); pj_pt.push_back(idOf(p->key2()));
      pj_z.push_back(p->measured().x()); pj_z.push_back(p->measured().y());
      pj_nz.push_back(nt.add(p->noiseModel(), 2));
      const Cal3_S2* K = p->calibration().get();
      auto it = calib_id.find(K);
      if (it == calib_id.end()) {
        it = calib_id.emplace(K, (int32_t)(calib.size() / 5)).first;
        calib.insert(calib.end(), {K->fx(), K->fy(), K->skew(), K->px(), K->py()});
        calib_dist.insert(calib_dist.end(), {0.0, 0.0, 0.0, 0.0});
      }
      pj_cal.push_back(it->second);
      if (p->body_P_sensor()) { pj_sen.push_back((int32_t)(sensor.size() / 12)); sensor.resize(sensor.size() + 12); packPose(*p->body_P_sensor(), sensor.data() + sensor.size() - 12); }
      else pj_sen.push_back(-1);
    } else if (auto pd = std::dynamic_pointer_cast<ProjFactorDS2>(f)) {   // the same factor with a Cal3DS2 calibration (section 8(f) #3)
      m.fac_map.emplace_back(GTG_FAC_PROJECTION, (int64_t)pj_pose.size());
      if (pd->throwCheirality()) throw std::invalid_argument("GenericProjectionFactor with throwCheirality is not supported");
      pj_pose.push_back(idOf(pd->key1())); pj_pt.push_back(idOf(pd->key2()));
      pj_z.push_back(pd->measured().x()); pj_z.push_back(pd->measured().y());
      pj_nz.push_back(nt.add(pd->noiseModel(), 2));
      const Cal3DS2* K = pd->calibration().get();
      auto it = calib_id.find(K);
      if (it == calib_id.end()) {
        it = calib_id.emplace(K, (int32_t)(calib.size() / 5)).first;
        calib.insert(calib.end(), {K->fx(), K->fy(), K->skew(), K->px(), K->py()});
        calib_dist.insert(calib_dist.end(), {K->k1(), K->k2(), K->p1(), K->p2()});
        any_distortion = true;
      }
      pj_cal.push_back(it->second);
      if (pd->body_P_sensor()) { pj_sen.push_back((int32_t)(sensor.size() / 12)); sensor.resize(sensor.size() + 12); packPose(*pd->body_P_sensor(), sensor.data() + sensor.size() - 12); }
      else pj_sen.push_back(-1);
    } else if (auto sf = std::dynamic_pointer_cast<SmartFactor>(f)) {
      m.fac_map.emplace_back(-2, (int64_t)sm_nz.size());   // (no Jacobian record: its linearisation is a Hessian factor)
      const SmartProjectionParams& sp = (*sf).*SmartAccess::params();
      const TriangulationParameters& tp = sp.triangulation;
      if (sp.linearizationMode != HESSIAN) throw std::invalid_argument("SmartProjectionFactor: only the HESSIAN linearisation is supported");
      if (tp.enableEPI || tp.useLOST) throw std::invalid_argument("SmartProjectionFactor: enableEPI / useLOST are not supported");
      if (sp.throwCheirality) throw std::invalid_argument("SmartProjectionFactor with throwCheirality is not supported");
      const SharedIsotropic& iso = (*sf).*SmartAccess::noise();
      sm_nz.push_back(nt.add(iso, 2));
      const auto& zs = sf->measured();
      if (zs.size() != sf->keys().size() || zs.empty()) throw std::invalid_argument("SmartProjectionFactor: measurements and keys do not match");
      for (size_t k = 0; k < zs.size(); k++) { sm_cam.push_back(idOf(sf->keys()[k])); sm_z.push_back(zs[k].x()); sm_z.push_back(zs[k].y()); }
      sm_ptr.push_back((int64_t)sm_cam.size());
      sm_prm.insert(sm_prm.end(), {tp.rankTolerance, tp.landmarkDistanceThreshold, tp.dynamicOutlierRejectionThreshold, sp.retriangulationThreshold,
                                   sp.degeneracyMode == ZERO_ON_DEGENERACY ? 1.0 : (sp.degeneracyMode == HANDLE_INFINITY ? 2.0 : 0.0), 0.0, 0.0, 0.0});
    } else if (auto b = std::dynamic_pointer_cast<BetweenFactor<Pose3>>(f)) {
      m.fac_map.emplace_back(GTG_FAC_BETWEEN_POSE3, (int64_t)bt_1.size());
      bt_1.push_back(idOf(b->key1())); bt_2.push_back(idOf(b->key2()));
      bt_z.resize(bt_z.size() + 12); packPose(b->measured(), bt_z.data() + bt_z.size() - 12);
      bt_nz.push_back(nt.add(b->noiseModel(), 6));
    } else if (auto b2 = std::dynamic_pointer_cast<BetweenFactor<Pose2>>(f)) {
      // same table as BetweenFactor<Pose3>: the factor's type follows from its variables', (x, y, theta) in the first 3 doubles
      m.fac_map.emplace_back(GTG_FAC_BETWEEN_POSE3, (int64_t)bt_1.size());
      bt_1.push_back(idOf(b2->key1())); bt_2.push_back(idOf(b2->key2()));
      const Pose2& z = b2->measured();
      bt_z.insert(bt_z.end(), {z.x(), z.y(), z.theta(), 0, 0, 0, 0, 0, 0, 0, 0, 0});
      bt_nz.push_back(nt.add(b2->noiseModel(), 3));
    } else if (auto q2 = std::dynamic_pointer_cast<PriorFactor<Pose2>>(f)) {
      m.fac_map.emplace_back(GTG_FAC_PRIOR, (int64_t)pr_var.size());
      pr_var.push_back(idOf(q2->key())); pr_off.push_back((int64_t)pr_data.size());
      pr_data.insert(pr_data.end(), {q2->prior().x(), q2->prior().y(), q2->prior().theta()});
      pr_nz.push_back(nt.add(q2->noiseModel(), 3));
    } else if (auto pp = std::dynamic_pointer_cast<PriorFactor<Pose3>>(f)) {
      m.fac_map.emplace_back(GTG_FAC_PRIOR, (int64_t)pr_var.size());
      pr_var.push_back(idOf(pp->key())); pr_off.push_back((int64_t)pr_data.size());
      pr_data.resize(pr_data.size() + 12); packPose(pp->prior(), pr_data.data() + pr_data.size() - 12);
      pr_nz.push_back(nt.add(pp->noiseModel(), 6));
    } else if (auto pc = std::dynamic_pointer_cast<PriorFactor<SfmCamera>>(f)) {
      m.fac_map.emplace_back(GTG_FAC_PRIOR, (int64_t)pr_var.size());
      pr_var.push_back(idOf(pc->key())); pr_off.push_back((int64_t)pr_data.size());
      pr_data.resize(pr_data.size() + 17); packCamera(pc->prior(), pr_data.data() + pr_data.size() - 17);
      pr_nz.push_back(nt.add(pc->noiseModel(), 9));
    } else if (auto p3 = std::dynamic_pointer_cast<PriorFactor<Point3>>(f)) {
      m.fac_map.emplace_back(GTG_FAC_PRIOR, (int64_t)pr_var.size());
      pr_var.push_back(idOf(p3->key())); pr_off.push_back((int64_t)pr_data.size());
      pr_data.insert(pr_data.end(), {p3->prior().x(), p3->prior().y(), p3->prior().z()});
      pr_nz.push_back(nt.add(p3->noiseModel(), 3));
    } else {
      throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: factor type outside the GPU hot path "
                                  "(supported: GeneralSFMFactor<SfmCamera,Point3>, GenericProjectionFactor<Pose3,Point3,Cal3_S2|Cal3DS2>, "
                                  "SmartProjectionFactor<SfmCamera>, BetweenFactor<Pose3|Pose2>, PriorFactor<Pose3|Pose2|SfmCamera|Point3>)");
    }
  }
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
  if (pb.n_smart && shards.n_shards > 1) throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: smart factors on a sharded graph are not supported");
  pb.n_prior = (int64_t)pr_var.size(); pb.prior_var = pr_var.data(); pb.prior_off = pr_off.data(); pb.prior_data = pr_data.data(); pb.prior_noise = pr_nz.data();

  if (shards.n_shards < 1 || shards.shard < 0 || shards.shard >= shards.n_shards) throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: bad ShardSpec");
  if (shards.n_shards > 1 && !shards.allreduce) throw std::invalid_argument("GpuLevenbergMarquardtOptimizer: n_shards > 1 needs an all-reduce callback");
  check(gtg_create(&m.h, device), "gtg_create");
  if (shards.allreduce) check(gtg_set_allreduce(m.h, shards.allreduce, shards.user), "gtg_set_allreduce");   // before the upload: it verifies the layout across the shards
  check(gtg_upload_problem(m.h, &pb, shards.shard, shards.n_shards), "gtg_upload_problem");
  check(gtg_set_values(m.h, m.packed.data(), (int64_t)m.packed.size()), "gtg_set_values");
  const State* s = static_cast<const State*>(state_.get());
  m.error = s->error; m.lambda = s->lambda; m.factor = s->currentFactor; m.iterations = s->iterations; m.inner = s->totalNumberInnerIterations;
}

void GpuLevenbergMarquardtOptimizer::syncValuesToHost(bool force) {
  Impl& m = *impl_;
  if (!m.host_values_stale && !force) return;
  check(gtg_get_values(m.h, m.packed.data(), (int64_t)m.packed.size()), "gtg_get_values");
  Values vals;
  for (size_t v = 0; v < m.keys.size(); v++) {
    const double* p = m.packed.data() + m.val_off[v];
    if (m.var_type[v] == GTG_VAR_POSE3) vals.insert(m.keys[v], unpackPose(p));
    else if (m.var_type[v] == GTG_VAR_SFM_CAMERA) vals.insert(m.keys[v], SfmCamera(unpackPose(p), Cal3Bundler(p[12], p[13], p[14], p[15], p[16])));
    else if (m.var_type[v] == GTG_VAR_POSE2) vals.insert(m.keys[v], Pose2(p[0], p[1], p[2]));
    else vals.insert(m.keys[v], Point3(p[0], p[1], p[2]));
  }
  state_.reset(new State(std::move(vals), m.error, m.lambda, m.factor, (unsigned)m.iterations, (unsigned)m.inner));
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
      const int d = (m.var_type[m.id.at(keys[0])] == GTG_VAR_POSE2) ? 3 : 6;   // Pose2: 3x3 blocks inside the same record
      out->emplace_shared<JacobianFactor>(keys[0], Matrix(Eigen::Map<const RowMat>(r, d, d)), keys[1], Matrix(Eigen::Map<const RowMat>(r + 36, d, d)),
                                          Vector(Eigen::Map<const Vector>(r + 72, d)));
    } else {
      const int32_t vt = m.var_type[m.id.at(keys[0])];
      const int d = vt == GTG_VAR_POSE3 ? 6 : vt == GTG_VAR_SFM_CAMERA ? 9 : 3;
      out->emplace_shared<JacobianFactor>(keys[0], Matrix(Eigen::Map<const RowMat>(r, d, d)), Vector(Eigen::Map<const Vector>(r + 81, d)));
    }
  }
  return out;
}

GaussianFactorGraph::shared_ptr GpuLevenbergMarquardtOptimizer::linearize() const {
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
  check(gtg_linearize(m.h), "gtg_linearize");   // the device's own linearisation at values(): what `gfg` was built from
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
  do {   // NonlinearOptimizer::defaultOptimize, NonlinearOptimizer.cpp:86-105
    currentError = newError;
    iterateDevice();
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
  syncValuesToHost(true);
  return values();
}

void GpuLevenbergMarquardtOptimizer::enablePhaseTiming(bool on) { gtg_enable_timing(impl_->h, on ? 1 : 0); }

std::vector<double> GpuLevenbergMarquardtOptimizer::phaseMilliseconds() const {
  std::vector<double> ms(GTG_PH_COUNT, 0.0);
  std::vector<int64_t> calls(GTG_PH_COUNT, 0);
  gtg_get_phase_ms(impl_->h, ms.data(), calls.data(), GTG_PH_COUNT);
  return ms;
}

}  // namespace gtsam_amd
