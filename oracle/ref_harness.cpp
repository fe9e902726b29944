// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C-callable probes over the REAL reference (borglab/gtsam compiled from /root/reference by
// oracle/Makefile into oracle/_ref/libgtsam_ref.so).  Used by
//   * tests/golden/make_golden.py   to generate the committed golden fixtures,
//   * tests/ (not gpu)              to pin the numpy restatement oracle/gtsam_oracle.py,
//   * tests/ (gpu) and bench.py's cpu_baseline leg, when the prebuilt .so travelled to the box.
// Nothing in the product path (gtsam_amd/) may load this file's library.
//
// It takes the same structure-of-arrays problem description as the product's C ABI
// (include/gtsam_amd.h: gtg_problem) and builds the corresponding gtsam::NonlinearFactorGraph /
// gtsam::Values with the reference's own classes, so that every number it returns is computed by
// the reference's code paths: GeneralSFMFactor.h:127-177, ProjectionFactor.h:138-166,
// BetweenFactor.h:111-124, PriorFactor.h:98-102, NonlinearFactorGraph.cpp:170-179,239-278,
// GaussianFactorGraph.cpp:71-78,279-319, LevenbergMarquardtOptimizer.cpp:121-308.

#include "../include/gtsam_amd.h"

#include <gtsam/geometry/Cal3Bundler.h>
#include <gtsam/geometry/Cal3DS2.h>
#include <gtsam/geometry/Cal3_S2.h>
#include <gtsam/geometry/PinholeCamera.h>
#include <gtsam/geometry/Point3.h>
#include <gtsam/geometry/Pose2.h>
#include <gtsam/geometry/Pose3.h>
#include <gtsam/inference/Ordering.h>
#include <gtsam/linear/GaussianBayesNet.h>
#include <gtsam/linear/GaussianFactorGraph.h>
#include <gtsam/linear/JacobianFactor.h>
#include <gtsam/linear/NoiseModel.h>
#include <gtsam/linear/PCGSolver.h>
#include <gtsam/linear/Preconditioner.h>
#include <gtsam/linear/VectorValues.h>
#include <gtsam/linear/linearExceptions.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/PriorFactor.h>
#include <gtsam/nonlinear/Values.h>
#include <gtsam/nonlinear/internal/LevenbergMarquardtState.h>
#include <gtsam/sfm/SfmData.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/GeneralSFMFactor.h>
#include <gtsam/slam/ProjectionFactor.h>
#include <gtsam/slam/SmartProjectionFactor.h>
#include <limits>
#include <gtsam/slam/dataset.h>

#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <functional>
#include <thread>
#include <functional>
#include <vector>

using namespace gtsam;
typedef PinholeCamera<Cal3Bundler> Camera;
typedef GeneralSFMFactor<Camera, Point3> SfmFactor;
typedef GenericProjectionFactor<Pose3, Point3, Cal3_S2> ProjFactor;
typedef GenericProjectionFactor<Pose3, Point3, Cal3DS2> ProjFactorDS2;
typedef SmartProjectionFactor<Camera> SmartFactor;   // timing/timeSFMBALsmart.cpp:31   // calibration entries with distortion (gtg_problem.calib_distortion)

namespace {

Pose3 unpackPose(const double* p) {
  Matrix3 R;
  R << p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8];
  return Pose3(Rot3(R), Point3(p[9], p[10], p[11]));
}
void packPose(const Pose3& T, double* p) {
  const Matrix3 R = T.rotation().matrix();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) p[3 * i + j] = R(i, j);
  p[9] = T.x(); p[10] = T.y(); p[11] = T.z();
}
Camera unpackCamera(const double* p) {
  return Camera(unpackPose(p), Cal3Bundler(p[12], p[13], p[14], p[15], p[16]));
}
void packCamera(const Camera& c, double* p) {
  packPose(c.pose(), p);
  p[12] = c.calibration().fx(); p[13] = c.calibration().k1(); p[14] = c.calibration().k2();
  p[15] = c.calibration().px(); p[16] = c.calibration().py();
}
int storageSize(int t) { return t == GTG_VAR_POSE3 ? 12 : t == GTG_VAR_SFM_CAMERA ? 17 : 3; }
int tangentDim(int t) { return t == GTG_VAR_POSE3 ? 6 : t == GTG_VAR_SFM_CAMERA ? 9 : 3; }   // POINT3, POSE2: 3

struct RefGraph {
  int n_vars = 0;
  std::vector<int> var_type;
  std::vector<int64_t> val_off, dim_off;
  int64_t val_size = 0, dim_size = 0;
  NonlinearFactorGraph graph;
  // factor index ranges per type inside `graph` (inserted in this order)
  size_t beg[4], end[4];
  Values unpack(const double* v) const {
    Values vals;
    for (int i = 0; i < n_vars; i++) {
      const double* p = v + val_off[i];
      if (var_type[i] == GTG_VAR_POSE3) vals.insert(Key(i), unpackPose(p));
      else if (var_type[i] == GTG_VAR_SFM_CAMERA) vals.insert(Key(i), unpackCamera(p));
      else if (var_type[i] == GTG_VAR_POSE2) vals.insert(Key(i), Pose2(p[0], p[1], p[2]));
      else vals.insert(Key(i), Point3(p[0], p[1], p[2]));
    }
    return vals;
  }
  void pack(const Values& vals, double* v) const {
    for (int i = 0; i < n_vars; i++) {
      double* p = v + val_off[i];
      if (var_type[i] == GTG_VAR_POSE3) packPose(vals.at<Pose3>(Key(i)), p);
      else if (var_type[i] == GTG_VAR_SFM_CAMERA) packCamera(vals.at<Camera>(Key(i)), p);
      else if (var_type[i] == GTG_VAR_POSE2) { const Pose2 q = vals.at<Pose2>(Key(i)); p[0] = q.x(); p[1] = q.y(); p[2] = q.theta(); }
      else { const Point3 q = vals.at<Point3>(Key(i)); p[0] = q.x(); p[1] = q.y(); p[2] = q.z(); }
    }
  }
  VectorValues unpackDelta(const double* d) const {
    VectorValues vv;
    for (int i = 0; i < n_vars; i++) {
      const int n = tangentDim(var_type[i]);
      vv.insert(Key(i), Eigen::Map<const Vector>(d + dim_off[i], n));
    }
    return vv;
  }
  void packDelta(const VectorValues& vv, double* d) const {
    for (int i = 0; i < n_vars; i++) {
      const int n = tangentDim(var_type[i]);
      const Vector& v = vv.at(Key(i));
      for (int k = 0; k < n; k++) d[dim_off[i] + k] = v(k);
    }
  }
  Ordering ordering(int kind) const {
    if (kind == 1) {  // Schur ordering of timing/timeSFMBAL.h:74-83: all points, then the rest
      Ordering o;
      for (int i = 0; i < n_vars; i++) if (var_type[i] == GTG_VAR_POINT3) o.push_back(Key(i));
      for (int i = 0; i < n_vars; i++) if (var_type[i] != GTG_VAR_POINT3) o.push_back(Key(i));
      return o;
    }
    return Ordering::Create(Ordering::COLAMD, graph);
  }
};

SharedNoiseModel makeBaseNoise(const gtg_problem* p, int idx) {
  const int kind = p->noise_kind[idx], dim = p->noise_dim[idx];
  const double* d = p->noise_data + p->noise_off[idx];
  switch (kind) {
    case GTG_NOISE_UNIT: return noiseModel::Unit::Create(dim);
    case GTG_NOISE_ISOTROPIC: return noiseModel::Isotropic::Sigma(dim, d[0], false);
    case GTG_NOISE_DIAGONAL: return noiseModel::Diagonal::Sigmas(Eigen::Map<const Vector>(d, dim), false);
    default: {
      Matrix R(dim, dim);
      for (int i = 0; i < dim; i++) for (int j = 0; j < dim; j++) R(i, j) = d[i * dim + j];
      return noiseModel::Gaussian::SqrtInformation(R, false);
    }
  }
}
SharedNoiseModel makeNoise(const gtg_problem* p, int idx) {
  SharedNoiseModel base = makeBaseNoise(p, idx);
  const int rk = p->noise_robust ? p->noise_robust[idx] : GTG_ROBUST_NONE;
  if (rk == GTG_ROBUST_NONE) return base;
  const double c = p->noise_robust_param[idx];
  noiseModel::mEstimator::Base::shared_ptr est;
  switch (rk) {
    case GTG_ROBUST_FAIR: est = noiseModel::mEstimator::Fair::Create(c); break;
    case GTG_ROBUST_HUBER: est = noiseModel::mEstimator::Huber::Create(c); break;
    case GTG_ROBUST_CAUCHY: est = noiseModel::mEstimator::Cauchy::Create(c); break;
    case GTG_ROBUST_TUKEY: est = noiseModel::mEstimator::Tukey::Create(c); break;
    case GTG_ROBUST_WELSCH: est = noiseModel::mEstimator::Welsch::Create(c); break;
    case GTG_ROBUST_DCS: est = noiseModel::mEstimator::DCS::Create(c); break;
    case GTG_ROBUST_L2WITHDEADZONE: est = noiseModel::mEstimator::L2WithDeadZone::Create(c); break;
    default: est = noiseModel::mEstimator::GemanMcClure::Create(c); break;
  }
  return noiseModel::Robust::Create(est, base);
}

}  // namespace

extern "C" {

void* ref_graph_create(const gtg_problem* p) {
  RefGraph* g = new RefGraph;
  g->n_vars = p->n_vars;
  g->var_type.assign(p->var_type, p->var_type + p->n_vars);
  g->val_off.resize(p->n_vars); g->dim_off.resize(p->n_vars);
  for (int i = 0; i < p->n_vars; i++) {
    g->val_off[i] = g->val_size; g->dim_off[i] = g->dim_size;
    g->val_size += storageSize(p->var_type[i]); g->dim_size += tangentDim(p->var_type[i]);
  }
  std::vector<SharedNoiseModel> noise(p->n_noise);
  for (int i = 0; i < p->n_noise; i++) noise[i] = makeNoise(p, i);

  g->beg[0] = g->graph.size();
  for (int64_t i = 0; i < p->n_sfm; i++)
    g->graph.emplace_shared<SfmFactor>(Point2(p->sfm_z[2 * i], p->sfm_z[2 * i + 1]),
                                        noise[p->sfm_noise[i]], Key(p->sfm_cam[i]), Key(p->sfm_point[i]));
  g->end[0] = g->beg[1] = g->graph.size();
  std::vector<std::shared_ptr<Cal3_S2>> calibs(p->n_calib);
  for (int i = 0; i < p->n_calib; i++) {
    const double* c = p->calib + 5 * i;
    calibs[i] = std::make_shared<Cal3_S2>(c[0], c[1], c[2], c[3], c[4]);
  }
  std::vector<std::shared_ptr<Cal3DS2>> calibs_ds2(p->n_calib);   // entries with a non-zero distortion are Cal3DS2 calibrations
  if (p->calib_distortion)
    for (int i = 0; i < p->n_calib; i++) {
      const double* c = p->calib + 5 * i; const double* d = p->calib_distortion + 4 * i;
      if (d[0] != 0.0 || d[1] != 0.0 || d[2] != 0.0 || d[3] != 0.0)
        calibs_ds2[i] = std::make_shared<Cal3DS2>(c[0], c[1], c[2], c[3], c[4], d[0], d[1], d[2], d[3]);
    }
  for (int64_t i = 0; i < p->n_proj; i++) {
    const Point2 z(p->proj_z[2 * i], p->proj_z[2 * i + 1]);
    std::optional<Pose3> sensor;
    if (p->proj_sensor && p->proj_sensor[i] >= 0) sensor = unpackPose(p->sensor + 12 * p->proj_sensor[i]);
    // throwCheirality=false, verboseCheirality=false: the defaults (ProjectionFactor.h:87-92)
    if (calibs_ds2[p->proj_calib[i]])
      g->graph.emplace_shared<ProjFactorDS2>(z, noise[p->proj_noise[i]], Key(p->proj_pose[i]),
                                              Key(p->proj_point[i]), calibs_ds2[p->proj_calib[i]], sensor);
    else
      g->graph.emplace_shared<ProjFactor>(z, noise[p->proj_noise[i]], Key(p->proj_pose[i]),
                                           Key(p->proj_point[i]), calibs[p->proj_calib[i]], sensor);
  }
  g->end[1] = g->beg[2] = g->graph.size();
  for (int64_t i = 0; i < p->n_between; i++) {
    if (p->var_type[p->between_v1[i]] == GTG_VAR_POSE2) {   // measurement = first 3 of the 12 doubles
      const double* z = p->between_z + 12 * i;
      g->graph.emplace_shared<BetweenFactor<Pose2>>(Key(p->between_v1[i]), Key(p->between_v2[i]), Pose2(z[0], z[1], z[2]),
                                                     noise[p->between_noise[i]]);
    } else {
      g->graph.emplace_shared<BetweenFactor<Pose3>>(Key(p->between_v1[i]), Key(p->between_v2[i]),
                                                     unpackPose(p->between_z + 12 * i), noise[p->between_noise[i]]);
    }
  }
  g->end[2] = g->beg[3] = g->graph.size();
  for (int64_t i = 0; i < p->n_prior; i++) {
    const int v = p->prior_var[i];
    const double* d = p->prior_data + p->prior_off[i];
    const SharedNoiseModel& nm = noise[p->prior_noise[i]];
    if (p->var_type[v] == GTG_VAR_POSE3) g->graph.addPrior(Key(v), unpackPose(d), nm);
    else if (p->var_type[v] == GTG_VAR_SFM_CAMERA) g->graph.addPrior(Key(v), unpackCamera(d), nm);
    else if (p->var_type[v] == GTG_VAR_POSE2) g->graph.addPrior(Key(v), Pose2(d[0], d[1], d[2]), nm);
    else g->graph.addPrior(Key(v), Point3(d[0], d[1], d[2]), nm);
  }
  g->end[3] = g->graph.size();
  // smart factors (after the four indexed factor types): one per track, parameters as passed
  for (int64_t i = 0; i < p->n_smart; i++) {
    const double* sp = p->smart_params + 8 * i;
    const LinearizationMode lm = sp[5] == 1.0 ? IMPLICIT_SCHUR : (sp[5] == 2.0 ? JACOBIAN_Q : (sp[5] == 3.0 ? JACOBIAN_SVD : HESSIAN));
    SmartProjectionParams params(lm, sp[4] == 1.0 ? ZERO_ON_DEGENERACY : (sp[4] == 2.0 ? HANDLE_INFINITY : IGNORE_DEGENERACY), false, false, sp[3]);
    params.setRankTolerance(sp[0]);
    params.setEnableEPI(sp[6] != 0.0);
    params.setLandmarkDistanceThreshold(sp[1]);
    params.setDynamicOutlierRejectionThreshold(sp[2]);
    auto f = std::make_shared<SmartFactor>(noise[p->smart_noise[i]], params);
    for (int64_t k = p->smart_ptr[i]; k < p->smart_ptr[i + 1]; k++)
      f->add(Point2(p->smart_z[2 * k], p->smart_z[2 * k + 1]), Key(p->smart_cam[k]));
    g->graph.push_back(f);
  }
  return g;
}

void ref_graph_destroy(void* h) { delete static_cast<RefGraph*>(h); }
int64_t ref_graph_values_size(void* h) { return static_cast<RefGraph*>(h)->val_size; }
int64_t ref_graph_tangent_size(void* h) { return static_cast<RefGraph*>(h)->dim_size; }

// NaN where the reference throws a CheiralityException (a smart factor's point at infinity behind one of its cameras:
// CalibratedCamera.cpp:146-149, not caught by SmartProjectionFactor)
double ref_graph_error(void* h, const double* values) {
  RefGraph* g = static_cast<RefGraph*>(h);
  try { return g->graph.error(g->unpack(values)); } catch (const CheiralityException&) { return std::numeric_limits<double>::quiet_NaN(); }
}

// Same layout as gtg_get_jacobians(): row-major per factor [A1 | A2 | b]; PRIOR padded to 9.
int ref_graph_jacobians(void* h, const double* values, int type, double* out, int64_t n_out) {
  RefGraph* g = static_cast<RefGraph*>(h);
  const Values vals = g->unpack(values);
  int64_t pos = 0;
  for (size_t i = g->beg[type]; i < g->end[type]; i++) {
    auto jf = std::dynamic_pointer_cast<JacobianFactor>(g->graph[i]->linearize(vals));
    if (!jf) return -1;
    if (type == GTG_FAC_PRIOR) {
      if (pos + 90 > n_out) return -2;
      std::fill(out + pos, out + pos + 90, 0.0);
      const Matrix A = jf->getA(jf->begin());
      const Vector b = jf->getb();
      const int d = (int)A.rows();
      for (int r = 0; r < d; r++) for (int c = 0; c < d; c++) out[pos + r * d + c] = A(r, c);
      for (int r = 0; r < d; r++) out[pos + 81 + r] = b(r);
      pos += 90;
    } else if (type == GTG_FAC_BETWEEN_POSE3 && jf->getA(jf->begin()).rows() == 3) {
      // BetweenFactor<Pose2>: 3x3 blocks inside the 78-double between record (A1 at 0, A2 at 36, b at 72)
      if (pos + 78 > n_out) return -2;
      std::fill(out + pos, out + pos + 78, 0.0);
      int blk = 0;
      for (auto it = jf->begin(); it != jf->end(); ++it, ++blk) {
        const Matrix A = jf->getA(it);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out[pos + 36 * blk + 3 * r + c] = A(r, c);
      }
      const Vector b = jf->getb();
      for (int r = 0; r < 3; r++) out[pos + 72 + r] = b(r);
      pos += 78;
    } else {
      for (auto it = jf->begin(); it != jf->end(); ++it) {
        const Matrix A = jf->getA(it);
        if (pos + A.size() > n_out) return -2;
        for (int r = 0; r < A.rows(); r++) for (int c = 0; c < A.cols(); c++) out[pos++] = A(r, c);
      }
      const Vector b = jf->getb();
      if (pos + b.size() > n_out) return -2;
      for (int r = 0; r < b.size(); r++) out[pos++] = b(r);
    }
  }
  return 0;
}

// Dense undamped information matrix (variable id order) and gradient J^T b.
int ref_graph_hessian(void* h, const double* values, double* H, double* grad) {
  RefGraph* g = static_cast<RefGraph*>(h);
  const Values vals = g->unpack(values);
  auto lin = g->graph.linearize(vals);
  Ordering ord;
  for (int i = 0; i < g->n_vars; i++) ord.push_back(Key(i));
  auto Hg = lin->hessian(ord);
  const int64_t n = g->dim_size;
  if (H) for (int64_t r = 0; r < n; r++) for (int64_t c = 0; c < n; c++) H[r * n + c] = Hg.first(r, c);
  if (grad) for (int64_t r = 0; r < n; r++) grad[r] = Hg.second(r);
  return 0;
}

int ref_graph_hessian_diagonal(void* h, const double* values, double* d) {
  RefGraph* g = static_cast<RefGraph*>(h);
  try {
    auto lin = g->graph.linearize(g->unpack(values));
    g->packDelta(lin->hessianDiagonal(), d);
  } catch (const CheiralityException&) { return 3; }      // (see ref_graph_error)
  return 0;
}

// One tryLambda body up to the linear errors: damp + solve. Returns 1 on
// IndeterminantLinearSystemException. lin_err[0]=linear.error(0), [1]=linear.error(delta).
int ref_graph_solve(void* h, const double* values, double lambda, int diagonal_damping,
                    double min_diag, double max_diag, int ordering_kind, double* delta,
                    double* lin_err) {
  RefGraph* g = static_cast<RefGraph*>(h);
  const Values vals = g->unpack(values);
  auto lin = g->graph.linearize(vals);
  internal::LevenbergMarquardtState state(vals, 0.0, lambda, 10.0);
  GaussianFactorGraph damped;
  if (diagonal_damping) {
    VectorValues sq = lin->hessianDiagonal();
    for (auto& [key, value] : sq) value = value.cwiseMax(min_diag).cwiseMin(max_diag).cwiseSqrt();
    damped = state.buildDampedSystem(*lin, sq);
  } else {
    damped = state.buildDampedSystem(*lin);
  }
  try {
    VectorValues d = damped.optimize(g->ordering(ordering_kind), EliminatePreferCholesky);
    g->packDelta(d, delta);
    if (lin_err) { lin_err[0] = lin->error(VectorValues::Zero(d)); lin_err[1] = lin->error(d); }
  } catch (const IndeterminantLinearSystemException&) {
    return 1;
  }
  return 0;
}

// The reference's own iterative path on the same damped system: PCGSolver with a block-Jacobi preconditioner on the FULL
// system (NonlinearOptimizer.cpp:154-172 Iterative branch, PCGSolver.cpp:51-64).  It is not the Schur system the GPU path
// iterates on, so iteration counts differ; with tight tolerances both converge to the direct solution.
int ref_graph_solve_pcg(void* h, const double* values, double lambda, int diagonal_damping, double min_diag, double max_diag,
                        int max_iterations, double epsilon_rel, double epsilon_abs, double* delta) {
  RefGraph* g = static_cast<RefGraph*>(h);
  const Values vals = g->unpack(values);
  auto lin = g->graph.linearize(vals);
  internal::LevenbergMarquardtState state(vals, 0.0, lambda, 10.0);
  GaussianFactorGraph damped;
  if (diagonal_damping) {
    VectorValues sq = lin->hessianDiagonal();
    for (auto& [key, value] : sq) value = value.cwiseMax(min_diag).cwiseMin(max_diag).cwiseSqrt();
    damped = state.buildDampedSystem(*lin, sq);
  } else {
    damped = state.buildDampedSystem(*lin);
  }
  auto params = std::make_shared<PCGSolverParameters>();
  params->preconditioner = std::make_shared<BlockJacobiPreconditionerParameters>();
  params->maxIterations = max_iterations; params->epsilon_rel = epsilon_rel; params->epsilon_abs = epsilon_abs;
  PCGSolver solver(*params);
  Ordering ord;
  for (int i = 0; i < g->n_vars; i++) ord.push_back(Key(i));
  const KeyInfo info(damped, ord);
  const std::map<Key, Vector> no_lambda;
  const VectorValues d = solver.optimize(damped, info, no_lambda, info.x0());
  g->packDelta(d, delta);
  return 0;
}

int ref_graph_retract(void* h, const double* values, const double* delta, double* out) {
  RefGraph* g = static_cast<RefGraph*>(h);
  g->pack(g->unpack(values).retract(g->unpackDelta(delta)), out);
  return 0;
}

struct ref_lm_params {
  int32_t max_iterations; double relative_error_tol, absolute_error_tol, error_tol;
  double lambda_initial, lambda_factor, lambda_upper_bound, lambda_lower_bound, min_model_fidelity;
  int32_t diagonal_damping, use_fixed_lambda_factor; double min_diagonal, max_diagonal;
  int32_t ordering_kind;  // 0 COLAMD, 1 Schur (points first)
};

static LevenbergMarquardtParams lm_params_from(const ref_lm_params* rp) {
  LevenbergMarquardtParams params;
  params.maxIterations = rp->max_iterations;
  params.relativeErrorTol = rp->relative_error_tol;
  params.absoluteErrorTol = rp->absolute_error_tol;
  params.errorTol = rp->error_tol;
  params.lambdaInitial = rp->lambda_initial;
  params.lambdaFactor = rp->lambda_factor;
  params.lambdaUpperBound = rp->lambda_upper_bound;
  params.lambdaLowerBound = rp->lambda_lower_bound;
  params.minModelFidelity = rp->min_model_fidelity;
  params.diagonalDamping = rp->diagonal_damping;
  params.useFixedLambdaFactor = rp->use_fixed_lambda_factor;
  params.minDiagonal = rp->min_diagonal;
  params.maxDiagonal = rp->max_diagonal;
  return params;
}

// The reference's own optimize() with LevenbergMarquardtParams::logFile set: the CSV that
// LevenbergMarquardtOptimizer::writeLogFile (LevenbergMarquardtOptimizer.cpp:101-118) appends to `log_path`
// (inner iterations, seconds, error, lambda, outer iterations).  Returns the outer iteration count.
int ref_graph_lm_logfile(void* h, const double* values0, const ref_lm_params* rp, const char* log_path, double* values_out) {
  RefGraph* g = static_cast<RefGraph*>(h);
  LevenbergMarquardtParams params = lm_params_from(rp);
  params.logFile = log_path;
  if (rp->ordering_kind == 1) params.ordering = g->ordering(1);
  LevenbergMarquardtOptimizer lm(g->graph, g->unpack(values0), params);
  lm.optimize();
  if (values_out) g->pack(lm.values(), values_out);
  return (int)lm.iterations();
}

// Runs the reference's LM (defaultOptimize loop restated around lm.iterate() so that a
// per-outer-iteration trace can be recorded). trace rows: [inner_iterations, error, lambda, seconds].
int ref_graph_lm(void* h, const double* values0, const ref_lm_params* rp, double* values_out,
                 int max_trace, double* trace, int* n_trace, double* total_seconds) {
  RefGraph* g = static_cast<RefGraph*>(h);
  LevenbergMarquardtParams params;
  params.maxIterations = rp->max_iterations;
  params.relativeErrorTol = rp->relative_error_tol;
  params.absoluteErrorTol = rp->absolute_error_tol;
  params.errorTol = rp->error_tol;
  params.lambdaInitial = rp->lambda_initial;
  params.lambdaFactor = rp->lambda_factor;
  params.lambdaUpperBound = rp->lambda_upper_bound;
  params.lambdaLowerBound = rp->lambda_lower_bound;
  params.minModelFidelity = rp->min_model_fidelity;
  params.diagonalDamping = rp->diagonal_damping;
  params.useFixedLambdaFactor = rp->use_fixed_lambda_factor;
  params.minDiagonal = rp->min_diagonal;
  params.maxDiagonal = rp->max_diagonal;
  const Values initial = g->unpack(values0);
  auto t0 = std::chrono::high_resolution_clock::now();
  if (rp->ordering_kind == 1) params.ordering = g->ordering(1);
  LevenbergMarquardtOptimizer lm(g->graph, initial, params);
  int nt = 0;
  auto rec = [&]() {
    if (nt < max_trace) {
      auto t = std::chrono::high_resolution_clock::now();
      trace[4 * nt + 0] = lm.getInnerIterations();
      trace[4 * nt + 1] = lm.error();
      trace[4 * nt + 2] = lm.lambda();
      trace[4 * nt + 3] = std::chrono::duration<double>(t - t0).count();
      nt++;
    }
  };
  rec();
  // NonlinearOptimizer::defaultOptimize, nonlinear/NonlinearOptimizer.cpp:62-117
  double currentError = lm.error();
  if (!(currentError <= params.errorTol) && lm.iterations() < (size_t)params.maxIterations) {
    double newError = currentError;
    do {
      currentError = newError;
      lm.iterate();
      newError = lm.error();
      rec();
    } while (lm.iterations() < (size_t)params.maxIterations &&
             !checkConvergence(params.relativeErrorTol, params.absoluteErrorTol, params.errorTol,
                               currentError, newError, params.verbosity) &&
             std::isfinite(currentError));
  }
  auto t1 = std::chrono::high_resolution_clock::now();
  if (total_seconds) *total_seconds = std::chrono::duration<double>(t1 - t0).count();
  if (n_trace) *n_trace = nt;
  if (values_out) g->pack(lm.values(), values_out);
  return (int)lm.iterations();
}

// Phase timings of ONE LM iteration made of the reference's own calls in the order iterate()/
// tryLambda() make them (LevenbergMarquardtOptimizer.cpp:121-308). ms[8]: linearize,
// hessianDiagonal, damp, eliminate+solve, linear error x2, retract, nonlinear error, total.
// res (optional, 6 doubles): graph.error(values), linear error at 0 and at delta, graph.error(retract(values, delta)),
// |delta|_2, |delta|_inf -- what LM's accept / reject decision is made from (LevenbergMarquardtOptimizer.cpp:180-235); the
// bench line compares the device's numbers of the same lambda try with them (parity_vs_reference).
int ref_graph_iteration_phases2(void* h, const double* values, double lambda, int diagonal_damping,
                                int ordering_kind, double* ms, double* res) {
  RefGraph* g = static_cast<RefGraph*>(h);
  using clk = std::chrono::high_resolution_clock;
  auto msec = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const Values vals = g->unpack(values);
  const Ordering ord = g->ordering(ordering_kind);
  auto t0 = clk::now();
  auto lin = g->graph.linearize(vals);
  auto t1 = clk::now();
  VectorValues sq;
  if (diagonal_damping) {
    sq = lin->hessianDiagonal();
    for (auto& [key, value] : sq) value = value.cwiseMax(1e-6).cwiseMin(1e32).cwiseSqrt();
  }
  auto t2 = clk::now();
  internal::LevenbergMarquardtState state(vals, 0.0, lambda, 10.0);
  GaussianFactorGraph damped = diagonal_damping ? state.buildDampedSystem(*lin, sq) : state.buildDampedSystem(*lin);
  auto t3 = clk::now();
  VectorValues d;
  int status = 0;
  try { d = damped.optimize(ord, EliminatePreferCholesky); } catch (const IndeterminantLinearSystemException&) { status = 1; }
  auto t4 = clk::now();
  double e = 0, l0 = 0, l1 = 0, e1 = 0;
  if (!status) { l0 = lin->error(VectorValues::Zero(d)); l1 = lin->error(d); e = l0 + l1; }
  auto t5 = clk::now();
  Values nv; if (!status) nv = vals.retract(d);
  auto t6 = clk::now();
  if (!status) { e1 = g->graph.error(nv); e += e1; }
  auto t7 = clk::now();
  ms[0] = msec(t0, t1); ms[1] = msec(t1, t2); ms[2] = msec(t2, t3); ms[3] = msec(t3, t4);
  ms[4] = msec(t4, t5); ms[5] = msec(t5, t6); ms[6] = msec(t6, t7); ms[7] = msec(t0, t7);
  if (res) {
    res[0] = g->graph.error(vals); res[1] = l0; res[2] = l1; res[3] = e1;
    res[4] = status ? 0.0 : d.norm(); res[5] = 0.0;
    if (!status) for (const auto& [key, value] : d) res[5] = std::max(res[5], value.cwiseAbs().maxCoeff());
  }
  return status + (e != e ? 2 : 0);
}
// The same iteration with the work the reference's TBB build would spread over cores split over `n_threads` std::threads HERE,
// in the harness (the image has no TBB headers, so the library itself is built single-threaded): a multi-thread figure next
// to the one-thread one, made of the reference's own calls --
//   linearize:  factor->linearize(values) over contiguous chunks of the graph (what NonlinearFactorGraph::linearize does under
//               TBB, nonlinear/NonlinearFactorGraph.cpp:_LinearizeOneFactor);
//   eliminate:  the landmarks of the Schur ordering in n_threads groups, each group's factors through
//               eliminatePartialSequential on its own thread (the leaves of the elimination tree, which TBB runs in parallel),
//               then the remaining camera / pose system on ONE thread (a chain of dense fronts: no tree parallelism to use),
//               then the groups' back-substitution in parallel;
//   errors and retract as in the one-thread variant (1 % of the time).
// ms[5]: linearize, eliminate landmarks, eliminate + solve the rest, back-substitute, total.  res as in ..._phases2.
int ref_graph_iteration_mt(void* h, const double* values, double lambda, int diagonal_damping, int n_threads, double* ms, double* res) {
  RefGraph* g = static_cast<RefGraph*>(h);
  using clk = std::chrono::high_resolution_clock;
  auto msec = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const Values vals = g->unpack(values);
  const size_t nf = g->graph.size();
  const int nth = std::max(1, n_threads);
  auto run = [&](const std::function<void(int)>& body) {
    std::vector<std::thread> th;
    for (int t = 1; t < nth; t++) th.emplace_back(body, t);
    body(0);
    for (auto& x : th) x.join();
  };
  auto t0 = clk::now();
  GaussianFactorGraph lin;
  lin.resize(nf);
  run([&](int t) { for (size_t i = nf * t / nth; i < nf * (t + 1) / nth; i++) lin.at(i) = g->graph[i]->linearize(vals); });
  auto t1 = clk::now();
  VectorValues sq;
  if (diagonal_damping) {
    sq = lin.hessianDiagonal();
    for (auto& [key, value] : sq) value = value.cwiseMax(1e-6).cwiseMin(1e32).cwiseSqrt();
  }
  internal::LevenbergMarquardtState state(vals, 0.0, lambda, 10.0);
  GaussianFactorGraph damped = diagonal_damping ? state.buildDampedSystem(lin, sq) : state.buildDampedSystem(lin);
  // groups of landmarks (contiguous in the Schur ordering) and the factors that touch them
  std::vector<int> group_of(g->n_vars, -1);
  std::vector<Ordering> group_keys(nth);
  Ordering rest_keys;
  { int64_t n_pt = 0, k = 0;
    for (int i = 0; i < g->n_vars; i++) if (g->var_type[i] == GTG_VAR_POINT3) n_pt++;
    for (int i = 0; i < g->n_vars; i++) {
      if (g->var_type[i] == GTG_VAR_POINT3) { const int t = (int)(k++ * nth / std::max<int64_t>(n_pt, 1)); group_of[i] = t; group_keys[t].push_back(Key(i)); }
      else rest_keys.push_back(Key(i));
    } }
  std::vector<GaussianFactorGraph> sub(nth);
  GaussianFactorGraph rest;
  for (const auto& f : damped) {
    int t = -1;
    for (Key k : f->keys()) if (group_of[k] >= 0) { t = group_of[k]; break; }
    if (t >= 0) sub[t].push_back(f); else rest.push_back(f);
  }
  auto t2 = clk::now();
  std::vector<std::shared_ptr<GaussianBayesNet>> bn(nth);
  std::vector<std::shared_ptr<GaussianFactorGraph>> rem(nth);
  std::vector<int> bad(nth, 0);
  run([&](int t) {
    if (group_keys[t].empty()) return;
    try { auto r = sub[t].eliminatePartialSequential(group_keys[t], EliminatePreferCholesky); bn[t] = r.first; rem[t] = r.second; }
    catch (const IndeterminantLinearSystemException&) { bad[t] = 1; }
  });
  auto t3 = clk::now();
  int status = 0;
  for (int t = 0; t < nth; t++) { if (bad[t]) status = 1; else if (rem[t]) rest.push_back(*rem[t]); }
  VectorValues d;
  if (!status) { try { d = rest.optimize(rest_keys, EliminatePreferCholesky); } catch (const IndeterminantLinearSystemException&) { status = 1; } }
  auto t4 = clk::now();
  if (!status) {
    std::vector<VectorValues> part(nth);
    run([&](int t) { if (bn[t]) part[t] = bn[t]->optimize(d); });
    for (int t = 0; t < nth; t++) for (Key k : group_keys[t]) d.insert(k, part[t].at(k));
  }
  auto t5 = clk::now();
  double l0 = 0, l1 = 0, e1 = 0;
  if (!status) { l0 = lin.error(VectorValues::Zero(d)); l1 = lin.error(d); e1 = g->graph.error(vals.retract(d)); }
  auto t6 = clk::now();
  ms[0] = msec(t0, t1); ms[1] = msec(t2, t3); ms[2] = msec(t3, t4); ms[3] = msec(t4, t5); ms[4] = msec(t0, t6);
  if (res) {
    res[0] = g->graph.error(vals); res[1] = l0; res[2] = l1; res[3] = e1;
    res[4] = status ? 0.0 : d.norm(); res[5] = 0.0;
    if (!status) for (const auto& [key, value] : d) res[5] = std::max(res[5], value.cwiseAbs().maxCoeff());
  }
  return status;
}
// gtsam::triangulateSafe (geometry/triangulation.h:697-752) for m PinholeCamera<Cal3Bundler> cameras (17 doubles each):
// status 0 VALID, 1 DEGENERATE, 2 BEHIND_CAMERA, 3 OUTLIER, 4 FAR_POINT; point filled when valid
// enable_epi: TriangulationParameters::enableEPI -- the DLT point refined by triangulateNonlinear (triangulation.h:211-221: LM on
// TriangulationFactors, triangulation.cpp:177-195) before the checks
int ref_triangulate_safe_epi(int m, const double* cams17, const double* z, double rank_tol, double dist_thr, double outlier_thr, int enable_epi, double* point);
int ref_triangulate_safe(int m, const double* cams17, const double* z, double rank_tol, double dist_thr, double outlier_thr, double* point) {
  return ref_triangulate_safe_epi(m, cams17, z, rank_tol, dist_thr, outlier_thr, 0, point);
}
int ref_triangulate_safe_epi(int m, const double* cams17, const double* z, double rank_tol, double dist_thr, double outlier_thr, int enable_epi, double* point) {
  CameraSet<Camera> cameras;
  Point2Vector measured;
  for (int k = 0; k < m; k++) { cameras.push_back(unpackCamera(cams17 + 17 * k)); measured.emplace_back(z[2 * k], z[2 * k + 1]); }
  TriangulationParameters params(rank_tol, enable_epi != 0, dist_thr, outlier_thr);
  // (TriangulationFactor::linearize projects without the try / catch of its evaluateError, slam/TriangulationFactor.h:148-170: a
  // refinement that linearises at a point behind one of the cameras throws a CheiralityException through triangulateSafe -> 6)
  TriangulationResult r = TriangulationResult::Degenerate();
  try { r = triangulateSafe(cameras, measured, params); } catch (const CheiralityException&) { return 6; }
  if (r.valid()) { point[0] = r->x(); point[1] = r->y(); point[2] = r->z(); return 0; }
  return r.degenerate() ? 1 : r.behindCamera() ? 2 : r.outlier() ? 3 : 4;
}
int ref_graph_iteration_phases(void* h, const double* values, double lambda, int diagonal_damping,
                               int ordering_kind, double* ms) {
  return ref_graph_iteration_phases2(h, values, lambda, diagonal_damping, ordering_kind, ms, nullptr);
}

// ---- dataset loaders (reference's own parsers) -------------------------------------------------
static SfmData g_bal;
int ref_load_bal(const char* path, int64_t* n_cam, int64_t* n_pt, int64_t* n_obs) {
  g_bal = SfmData::FromBalFile(path);  // sfm/SfmData.cpp:189-246 (parses through float)
  *n_cam = g_bal.numberCameras(); *n_pt = g_bal.numberTracks();
  int64_t k = 0; for (const SfmTrack& t : g_bal.tracks) k += t.numberMeasurements();
  *n_obs = k;
  return 0;
}
int ref_bal_fill(double* cams17, double* pts3, int32_t* obs_cam, int32_t* obs_pt, double* obs_z) {
  for (size_t i = 0; i < g_bal.numberCameras(); i++) packCamera(g_bal.cameras[i], cams17 + 17 * i);
  int64_t k = 0;
  for (size_t j = 0; j < g_bal.numberTracks(); j++) {
    const SfmTrack& t = g_bal.tracks[j];
    pts3[3 * j] = t.p.x(); pts3[3 * j + 1] = t.p.y(); pts3[3 * j + 2] = t.p.z();
    for (const SfmMeasurement& m : t.measurements) {
      obs_cam[k] = (int32_t)m.first; obs_pt[k] = (int32_t)j;
      obs_z[2 * k] = m.second.x(); obs_z[2 * k + 1] = m.second.y();
      k++;
    }
  }
  return 0;
}

// writeBAL of what ref_load_bal loaded (sfm/SfmData.cpp:249-327)
int ref_write_bal(const char* path) { return writeBAL(path, g_bal) ? 0 : 1; }

// 3D pose graph via readG2o(file, is3D=true) (slam/dataset.cpp:621-633, load3D :922-944).
static NonlinearFactorGraph::shared_ptr g_pg;
static Values::shared_ptr g_pg_init;
int ref_load_g2o3d(const char* path, int64_t* n_between, int64_t* n_vertices) {
  auto gv = readG2o(path, true);
  g_pg = gv.first; g_pg_init = gv.second;
  int64_t nb = 0;
  for (const auto& f : *g_pg) if (std::dynamic_pointer_cast<BetweenFactor<Pose3>>(f)) nb++;
  *n_between = nb; *n_vertices = g_pg_init->size();
  return 0;
}
// noise out: kind per factor + 36 doubles (sigma / sigmas / R row-major)
int ref_g2o3d_fill(int64_t* v1, int64_t* v2, double* z12, int32_t* noise_kind, double* noise36,
                   int64_t* vertex_keys, double* vertex_poses12) {
  int64_t k = 0;
  for (const auto& f : *g_pg) {
    auto bf = std::dynamic_pointer_cast<BetweenFactor<Pose3>>(f);
    if (!bf) continue;
    v1[k] = (int64_t)bf->key1(); v2[k] = (int64_t)bf->key2();
    packPose(bf->measured(), z12 + 12 * k);
    double* nd = noise36 + 36 * k;
    std::fill(nd, nd + 36, 0.0);
    auto nm = bf->noiseModel();
    if (nm->isUnit()) noise_kind[k] = GTG_NOISE_UNIT;
    else if (auto iso = std::dynamic_pointer_cast<noiseModel::Isotropic>(nm)) { noise_kind[k] = GTG_NOISE_ISOTROPIC; nd[0] = iso->sigma(); }
    else if (auto dg = std::dynamic_pointer_cast<noiseModel::Diagonal>(nm)) { noise_kind[k] = GTG_NOISE_DIAGONAL; for (int i = 0; i < 6; i++) nd[i] = dg->sigma(i); }
    else { auto ga = std::dynamic_pointer_cast<noiseModel::Gaussian>(nm); noise_kind[k] = GTG_NOISE_GAUSSIAN; const Matrix R = ga->R(); for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) nd[6 * i + j] = R(i, j); }
    k++;
  }
  int64_t i = 0;
  for (const auto& kv : *g_pg_init) {
    vertex_keys[i] = (int64_t)kv.key;
    packPose(kv.value.cast<Pose3>(), vertex_poses12 + 12 * i);
    i++;
  }
  return 0;
}

// writeG2o (slam/dataset.cpp:636-735) of the graph + initial estimate that ref_load_g2o3d / ref_load_2d loaded
int ref_write_g2o3d(const char* path) { writeG2o(*g_pg, *g_pg_init, path); return 0; }
int ref_write_g2o2d(const char* path);

// 2D pose graph via load2D (slam/dataset.cpp:208-330: VERTEX2 / VERTEX_SE2, EDGE2 / EDGE_SE2 / ODOMETRY).
static NonlinearFactorGraph::shared_ptr g_pg2;
static Values::shared_ptr g_pg2_init;
int ref_load_2d(const char* path, int64_t* n_between, int64_t* n_vertices) {
  auto gv = load2D(path);
  g_pg2 = gv.first; g_pg2_init = gv.second;
  int64_t nb = 0;
  for (const auto& f : *g_pg2) if (std::dynamic_pointer_cast<BetweenFactor<Pose2>>(f)) nb++;
  *n_between = nb; *n_vertices = g_pg2_init->size();
  return 0;
}
// noise out: kind per factor + 9 doubles (sigma / sigmas / R row-major)
int ref_2d_fill(int64_t* v1, int64_t* v2, double* z3, int32_t* noise_kind, double* noise9, int64_t* vertex_keys,
                double* vertex_poses3) {
  int64_t k = 0;
  for (const auto& f : *g_pg2) {
    auto bf = std::dynamic_pointer_cast<BetweenFactor<Pose2>>(f);
    if (!bf) continue;
    v1[k] = (int64_t)bf->key1(); v2[k] = (int64_t)bf->key2();
    z3[3 * k] = bf->measured().x(); z3[3 * k + 1] = bf->measured().y(); z3[3 * k + 2] = bf->measured().theta();
    double* nd = noise9 + 9 * k;
    std::fill(nd, nd + 9, 0.0);
    auto nm = bf->noiseModel();
    if (nm->isUnit()) noise_kind[k] = GTG_NOISE_UNIT;
    else if (auto iso = std::dynamic_pointer_cast<noiseModel::Isotropic>(nm)) { noise_kind[k] = GTG_NOISE_ISOTROPIC; nd[0] = iso->sigma(); }
    else if (auto dg = std::dynamic_pointer_cast<noiseModel::Diagonal>(nm)) { noise_kind[k] = GTG_NOISE_DIAGONAL; for (int i = 0; i < 3; i++) nd[i] = dg->sigma(i); }
    else { auto ga = std::dynamic_pointer_cast<noiseModel::Gaussian>(nm); noise_kind[k] = GTG_NOISE_GAUSSIAN; const Matrix R = ga->R(); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) nd[3 * i + j] = R(i, j); }
    k++;
  }
  int64_t i = 0;
  for (const auto& kv : *g_pg2_init) {
    vertex_keys[i] = (int64_t)kv.key;
    const Pose2 q = kv.value.cast<Pose2>();
    vertex_poses3[3 * i] = q.x(); vertex_poses3[3 * i + 1] = q.y(); vertex_poses3[3 * i + 2] = q.theta();
    i++;
  }
  return 0;
}

int ref_write_g2o2d(const char* path) { writeG2o(*g_pg2, *g_pg2_init, path); return 0; }

// ---- small-matrix probes used to pin the restated linear algebra --------------------------------
// gtsam::choleskyPartial (base/cholesky.cpp:107-158) on a col-major... we pass row-major symmetric.
bool ref_cholesky_partial(double* ABC, int n, int nFrontal);
}  // extern "C"

#include <gtsam/base/cholesky.h>
extern "C" bool ref_cholesky_partial(double* ABC, int n, int nFrontal) {
  Matrix M(n, n);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) M(i, j) = ABC[i * n + j];
  const bool ok = choleskyPartial(M, nFrontal);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) ABC[i * n + j] = M(i, j);
  return ok;
}
