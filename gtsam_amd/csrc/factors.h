// factors.h -- one-factor evaluators: whitened Jacobian blocks + rhs, and the factor's error.
//
// Each function is what ONE lane executes for ONE factor.  They restate, for the GPU,
//   GeneralSFMFactor::linearize / evaluateError      slam/GeneralSFMFactor.h:127-177
//   GenericProjectionFactor::evaluateError           slam/ProjectionFactor.h:138-166
//   BetweenFactor<Pose3>::evaluateError              slam/BetweenFactor.h:111-124
//   PriorFactor<T>::evaluateError                    nonlinear/PriorFactor.h:98-102
// followed by NoiseModelFactor::linearize's  b = -r, WhitenSystem(A, b)
// (nonlinear/NonlinearFactor.cpp:150-182) and NoiseModelFactor::error = 0.5*||whiten(r)||^2
// (NonlinearFactor.cpp:136-147).
#pragma once
#include "geom.h"

namespace gt {

// A noise-table entry as the factor evaluators see it: base model + optional m-estimator.
struct NoiseRef {
  int kind; const double* data;   // GTG_NOISE_*, device data (inverse sigmas / R)
  int rkind; double rk;           // GTG_ROBUST_* (0 = none) and its parameter
};

// m-estimators (linear/LossFunctions.cpp): weight(distance) and loss(distance), distance = ||whitened residual||
GT_HD double robust_weight(int rkind, double k, double d) {
  const double a = fabs(d);
  switch (rkind) {
    case 1: return 1.0 / (1.0 + a / k);                                    // Fair   :146-148
    case 2: return (a <= k) ? 1.0 : (k / a);                               // Huber  :179-182
    case 3: return (k * k) / (k * k + d * d);                              // Cauchy :217-219
    case 4: { if (a <= k) { const double t = 1.0 - d * d / (k * k); return t * t; } return 0.0; }   // Tukey :250-256
    case 5: return exp(-(d * d) / (k * k));                                // Welsch :289-292
    case 6: { const double c2 = k * k, c4 = c2 * c2, ce = c2 + d * d; return c4 / (ce * ce); }     // Geman-McClure :320-325
    case 7: { const double e2 = d * d; if (e2 > k) { const double w = 2.0 * k / (k + e2); return w * w; } return 1.0; }   // DCS :355-364
    case 8: return (a <= k) ? 0.0 : (-k + a) / a;                          // L2WithDeadZone :402-409 (distance = a norm: >= 0)
    default: return 1.0;
  }
}
GT_HD double robust_loss(int rkind, double k, double d) {
  const double a = fabs(d);
  switch (rkind) {
    case 1: { const double ne = a / k; return k * k * (ne - log1p(ne)); }                            // :150-155
    case 2: return (a <= k) ? d * d / 2 : k * (a - (k / 2));                                         // :184-191
    case 3: return k * k * log1p(d * d / (k * k)) * 0.5;                                             // :221-224
    case 4: { if (a <= k) { const double t = 1.0 - d * d / (k * k); return k * k * (1 - t * t * t) / 6.0; } return k * k / 6.0; }  // :258-267
    case 5: return k * k * 0.5 * -expm1(-(d * d) / (k * k));                                         // :294-297
    case 6: { const double c2 = k * k, e2 = d * d; return 0.5 * (c2 * e2) / (c2 + e2); }             // :327-331
    case 7: { const double e2 = d * d, e4 = e2 * e2, c2 = k * k; return (c2 * e2 + k * e4) / ((e2 + k) * (e2 + k)); }   // DCS :366-375
    case 8: return (a < k) ? 0.0 : 0.5 * (k - a) * (k - a);                                          // L2WithDeadZone :411-414
    default: return 0.5 * d * d;
  }
}
// NoiseModelFactor::error: noiseModel_->loss(squaredMahalanobisDistance(r)) (NonlinearFactor.cpp:136-147);
// plain models 0.5 d^2, Robust robust_->loss(sqrt(d^2)) (NoiseModel.h:716-718)
GT_HD double factor_loss(const NoiseRef& n, double sq) { return n.rkind ? robust_loss(n.rkind, n.rk, sqrt(sq)) : 0.5 * sq; }
// Robust::WhitenSystem = noise_->WhitenSystem then robust_->reweight (Block scheme): scale by sqrt(weight(||b||))
GT_HD double reweight_factor(const NoiseRef& n, const double* b, int m) {
  if (!n.rkind) return 1.0;
  double s = 0.0;
  for (int i = 0; i < m; i++) s += b[i] * b[i];
  return sqrt(robust_weight(n.rkind, n.rk, sqrt(s)));
}

// row lengths of the per-factor Jacobian records (doubles)
constexpr int kSfmRec = 2 * 9 + 2 * 3 + 2;   // [A1 2x9 | A2 2x3 | b 2]
constexpr int kProjRec = 2 * 6 + 2 * 3 + 2;  // [A1 2x6 | A2 2x3 | b 2]
constexpr int kBetweenRec = 36 + 36 + 6;     // [A1 6x6 | A2 6x6 | b 6]
constexpr int kPriorRec = 81 + 9;            // [A dxd packed | pad ... | b d at 81]

// ---- GeneralSFMFactor<PinholeCamera<Cal3Bundler>,Point3> ---------------------------------------
// NOTE (reference behaviour, reproduced on purpose): GeneralSFMFactor::linearize whitens H1, H2 and b through
// noiseModel->Whiten(Matrix) separately (GeneralSFMFactor.h:162-168); for a Robust model that path re-weights with
// an EMPTY error vector, i.e. weight 1 -- the m-estimator changes this factor's error(), not its linear system.
GT_HD void sfm_linearize(const double* cam, const double* pt, const double* z, const NoiseRef& n, double* J) {
  const int nkind = n.kind; const double* nd = n.data;
  double pi[2];
  if (!sfm_project(cam, pt, pi, J, J + 18)) {  // CheiralityException: H1,H2,b = 0 (:153-158)
    for (int i = 0; i < kSfmRec; i++) J[i] = 0.0;
    return;
  }
  J[24] = z[0] - pi[0];
  J[25] = z[1] - pi[1];
  whiten_cols<2>(nkind, nd, J, 9);
  whiten_cols<2>(nkind, nd, J + 18, 3);
  whiten_cols<2>(nkind, nd, J + 24, 1);
}
GT_HD double sfm_error(const double* cam, const double* pt, const double* z, const NoiseRef& n) {
  double pi[2];
  if (!sfm_project(cam, pt, pi, nullptr, nullptr)) return factor_loss(n, 0.0);  // evaluateError returns Z_2x1 (:131-137)
  double r[2] = {pi[0] - z[0], pi[1] - z[1]};
  whiten_cols<2>(n.kind, n.data, r, 1);
  return factor_loss(n, r[0] * r[0] + r[1] * r[1]);
}

// One measurement of a smart factor whose landmark is a point at infinity (SmartFactorBase::computeJacobians<Unit3>,
// slam/SmartFactorBase.h:316-324, whitened as :356-363): the same record, the landmark block 2 x 2 padded with a zero column.
// false where the reference throws a CheiralityException (not caught by the smart factor).
GT_HD bool sfm_linearize_at_infinity(const double* cam, const double* dir, const double* z, const NoiseRef& n, double* J) {
  double pi[2];
  if (!sfm_project_at_infinity(cam, dir, pi, J, J + 18)) {
    for (int i = 0; i < kSfmRec; i++) J[i] = 0.0;
    return false;
  }
  J[24] = z[0] - pi[0];
  J[25] = z[1] - pi[1];
  whiten_cols<2>(n.kind, n.data, J, 9);
  whiten_cols<2>(n.kind, n.data, J + 18, 3);
  whiten_cols<2>(n.kind, n.data, J + 24, 1);
  // A Robust model here: the factor's own linearize() whitens H1, H2 and b ONE AT A TIME through Robust::Whiten(Matrix), whose
  // WhitenSystem sees an empty b -- the m-estimator's weight is evaluated at distance 0 (slam/GeneralSFMFactor.h:162-168,
  // linear/NoiseModel.h:711-714).  That is exactly 1 for every estimator but L2WithDeadZone, whose weight(0) is 0: the reference
  // drops the factor from the linear system (and keeps it in the error); so does this.
  if (n.rkind) { const double w = sqrt(robust_weight(n.rkind, n.rk, 0.0)); if (w != 1.0) for (int i = 0; i < kSfmRec; i++) J[i] *= w; }
  return true;
}
// its share of SmartFactorBase::totalReprojectionError<Unit3> (SmartFactorBase.h:296-303): 0.5 |whitened (h - z)|^2
GT_HD bool sfm_error_at_infinity(const double* cam, const double* dir, const double* z, const NoiseRef& n, double* e) {
  double pi[2];
  *e = 0.0;
  if (!sfm_project_at_infinity(cam, dir, pi, nullptr, nullptr)) return false;
  double r[2] = {pi[0] - z[0], pi[1] - z[1]};
  whiten_cols<2>(n.kind, n.data, r, 1);
  *e = 0.5 * (r[0] * r[0] + r[1] * r[1]);
  return true;
}

// ---- GenericProjectionFactor<Pose3,Point3,Cal3_S2> ---------------------------------------------
GT_HD void proj_linearize(const double* pose, const double* K, const double* sensor, const double* pt,
                          const double* z, const NoiseRef& n, double* J) {
  const int nkind = n.kind; const double* nd = n.data;
  double pi[2];
  bool ok;
  if (sensor) {  // pose.compose(body_P_sensor, H0); H1 = H1 * H0, H0 = sensor^-1 AdjointMap (Lie.h:56-61)
    double T[12], Dp[12], Si[12], H0[36];
    pose_compose(pose, sensor, T);
    ok = s2_project(T, K, pt, pi, Dp, J + 12);
    if (ok) {
      pose_inverse(sensor, Si);
      pose_adjoint(Si, H0);
      for (int r = 0; r < 2; r++)
        for (int c = 0; c < 6; c++) {
          double acc = 0.0;
          for (int k = 0; k < 6; k++) acc += Dp[6 * r + k] * H0[6 * k + c];
          J[6 * r + c] = acc;
        }
    }
  } else {
    ok = s2_project(pose, K, pt, pi, J, J + 12);
  }
  if (!ok) {  // H = 0 and residual 2*fx (ProjectionFactor.h:157-165, throwCheirality_ = false)
    for (int i = 0; i < 18; i++) J[i] = 0.0;
    J[18] = -2.0 * K[0]; J[19] = -2.0 * K[0];
  } else {
    J[18] = -(pi[0] - z[0]); J[19] = -(pi[1] - z[1]);
  }
  whiten_cols<2>(nkind, nd, J, 6);
  whiten_cols<2>(nkind, nd, J + 12, 3);
  whiten_cols<2>(nkind, nd, J + 18, 1);
  if (n.rkind) { const double w = reweight_factor(n, J + 18, 2); for (int i = 0; i < kProjRec; i++) J[i] *= w; }
}
GT_HD double proj_error(const double* pose, const double* K, const double* sensor, const double* pt,
                        const double* z, const NoiseRef& n) {
  double pi[2], r[2];
  bool ok;
  if (sensor) {
    double T[12];
    pose_compose(pose, sensor, T);
    ok = s2_project(T, K, pt, pi, nullptr, nullptr);
  } else {
    ok = s2_project(pose, K, pt, pi, nullptr, nullptr);
  }
  if (ok) { r[0] = pi[0] - z[0]; r[1] = pi[1] - z[1]; }
  else { r[0] = 2.0 * K[0]; r[1] = 2.0 * K[0]; }
  whiten_cols<2>(n.kind, n.data, r, 1);
  return factor_loss(n, r[0] * r[0] + r[1] * r[1]);
}

// ---- BetweenFactor<Pose3> ---------------------------------------------------------------------
// h = T1^-1 T2 ; H1 = -Ad(h^-1), H2 = I (Lie.h:63-69) ; r = Logmap(z^-1 h), Jacobians NOT
// multiplied by dLog (GTSAM_SLOW_BUT_CORRECT_BETWEENFACTOR is OFF by default, BetweenFactor.h:115-123)
GT_HD void between_residual(const double* T1, const double* T2, const double* Z, double* h, double* r) {
  pose_between(T1, T2, h);
  pose_local(Z, h, r);
}
GT_HD void between_linearize(const double* T1, const double* T2, const double* Z, const NoiseRef& n, double* J) {
  const int nkind = n.kind; const double* nd = n.data;
  double h[12], hi[12], r[6];
  between_residual(T1, T2, Z, h, r);
  pose_inverse(h, hi);
  pose_adjoint(hi, J);
  for (int i = 0; i < 36; i++) J[i] = -J[i];
  for (int i = 0; i < 36; i++) J[36 + i] = 0.0;
  for (int i = 0; i < 6; i++) J[36 + 7 * i] = 1.0;
  for (int i = 0; i < 6; i++) J[72 + i] = -r[i];
  whiten_cols<6>(nkind, nd, J, 6);
  whiten_cols<6>(nkind, nd, J + 36, 6);
  whiten_cols<6>(nkind, nd, J + 72, 1);
  if (n.rkind) { const double w = reweight_factor(n, J + 72, 6); for (int i = 0; i < kBetweenRec; i++) J[i] *= w; }
}
GT_HD double between_error(const double* T1, const double* T2, const double* Z, const NoiseRef& n) {
  double h[12], r[6];
  between_residual(T1, T2, Z, h, r);
  whiten_cols<6>(n.kind, n.data, r, 1);
  double e = 0.0;
  for (int i = 0; i < 6; i++) e += r[i] * r[i];
  return factor_loss(n, e);
}

// ---- BetweenFactor<Pose2> ---------------------------------------------------------------------
// Same generic LieGroup::between (Lie.h:63-69): h = p1^-1 p2, H1 = -Ad(h^-1) (Pose2.cpp:126-135), H2 = I;
// r = Local(z, h) = (x, y, theta) of z^-1 h, Jacobians not multiplied by the chart derivative (default flags).
// Record = the BetweenFactor<Pose3> record with 3x3 blocks: A1 at [0..8], A2 at [36..44], b at [72..74].
GT_HD void between2_residual(const double* p1, const double* p2, const double* z, double* h, double* r) {
  pose2_between_cs(p1[0], p1[1], cos(p1[2]), sin(p1[2]), p2[0], p2[1], cos(p2[2]), sin(p2[2]), h);
  double g[4];
  pose2_between_cs(z[0], z[1], cos(z[2]), sin(z[2]), h[0], h[1], h[2], h[3], g);
  r[0] = g[0]; r[1] = g[1]; r[2] = atan2(g[3], g[2]);
}
GT_HD void between2_linearize(const double* p1, const double* p2, const double* z, const NoiseRef& n, double* J) {
  double h[4], r[3];
  between2_residual(p1, p2, z, h, r);
  for (int i = 0; i < kBetweenRec; i++) J[i] = 0.0;
  // h^-1 = (c, -s, unrotate(-t_h)); Ad(pose) = [c -s y; s c -x; 0 0 1]
  const double c = h[2], s = -h[3];
  const double x = h[2] * (-h[0]) + h[3] * (-h[1]), y = -h[3] * (-h[0]) + h[2] * (-h[1]);
  J[0] = -c; J[1] = s; J[2] = -y;
  J[3] = -s; J[4] = -c; J[5] = x;
  J[6] = 0.0; J[7] = 0.0; J[8] = -1.0;
  J[36] = 1.0; J[40] = 1.0; J[44] = 1.0;
  for (int i = 0; i < 3; i++) J[72 + i] = -r[i];
  whiten_cols<3>(n.kind, n.data, J, 3);
  whiten_cols<3>(n.kind, n.data, J + 36, 3);
  whiten_cols<3>(n.kind, n.data, J + 72, 1);
  if (n.rkind) { const double w = reweight_factor(n, J + 72, 3); for (int i = 0; i < kBetweenRec; i++) J[i] *= w; }
}
GT_HD double between2_error(const double* p1, const double* p2, const double* z, const NoiseRef& n) {
  double h[4], r[3];
  between2_residual(p1, p2, z, h, r);
  whiten_cols<3>(n.kind, n.data, r, 1);
  return factor_loss(n, r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
}

// ---- PriorFactor<T> ---------------------------------------------------------------------------
// r = -Local(x, prior), H = I (approximate on purpose, PriorFactor.h:99).  d = tangent dim.
template <int D>
GT_HD void prior_linearize_d(int vtype, const double* x, const double* z, const NoiseRef& n, double* J) {
  const int nkind = n.kind; const double* nd = n.data;
  double loc[9];
  value_local(vtype, x, z, loc);
  for (int i = 0; i < kPriorRec; i++) J[i] = 0.0;
  for (int i = 0; i < D; i++) J[i * D + i] = 1.0;
  for (int i = 0; i < D; i++) J[81 + i] = loc[i];  // b = -r = Local(x, prior)
  whiten_cols<D>(nkind, nd, J, D);
  whiten_cols<D>(nkind, nd, J + 81, 1);
  if (n.rkind) { const double w = reweight_factor(n, J + 81, D); for (int i = 0; i < kPriorRec; i++) J[i] *= w; }
}
GT_HD void prior_linearize(int vtype, const double* x, const double* z, const NoiseRef& n, double* J) {
  if (vtype == 0) prior_linearize_d<6>(vtype, x, z, n, J);
  else if (vtype == 1) prior_linearize_d<9>(vtype, x, z, n, J);
  else prior_linearize_d<3>(vtype, x, z, n, J);
}
GT_HD double prior_error(int vtype, const double* x, const double* z, const NoiseRef& n) {
  double r[9];
  value_local(vtype, x, z, r);
  const int d = vtype == 0 ? 6 : vtype == 1 ? 9 : 3;
  if (d == 6) whiten_cols<6>(n.kind, n.data, r, 1);
  else if (d == 9) whiten_cols<9>(n.kind, n.data, r, 1);
  else whiten_cols<3>(n.kind, n.data, r, 1);
  double e = 0.0;
  for (int i = 0; i < d; i++) e += r[i] * r[i];
  return factor_loss(n, e);
}

}  // namespace gt
