// tests/hostmath/hostmath.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the per-factor device formulas (gtsam_amd/csrc/geom.h, factors.h) for the HOST with plain
// g++ so that `-m "not gpu"` tests can pin them against the oracle without a GPU.  This library is
// never loaded by the product (gtsam_amd/), which has no CPU path.
#include <vector>
#include "../../gtsam_amd/csrc/factors.h"
static gt::NoiseRef NR(int nk, const double* nd) { return gt::NoiseRef{nk, nd, 0, 0.0}; }
extern "C" {
void hm_sfm_linearize(long n, const double* cam, const double* pt, const double* z, int nk, const double* nd, double* J) {
  for (long i = 0; i < n; i++) gt::sfm_linearize(cam + 17 * i, pt + 3 * i, z + 2 * i, NR(nk, nd), J + gt::kSfmRec * i);
}
void hm_sfm_error(long n, const double* cam, const double* pt, const double* z, int nk, const double* nd, double* e) {
  for (long i = 0; i < n; i++) e[i] = gt::sfm_error(cam + 17 * i, pt + 3 * i, z + 2 * i, NR(nk, nd));
}
void hm_proj_linearize(long n, const double* pose, const double* K, const double* sensor, const double* pt, const double* z, int nk, const double* nd, double* J) {
  for (long i = 0; i < n; i++) gt::proj_linearize(pose + 12 * i, K, sensor, pt + 3 * i, z + 2 * i, NR(nk, nd), J + gt::kProjRec * i);
}
void hm_proj_error(long n, const double* pose, const double* K, const double* sensor, const double* pt, const double* z, int nk, const double* nd, double* e) {
  for (long i = 0; i < n; i++) e[i] = gt::proj_error(pose + 12 * i, K, sensor, pt + 3 * i, z + 2 * i, NR(nk, nd));
}
void hm_between_linearize(long n, const double* T1, const double* T2, const double* Z, int nk, const double* nd, double* J) {
  for (long i = 0; i < n; i++) gt::between_linearize(T1 + 12 * i, T2 + 12 * i, Z + 12 * i, NR(nk, nd), J + gt::kBetweenRec * i);
}
void hm_between_error(long n, const double* T1, const double* T2, const double* Z, int nk, const double* nd, double* e) {
  for (long i = 0; i < n; i++) e[i] = gt::between_error(T1 + 12 * i, T2 + 12 * i, Z + 12 * i, NR(nk, nd));
}
void hm_between2_linearize(long n, const double* p1, const double* p2, const double* z, int nk, const double* nd, double* J) {
  for (long i = 0; i < n; i++) gt::between2_linearize(p1 + 3 * i, p2 + 3 * i, z + 3 * i, NR(nk, nd), J + gt::kBetweenRec * i);
}
void hm_between2_error(long n, const double* p1, const double* p2, const double* z, int nk, const double* nd, double* e) {
  for (long i = 0; i < n; i++) e[i] = gt::between2_error(p1 + 3 * i, p2 + 3 * i, z + 3 * i, NR(nk, nd));
}
void hm_prior_linearize(int vtype, const double* x, const double* z, int nk, const double* nd, double* J) { gt::prior_linearize(vtype, x, z, NR(nk, nd), J); }
double hm_prior_error(int vtype, const double* x, const double* z, int nk, const double* nd) { return gt::prior_error(vtype, x, z, NR(nk, nd)); }
void hm_retract(int vtype, long n, const double* x, const double* d, double* y) {
  const int st = vtype == 0 ? 12 : vtype == 1 ? 17 : 3, dm = vtype == 0 ? 6 : vtype == 1 ? 9 : 3;
  for (long i = 0; i < n; i++) gt::value_retract(vtype, x + st * i, d + dm * i, y + st * i);
}
void hm_local(int vtype, long n, const double* x, const double* z, double* d) {
  const int st = vtype == 0 ? 12 : vtype == 1 ? 17 : 3, dm = vtype == 0 ? 6 : vtype == 1 ? 9 : 3;
  for (long i = 0; i < n; i++) gt::value_local(vtype, x + st * i, z + st * i, d + dm * i);
}
double hm_robust_weight(int rk, double k, double d) { return gt::robust_weight(rk, k, d); }
double hm_robust_loss(int rk, double k, double d) { return gt::robust_loss(rk, k, d); }
void hm_so3_logmap(long n, const double* R, double* w) { for (long i = 0; i < n; i++) gt::so3_logmap(R + 9 * i, w + 3 * i); }
void hm_so3_expmap(long n, const double* w, double* R) { for (long i = 0; i < n; i++) gt::so3_expmap(w + 3 * i, R + 9 * i); }
// a smart factor's point at infinity: direction of measurement z seen from cam0, then one record / error per camera
int hm_sfm_backproject_at_infinity(const double* cam, const double* z, double* dir) { return gt::sfm_backproject_at_infinity(cam, z, dir) ? 1 : 0; }
int hm_sfm_linearize_at_infinity(long n, const double* cam, const double* dir, const double* z, int nk, const double* nd, double* J, double* e) {
  int bad = 0;
  for (long i = 0; i < n; i++) {
    if (!gt::sfm_linearize_at_infinity(cam + 17 * i, dir + 3 * i, z + 2 * i, NR(nk, nd), J + gt::kSfmRec * i)) bad++;
    gt::sfm_error_at_infinity(cam + 17 * i, dir + 3 * i, z + 2 * i, NR(nk, nd), e + i);
  }
  return bad;
}
// smart factors: gtsam::triangulateSafe for m PinholeCamera<Cal3Bundler> cameras (17 doubles each) -> status, point
int hm_smart_triangulate_epi(int m, const double* cams17, const double* z, double rank_tol, double dist_thr, double outlier_thr, int enable_epi, double* point) {
  std::vector<int32_t> ids(m); std::vector<int64_t> off(m);
  for (int k = 0; k < m; k++) { ids[k] = k; off[k] = 17 * k; }
  return gt::smart_triangulate(m, ids.data(), off.data(), cams17, z, rank_tol, dist_thr, outlier_thr, point, enable_epi != 0);
}
int hm_smart_triangulate(int m, const double* cams17, const double* z, double rank_tol, double dist_thr, double outlier_thr, double* point) {
  return hm_smart_triangulate_epi(m, cams17, z, rank_tol, dist_thr, outlier_thr, 0, point);
}
}
