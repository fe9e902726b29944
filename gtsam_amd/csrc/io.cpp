// io.cpp -- native BAL reader / writer straight to and from the SoA layout the device path uploads
// (the wire format on the bundle-adjustment side of the hot path).
//
// Host-only code (no HIP): part of libgtsam_amd.so so that a caller of the C ABI gets from a file on disk to
// gtg_upload_problem() without a per-token interpreter loop (a Ladybug-1723 file has 2.7 M numbers, a Venice-1778 file
// 20 M).  The numbers are the ones the reference's own loader produces:
//   gtg_io_read_bal   <-> SfmData::FromBalFile  gtsam/sfm/SfmData.cpp:189-246  (every number is read into a `float`;
//                         pose = openGL2gtsam(Rodrigues(w), t) :79-85; measurement (u, -v); Cal3Bundler(f, k1, k2);
//                         tracks[j].measurements in file order  => observations grouped by point, stable)
//   gtg_io_write_bal  <-> writeBAL              gtsam/sfm/SfmData.cpp:249-327  (precision 20; gtsam2openGL :88-99;
//                         Rot3::Logmap = SO3::Logmap geometry/SO3.cpp:247-323; pixel (u - u0, -(v - v0)))
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/gtsam_amd.h"

namespace {

thread_local std::string io_error;

struct FileText {
  std::vector<char> buf;
  bool load(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize((size_t)std::max(0L, n) + 1);
    const size_t got = n > 0 ? std::fread(buf.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    buf[got] = 0;
    buf.resize(got + 1);
    return true;
  }
};

// whitespace-separated tokens, the way `is >> x` consumes them
struct Tokens {
  char* p;
  explicit Tokens(char* b) : p(b) {}
  bool next_ll(long long* v) {
    while (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t') p++;
    if (!*p) return false;
    char* e; errno = 0;
    *v = std::strtoll(p, &e, 10);
    if (e == p) return false;
    p = e; return true;
  }
  bool next_float(double* v) {   // `float x; is >> x`: the nearest binary32, then widened
    while (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t') p++;
    if (!*p) return false;
    char* e;
    const float f = std::strtof(p, &e);
    if (e == p) return false;
    p = e; *v = (double)f; return true;
  }
};

// Rot3::Rodrigues = SO3::Expmap (geometry/SO3.cpp:50-88): near zero I + W, else I + sin(t) K + 2 sin^2(t/2) K^2
void rodrigues(const double w[3], double R[9]) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (t2 <= std::numeric_limits<double>::epsilon()) { for (int i = 0; i < 9; i++) R[i] += W[i]; return; }
  const double t = std::sqrt(t2);
  double K[9], KK[9];
  for (int i = 0; i < 9; i++) K[i] = W[i] / t;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += K[3 * i + k] * K[3 * k + j]; KK[3 * i + j] = s; }
  const double s2 = std::sin(t / 2.0), st = std::sin(t), one_minus_cos = 2.0 * s2 * s2;
  for (int i = 0; i < 9; i++) R[i] += st * K[i] + one_minus_cos * KK[i];
}

// SO3::Logmap (geometry/SO3.cpp:247-323), the three branches of the reference
void so3_logmap(const double R[9], double w[3]) {
  const double R11 = R[0], R12 = R[1], R13 = R[2], R21 = R[3], R22 = R[4], R23 = R[5], R31 = R[6], R32 = R[7], R33 = R[8];
  const double tr = R11 + R22 + R33;
  if (tr + 1.0 < 1e-3) {   // angle close to pi
    // axis a = the largest diagonal entry (ties: 3 over 2 over 1), (b, c) the next two cyclically -- same arithmetic as geom.h
    double along_a, along_b, along_c, anti;
    int a;
    if (R33 > R22 && R33 > R11) { a = 2; anti = R21 - R12; along_a = 2.0 + 2.0 * R33; along_b = R31 + R13; along_c = R23 + R32; }
    else if (R22 > R11) { a = 1; anti = R13 - R31; along_a = 2.0 + 2.0 * R22; along_b = R23 + R32; along_c = R12 + R21; }
    else { a = 0; anti = R32 - R23; along_a = 2.0 + 2.0 * R11; along_b = R12 + R21; along_c = R31 + R13; }
    const double inv_root = 1 / std::sqrt(along_a);
    const double len = std::sqrt(along_a * along_a + along_b * along_b + along_c * along_c + anti * anti);
    const double sign = anti < 0 ? -1.0 : 1.0;
    const double k = 0.5 * inv_root * (M_PI - (2 * sign * anti) / len);
    const double wa = sign * k * along_a, wb = sign * k * along_b, wc = sign * k * along_c;
    if (a == 2) { w[0] = wb; w[1] = wc; w[2] = wa; }
    else if (a == 1) { w[0] = wc; w[1] = wa; w[2] = wb; }
    else { w[0] = wa; w[1] = wb; w[2] = wc; }
    return;
  }
  double magnitude;
  const double tr_3 = tr - 3.0;
  if (tr_3 < -1e-6) {
    const double theta = std::acos((tr - 1.0) / 2.0);
    magnitude = theta / (2.0 * std::sin(theta));
  } else {   // near zero: Taylor expansion of theta / (2 sin theta) in tr - 3
    magnitude = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0;
  }
  w[0] = magnitude * (R32 - R23); w[1] = magnitude * (R13 - R31); w[2] = magnitude * (R21 - R12);
}

void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = s; }
}

}  // namespace

namespace gtg_io { void set_error(const std::string& s) { io_error = s; } }   // (io_g2o.cpp reports through the same thread-local text)

extern "C" {

const char* gtg_io_last_error(void) { return io_error.c_str(); }

int gtg_io_bal_sizes(const char* path, int64_t* n_cams, int64_t* n_points, int64_t* n_obs) {
  if (!path || !n_cams || !n_points || !n_obs) { io_error = "null argument"; return GTG_ERR_USAGE; }
  FILE* f = std::fopen(path, "rb");
  if (!f) { io_error = "Error in FromBalFile: can not find the file!!"; return GTG_ERR_USAGE; }   // SfmData.cpp:192-194
  long long a = 0, b = 0, c = 0;
  const int got = std::fscanf(f, "%lld %lld %lld", &a, &b, &c);
  std::fclose(f);
  if (got != 3 || a < 0 || b < 0 || c < 0) { io_error = "BAL header: expected <cameras> <points> <observations>"; return GTG_ERR_USAGE; }
  *n_cams = a; *n_points = b; *n_obs = c;
  return GTG_OK;
}

int gtg_io_read_bal(const char* path, int64_t n_cams, int64_t n_points, int64_t n_obs, double* cams17, double* points3,
                    int32_t* obs_cam, int32_t* obs_point, double* obs_z) {
  if (!path || !cams17 || !points3 || !obs_cam || !obs_point || !obs_z) { io_error = "null argument"; return GTG_ERR_USAGE; }
  FileText ft;
  if (!ft.load(path)) { io_error = "Error in FromBalFile: can not find the file!!"; return GTG_ERR_USAGE; }
  Tokens tk(ft.buf.data());
  long long a, b, c;
  if (!tk.next_ll(&a) || !tk.next_ll(&b) || !tk.next_ll(&c) || a != n_cams || b != n_points || c != n_obs) {
    io_error = "BAL header does not match the sizes passed in (call gtg_io_bal_sizes first)"; return GTG_ERR_USAGE;
  }
  // observations in file order, then grouped by point (tracks[j].measurements.emplace_back: stable)
  std::vector<int32_t> fc((size_t)n_obs), fp((size_t)n_obs);
  std::vector<double> fz(2 * (size_t)n_obs);
  std::vector<int64_t> track_ptr((size_t)n_points + 1, 0);
  for (int64_t k = 0; k < n_obs; k++) {
    long long i, j; double u, v;
    if (!tk.next_ll(&i) || !tk.next_ll(&j) || !tk.next_float(&u) || !tk.next_float(&v)) { io_error = "BAL file: truncated observation block"; return GTG_ERR_USAGE; }
    if (i < 0 || i >= n_cams || j < 0 || j >= n_points) { io_error = "BAL file: observation refers to a camera / point that does not exist"; return GTG_ERR_USAGE; }
    fc[k] = (int32_t)i; fp[k] = (int32_t)j; fz[2 * k] = u; fz[2 * k + 1] = -v;
    track_ptr[j + 1]++;
  }
  for (int64_t j = 0; j < n_points; j++) track_ptr[j + 1] += track_ptr[j];
  {
    std::vector<int64_t> w(track_ptr.begin(), track_ptr.end() - 1);
    for (int64_t k = 0; k < n_obs; k++) {
      const int64_t d = w[fp[k]]++;
      obs_cam[d] = fc[k]; obs_point[d] = fp[k]; obs_z[2 * d] = fz[2 * k]; obs_z[2 * d + 1] = fz[2 * k + 1];
    }
  }
  for (int64_t i = 0; i < n_cams; i++) {
    double v[9];
    for (int k = 0; k < 9; k++) if (!tk.next_float(&v[k])) { io_error = "BAL file: truncated camera block"; return GTG_ERR_USAGE; }
    double R[9];
    rodrigues(v, R);
    // openGL2gtsam: wRc = R^T * diag(1, -1, -1);  wTc = R^T * (-t)
    double* o = cams17 + 17 * i;
    for (int r = 0; r < 3; r++) { o[3 * r + 0] = R[3 * 0 + r]; o[3 * r + 1] = -R[3 * 1 + r]; o[3 * r + 2] = -R[3 * 2 + r]; }
    for (int r = 0; r < 3; r++) o[9 + r] = R[3 * 0 + r] * (-v[3]) + R[3 * 1 + r] * (-v[4]) + R[3 * 2 + r] * (-v[5]);
    o[12] = v[6]; o[13] = v[7]; o[14] = v[8]; o[15] = 0.0; o[16] = 0.0;
  }
  for (int64_t j = 0; j < 3 * n_points; j++)
    if (!tk.next_float(&points3[j])) { io_error = "BAL file: truncated point block"; return GTG_ERR_USAGE; }
  return GTG_OK;
}

int gtg_io_write_bal(const char* path, int64_t n_cams, int64_t n_points, int64_t n_obs, const double* cams17, const double* points3,
                     const int32_t* obs_cam, const int32_t* obs_point, const double* obs_z) {
  if (!path || !cams17 || !points3 || (n_obs && (!obs_cam || !obs_point || !obs_z))) { io_error = "null argument"; return GTG_ERR_USAGE; }
  for (int64_t k = 1; k < n_obs; k++)
    if (obs_point[k] < obs_point[k - 1]) { io_error = "writeBAL: observations must be grouped by point (track order)"; return GTG_ERR_USAGE; }
  FILE* f = std::fopen(path, "w");
  if (!f) { io_error = "Error in writeBAL: can not open the file!!"; return GTG_ERR_USAGE; }   // SfmData.cpp:254-257
  std::fprintf(f, "%lld %lld %lld\n\n", (long long)n_cams, (long long)n_points, (long long)n_obs);
  for (int64_t k = 0; k < n_obs; k++) {
    const double* c = cams17 + 17 * (int64_t)obs_cam[k];
    const double px = obs_z[2 * k] - c[15], py = -(obs_z[2 * k + 1] - c[16]);
    std::fprintf(f, "%d %d %.20g %.20g\n", obs_cam[k], obs_point[k], px, py);
  }
  std::fprintf(f, "\n");
  for (int64_t i = 0; i < n_cams; i++) {
    const double* c = cams17 + 17 * i;
    // gtsam2openGL: cRw = diag(1,-1,-1) * R^T ; t = cRw * (-t_gtsam)
    double Rt[9], cRw[9];
    for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++) Rt[3 * r + q] = c[3 * q + r];
    const double R90[9] = {1, 0, 0, 0, -1, 0, 0, 0, -1};
    mat3_mul(R90, Rt, cRw);
    double t[3], w[3];
    for (int r = 0; r < 3; r++) t[r] = cRw[3 * r] * (-c[9]) + cRw[3 * r + 1] * (-c[10]) + cRw[3 * r + 2] * (-c[11]);
    so3_logmap(cRw, w);
    std::fprintf(f, "%.20g\n%.20g\n%.20g\n%.20g\n%.20g\n%.20g\n%.20g\n%.20g\n%.20g\n\n", w[0], w[1], w[2], t[0], t[1], t[2], c[12], c[13], c[14]);
  }
  for (int64_t j = 0; j < n_points; j++) std::fprintf(f, "%.20g\n%.20g\n%.20g\n\n", points3[3 * j], points3[3 * j + 1], points3[3 * j + 2]);
  const bool ok = std::fclose(f) == 0;
  if (!ok) { io_error = "writeBAL: write failed"; return GTG_ERR_HIP; }
  return GTG_OK;
}

}  // extern "C"
