// io_g2o.cpp -- native readers / writer for the pose-graph side of the hot path: g2o and TORO text files
// straight to and from the arrays gtg_problem's between / prior tables are filled from.  Host-only code (no HIP), part of
// libgtsam_amd.so next to io.cpp (BAL).  The numbers are the ones the reference's own loaders produce:
//   gtg_io_read_g2o  (3-D) <-> load3D / readG2o(file, true)  gtsam/slam/dataset.cpp:738-944 -- VERTEX3 / VERTEX_SE3:QUAT,
//                    EDGE3 / EDGE_SE3:QUAT (quaternion normalised :738-744; Rot3::Ypr for the TORO tags :748-753; the information
//                    matrix of an EDGE_SE3:QUAT line swapped from g2o's t,R block order into GTSAM's R,t :848-853); vertices the
//                    file does not list are NOT created (:922-944);
//   gtg_io_read_g2o  (2-D) <-> load2D / readG2o(file, false)  :505-570, 621-633 -- VERTEX2 / VERTEX_SE2 / VERTEX, EDGE2 / EDGE /
//                    EDGE_SE2 / ODOMETRY; the six noise numbers interpreted by createNoiseModel (:215-290: G2O / TORO = information,
//                    GRAPH / COV = covariance, AUTO guesses GRAPH or COV from the zero pattern); a vertex an edge refers to and the
//                    file does not list is created: identity for key1, key1's pose * measurement for key2 (:541-546);
//   noise models: noiseModel::Gaussian::Information / Covariance with smart = true (linear/NoiseModel.cpp:83-131): a diagonal matrix
//                    becomes Diagonal::Variances -> Isotropic when all equal -> Unit when |variance - 1| < 1e-9, anything else the
//                    upper Cholesky factor R of the information matrix;
//   gtg_io_write_g2o       <-> writeG2o  :636-735 -- VERTEX_SE2 / VERTEX_SE3:QUAT lines of the estimate, EDGE_SE2 / EDGE_SE3:QUAT lines
//                    with the upper triangle of R^T R (3-D: in g2o's t,R order), numbers as `stream << double` prints them (%g),
//                    quaternion = Eigen::Quaternion(Matrix3).
// Output layout = what gtsam_amd/io.py's restatement returns (pinned against the reference's loaders in tests/test_io.py):
//   poses: 3-D 12 doubles (R row-major 9, t 3), 2-D 3 doubles (x, y, theta = atan2(sin yaw, cos yaw)); noise parameters: 36 (3-D) / 9
//   (2-D) doubles per edge -- sigma | sigmas | R row-major -- behind a GTG_NOISE_* kind; vertices sorted by key.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/gtsam_amd.h"

extern "C" const char* gtg_io_last_error(void);
namespace gtg_io { void set_error(const std::string& s); }

namespace {

struct Line { const char* tag; std::vector<const char*> tok; };   // tag + the tokens behind it (pointers into the file's buffer)

// the file as lines of whitespace-separated tokens (parseLines: `is >> tag`, the parser's `>>`s, then the rest of the line is ignored)
bool load_lines(const char* path, std::vector<char>* buf, std::vector<Line>* lines) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  buf->resize((size_t)std::max(0L, n) + 1);
  const size_t got = n > 0 ? std::fread(buf->data(), 1, (size_t)n, f) : 0;
  std::fclose(f);
  (*buf)[got] = 0;
  char* p = buf->data();
  char* const end = p + got;
  while (p < end) {
    char* eol = static_cast<char*>(std::memchr(p, '\n', (size_t)(end - p)));
    if (!eol) eol = end;
    *eol = 0;
    Line ln{nullptr, {}};
    char* q = p;
    while (q < eol) {
      while (q < eol && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
      if (q >= eol) break;
      char* t = q;
      while (q < eol && !(*q == ' ' || *q == '\t' || *q == '\r')) q++;
      *q = 0; q++;
      if (!ln.tag) ln.tag = t; else ln.tok.push_back(t);
    }
    if (ln.tag) lines->push_back(std::move(ln));
    p = eol + 1;
  }
  return true;
}

bool num(const Line& ln, size_t i, double* v) {
  if (i >= ln.tok.size()) return false;
  char* e; *v = std::strtod(ln.tok[i], &e);
  return e != ln.tok[i];
}
bool idx(const Line& ln, size_t i, int64_t* v) {
  if (i >= ln.tok.size()) return false;
  char* e; const unsigned long long u = std::strtoull(ln.tok[i], &e, 10);
  *v = (int64_t)u;
  return e != ln.tok[i];
}
bool is(const Line& ln, const char* a) { return !std::strcmp(ln.tag, a); }

// Rot3::Ypr(y, p, r) = RzRyRx(r, p, y) (geometry/Rot3.h), row-major
void ypr(double yaw, double pitch, double roll, double R[9]) {
  const double cx = std::cos(roll), sx = std::sin(roll), cy = std::cos(pitch), sy = std::sin(pitch), cz = std::cos(yaw), sz = std::sin(yaw);
  const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  double T[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Rz[3 * i + k] * Ry[3 * k + j]; T[3 * i + j] = s; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += T[3 * i + k] * Rx[3 * k + j]; R[3 * i + j] = s; }
}

// operator>>(Quaternion) normalises (dataset.cpp:738-744); Eigen quaternion -> rotation matrix
void quat(double x, double y, double z, double w, double R[9]) {
  const double n = std::sqrt(w * w + x * x + y * y + z * z);
  x /= n; y /= n; z /= n; w /= n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

// Eigen::Quaternion(Matrix3) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl): -> (x, y, z, w)
void to_quaternion(const double R[9], double q[4]) {
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0); q[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (R[3 * k + j] - R[3 * j + k]) * t; q[j] = (R[3 * j + i] + R[3 * i + j]) * t; q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
}

// upper Cholesky factor of a symmetric positive definite d x d matrix (Eigen LLT::matrixU), row-major; false: not positive definite
bool chol_upper(const double* M, int d, double* U) {
  for (int i = 0; i < d * d; i++) U[i] = 0.0;
  for (int j = 0; j < d; j++) {
    for (int i = 0; i <= j; i++) {
      double s = M[d * i + j];
      for (int k = 0; k < i; k++) s -= U[d * k + i] * U[d * k + j];
      if (i == j) { if (!(s > 0)) return false; U[d * j + j] = std::sqrt(s); }
      else U[d * i + j] = s / U[d * i + i];
    }
  }
  return true;
}

// noiseModel::Gaussian::Information(M, smart = true) / Covariance(M, smart = true) -> (kind, parameters); `out` holds d * d doubles
bool smart_model(const double* M, int d, bool covariance, int32_t* kind, double* out) {
  for (int i = 0; i < d * d; i++) out[i] = 0.0;
  bool diagonal = true;
  for (int i = 0; i < d && diagonal; i++) for (int j = 0; j < d; j++) if (i != j && M[d * i + j] != 0.0) { diagonal = false; break; }
  if (diagonal) {
    double var[6]; bool equal = true;
    for (int i = 0; i < d; i++) { var[i] = covariance ? M[d * i + i] : 1.0 / M[d * i + i]; if (var[i] != var[0]) equal = false; }
    if (equal) {
      if (std::abs(var[0] - 1.0) < 1e-9) { *kind = GTG_NOISE_UNIT; return true; }
      out[0] = std::sqrt(var[0]); *kind = GTG_NOISE_ISOTROPIC; return true;
    }
    for (int i = 0; i < d; i++) out[i] = std::sqrt(var[i]);
    *kind = GTG_NOISE_DIAGONAL; return true;
  }
  double info[36];
  if (covariance) {   // Covariance(M): Information(M.inverse()) -- 3 x 3 only on this path (load2D), cofactor inverse like Eigen's
    if (d != 3) return false;
    const double a = M[0], b = M[1], c = M[2], e = M[4], f = M[5], i9 = M[8], d3 = M[3], g = M[6], h = M[7];
    const double c00 = e * i9 - f * h, c01 = -(d3 * i9 - f * g), c02 = d3 * h - e * g;
    const double det = a * c00 + b * c01 + c * c02;
    if (det == 0.0) return false;
    const double inv = 1.0 / det;
    info[0] = c00 * inv; info[1] = -(b * i9 - c * h) * inv; info[2] = (b * f - c * e) * inv;
    info[3] = c01 * inv; info[4] = (a * i9 - c * g) * inv; info[5] = -(a * f - c * d3) * inv;
    info[6] = c02 * inv; info[7] = -(a * h - b * g) * inv; info[8] = (a * e - b * d3) * inv;
  } else {
    for (int i = 0; i < d * d; i++) info[i] = M[i];
  }
  *kind = GTG_NOISE_GAUSSIAN;
  return chol_upper(info, d, out);
}

// createNoiseModel's matrix (dataset.cpp:215-262): the six numbers of a 2-D edge -> (3 x 3 matrix, is it a covariance); nullptr: fine
const char* noise_matrix3(const double v[6], int fmt, double M[9], bool* covariance) {
  if (fmt == GTG_IO_NOISE_AUTO) {
    if (v[0] != 0 && v[1] == 0 && v[2] != 0 && v[3] != 0 && v[4] == 0 && v[5] == 0) fmt = GTG_IO_NOISE_GRAPH;
    else if (v[0] != 0 && v[1] == 0 && v[2] == 0 && v[3] != 0 && v[4] == 0 && v[5] != 0) fmt = GTG_IO_NOISE_COV;
    else return "load2D: unrecognized covariance matrix format in dataset file. Please specify the noise format.";
  }
  if (fmt == GTG_IO_NOISE_G2O || fmt == GTG_IO_NOISE_COV) {
    if (v[0] == 0 || v[3] == 0 || v[5] == 0) return "load2D::readNoiseModel looks like this is not G2O matrix order";
    const double m[9] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5]};
    std::memcpy(M, m, sizeof m);
  } else if (fmt == GTG_IO_NOISE_TORO || fmt == GTG_IO_NOISE_GRAPH) {
    if (v[0] == 0 || v[2] == 0 || v[3] == 0) return "load2D::readNoiseModel looks like this is not TORO matrix order";
    const double m[9] = {v[0], v[1], v[4], v[1], v[2], v[5], v[4], v[5], v[3]};
    std::memcpy(M, m, sizeof m);
  } else {
    return "load2D: invalid noise format";
  }
  *covariance = fmt == GTG_IO_NOISE_GRAPH || fmt == GTG_IO_NOISE_COV;
  return nullptr;
}

struct Graph {
  bool is3d = true;
  std::vector<int64_t> v1, v2;
  std::vector<double> z, noise;              // 12 / 3 and 36 / 9 doubles per edge
  std::vector<int32_t> kind;
  std::map<int64_t, std::vector<double>> vertices;   // key -> pose (12 doubles, or x y cos sin for a Pose2); iterates by key like Values
};

int fail(const std::string& s) { gtg_io::set_error(s); return GTG_ERR_USAGE; }

int parse3d(const std::vector<Line>& lines, Graph* g) {
  g->is3d = true;
  for (const Line& ln : lines) {
    const bool v_toro = is(ln, "VERTEX3"), v_quat = is(ln, "VERTEX_SE3:QUAT"), e_toro = is(ln, "EDGE3"), e_quat = is(ln, "EDGE_SE3:QUAT");
    if (v_toro || v_quat) {
      int64_t id; double t[3], a[4];
      bool ok = idx(ln, 0, &id) && num(ln, 1, &t[0]) && num(ln, 2, &t[1]) && num(ln, 3, &t[2]);
      for (int k = 0; k < (v_toro ? 3 : 4); k++) ok = ok && num(ln, 4 + (size_t)k, &a[k]);
      if (!ok) return fail("load3D: malformed vertex line");
      std::vector<double> p(12);
      if (v_toro) ypr(a[2], a[1], a[0], p.data()); else quat(a[0], a[1], a[2], a[3], p.data());   // (`is >> roll >> pitch >> yaw`: notice the order)
      p[9] = t[0]; p[10] = t[1]; p[11] = t[2];
      if (!g->vertices.emplace(id, std::move(p)).second) return fail("load3D: a vertex appears twice (ValuesKeyAlreadyExists in the reference)");
    } else if (e_toro || e_quat) {
      int64_t a, b; double t[3], r[4], up[21];
      bool ok = idx(ln, 0, &a) && idx(ln, 1, &b) && num(ln, 2, &t[0]) && num(ln, 3, &t[1]) && num(ln, 4, &t[2]);
      const int nr = e_toro ? 3 : 4;
      for (int k = 0; k < nr; k++) ok = ok && num(ln, 5 + (size_t)k, &r[k]);
      for (int k = 0; k < 21; k++) ok = ok && num(ln, 5 + (size_t)nr + (size_t)k, &up[k]);
      if (!ok) return fail("load3D: malformed edge line");
      double zz[12], m[36], mg[36];
      if (e_toro) ypr(r[2], r[1], r[0], zz); else quat(r[0], r[1], r[2], r[3], zz);
      zz[9] = t[0]; zz[10] = t[1]; zz[11] = t[2];
      int k = 0;
      for (int i = 0; i < 6; i++) for (int j = i; j < 6; j++) { m[6 * i + j] = m[6 * j + i] = up[k++]; }
      if (e_quat) {   // g2o stores t,R order (dataset.cpp:848-853)
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
          mg[6 * i + j] = m[6 * (i + 3) + (j + 3)]; mg[6 * (i + 3) + (j + 3)] = m[6 * i + j];
          mg[6 * (i + 3) + j] = m[6 * i + (j + 3)]; mg[6 * i + (j + 3)] = m[6 * (i + 3) + j];
        }
        std::memcpy(m, mg, sizeof m);
      }
      int32_t kd; double prm[36];
      if (!smart_model(m, 6, false, &kd, prm)) return fail("load3D: the information matrix of an edge is not positive definite");
      g->v1.push_back(a); g->v2.push_back(b); g->z.insert(g->z.end(), zz, zz + 12); g->kind.push_back(kd); g->noise.insert(g->noise.end(), prm, prm + 36);
    }
  }
  return GTG_OK;
}

int parse2d(const std::vector<Line>& lines, int fmt, Graph* g) {
  g->is3d = false;
  // first pass: the VERTEX lines (dataset.cpp:511-522)
  for (const Line& ln : lines) {
    if (!(is(ln, "VERTEX2") || is(ln, "VERTEX_SE2") || is(ln, "VERTEX"))) continue;
    int64_t id; double x, y, yaw;
    if (!(idx(ln, 0, &id) && num(ln, 1, &x) && num(ln, 2, &y) && num(ln, 3, &yaw))) return fail("parseVertexPose encountered malformed line");
    if (!g->vertices.emplace(id, std::vector<double>{x, y, std::cos(yaw), std::sin(yaw)}).second)
      return fail("load2D: a vertex appears twice (ValuesKeyAlreadyExists in the reference)");
  }
  for (const Line& ln : lines) {
    if (!(is(ln, "EDGE2") || is(ln, "EDGE") || is(ln, "EDGE_SE2") || is(ln, "ODOMETRY"))) continue;
    int64_t k1, k2; double zx, zy, zt, v[6];
    bool ok = idx(ln, 0, &k1) && idx(ln, 1, &k2) && num(ln, 2, &zx) && num(ln, 3, &zy) && num(ln, 4, &zt);
    if (!ok) return fail("parseEdge encountered malformed line");
    for (int k = 0; k < 6; k++) ok = ok && num(ln, 5 + (size_t)k, &v[k]);
    if (!ok) return fail("load2D: malformed noise numbers on an edge line");
    double M[9]; bool cov = false;
    if (const char* why = noise_matrix3(v, fmt, M, &cov)) return fail(why);
    int32_t kd; double prm[9];
    if (!smart_model(M, 3, cov, &kd, prm)) return fail("load2D: the noise matrix of an edge is not positive definite");
    // Pose2(x, y, yaw) keeps (cos, sin): theta() = atan2(sin yaw, cos yaw) is the file's angle wrapped into (-pi, pi]
    const double zc = std::cos(zt), zs = std::sin(zt);
    g->v1.push_back(k1); g->v2.push_back(k2);
    g->z.push_back(zx); g->z.push_back(zy); g->z.push_back(std::atan2(zs, zc));
    g->kind.push_back(kd); g->noise.insert(g->noise.end(), prm, prm + 9);
    // vertices a pure odometry file does not list: identity for key1, key1's pose * measurement for key2 (dataset.cpp:541-546; Pose2
    // product: Rot2 through fromCosSin, which renormalises when c^2 + s^2 is off by more than 1e-10, Pose2.h:131-133, Rot2.cpp:27-34)
    if (!g->vertices.count(k1)) g->vertices.emplace(k1, std::vector<double>{0.0, 0.0, 1.0, 0.0});
    if (!g->vertices.count(k2)) {
      const std::vector<double>& p = g->vertices[k1];
      const double x = p[0], y = p[1], c = p[2], s = p[3];
      double nc = c * zc - s * zs, ns = s * zc + c * zs;
      double scale = nc * nc + ns * ns;
      if (std::abs(scale - 1.0) > 1e-10) { scale = 1.0 / std::sqrt(scale); nc *= scale; ns *= scale; }
      g->vertices.emplace(k2, std::vector<double>{x + (c * zx + -s * zy), y + (s * zx + c * zy), nc, ns});
    }
  }
  return GTG_OK;
}

int load(const char* path, int is_3d, int fmt, Graph* g) {
  if (!path) return fail("null argument");
  std::vector<char> buf; std::vector<Line> lines;
  if (!load_lines(path, &buf, &lines)) return fail(std::string("parse: can not find file ") + path);   // dataset.cpp:131-133
  return is_3d ? parse3d(lines, g) : parse2d(lines, fmt, g);
}

// R^T R of a noise-table row
void information(int32_t kind, const double* prm, int d, double* info) {
  for (int i = 0; i < d * d; i++) info[i] = 0.0;
  if (kind == GTG_NOISE_UNIT) { for (int i = 0; i < d; i++) info[d * i + i] = 1.0; return; }
  if (kind == GTG_NOISE_ISOTROPIC) { for (int i = 0; i < d; i++) info[d * i + i] = 1.0 / (prm[0] * prm[0]); return; }
  if (kind == GTG_NOISE_DIAGONAL) { for (int i = 0; i < d; i++) info[d * i + i] = 1.0 / (prm[i] * prm[i]); return; }
  for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) { double s = 0; for (int k = 0; k < d; k++) s += prm[d * k + i] * prm[d * k + j]; info[d * i + j] = s; }
}

}  // namespace

extern "C" {

int gtg_io_g2o_sizes(const char* path, int is_3d, int noise_format, int64_t* n_edges, int64_t* n_vertices) {
  if (!n_edges || !n_vertices) return fail("null argument");
  Graph g;
  const int rc = load(path, is_3d, noise_format, &g);
  if (rc != GTG_OK) return rc;
  *n_edges = (int64_t)g.v1.size(); *n_vertices = (int64_t)g.vertices.size();
  return GTG_OK;
}

int gtg_io_read_g2o(const char* path, int is_3d, int noise_format, int64_t n_edges, int64_t n_vertices, int64_t* edge_v1, int64_t* edge_v2,
                    double* edge_z, int32_t* edge_noise_kind, double* edge_noise, int64_t* vertex_key, double* vertex_pose) {
  if ((n_edges && (!edge_v1 || !edge_v2 || !edge_z || !edge_noise_kind || !edge_noise)) || (n_vertices && (!vertex_key || !vertex_pose))) return fail("null argument");
  Graph g;
  const int rc = load(path, is_3d, noise_format, &g);
  if (rc != GTG_OK) return rc;
  if ((int64_t)g.v1.size() != n_edges || (int64_t)g.vertices.size() != n_vertices) return fail("g2o file does not match the sizes passed in (call gtg_io_g2o_sizes first)");
  std::copy(g.v1.begin(), g.v1.end(), edge_v1); std::copy(g.v2.begin(), g.v2.end(), edge_v2);
  std::copy(g.z.begin(), g.z.end(), edge_z); std::copy(g.kind.begin(), g.kind.end(), edge_noise_kind); std::copy(g.noise.begin(), g.noise.end(), edge_noise);
  int64_t k = 0;
  for (const auto& kv : g.vertices) {
    vertex_key[k] = kv.first;
    if (is_3d) std::copy(kv.second.begin(), kv.second.end(), vertex_pose + 12 * k);
    else { vertex_pose[3 * k] = kv.second[0]; vertex_pose[3 * k + 1] = kv.second[1]; vertex_pose[3 * k + 2] = std::atan2(kv.second[3], kv.second[2]); }
    k++;
  }
  return GTG_OK;
}

int gtg_io_write_g2o(const char* path, int is_3d, int64_t n_edges, const int64_t* edge_v1, const int64_t* edge_v2, const double* edge_z,
                     const int32_t* edge_noise_kind, const double* edge_noise, int64_t n_vertices, const int64_t* vertex_key, const double* vertex_pose,
                     int full_precision) {
  if (!path || (n_edges && (!edge_v1 || !edge_v2 || !edge_z || !edge_noise_kind || !edge_noise)) || (n_vertices && (!vertex_key || !vertex_pose))) return fail("null argument");
  FILE* f = std::fopen(path, "w");
  if (!f) return fail(std::string("writeG2o: can not open ") + path);
  const char* fmt = full_precision ? " %.17g" : " %g";     // `stream << double` prints six significant digits
  if (is_3d) {
    for (int64_t k = 0; k < n_vertices; k++) {
      const double* p = vertex_pose + 12 * k; double q[4];
      to_quaternion(p, q);
      std::fprintf(f, "VERTEX_SE3:QUAT %lld", (long long)vertex_key[k]);
      for (double x : {p[9], p[10], p[11], q[0], q[1], q[2], q[3]}) std::fprintf(f, fmt, x);
      std::fprintf(f, "\n");
    }
    for (int64_t k = 0; k < n_edges; k++) {
      const double* z = edge_z + 12 * k; double q[4], info[36], ig[36];
      information(edge_noise_kind[k], edge_noise + 36 * k, 6, info);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {   // g2o's t,R block order
        ig[6 * i + j] = info[6 * (i + 3) + (j + 3)]; ig[6 * (i + 3) + (j + 3)] = info[6 * i + j];
        ig[6 * i + (j + 3)] = info[6 * (i + 3) + j]; ig[6 * (i + 3) + j] = info[6 * i + (j + 3)];
      }
      to_quaternion(z, q);
      std::fprintf(f, "EDGE_SE3:QUAT %lld %lld", (long long)edge_v1[k], (long long)edge_v2[k]);
      for (double x : {z[9], z[10], z[11], q[0], q[1], q[2], q[3]}) std::fprintf(f, fmt, x);
      for (int i = 0; i < 6; i++) for (int j = i; j < 6; j++) std::fprintf(f, fmt, ig[6 * i + j]);
      std::fprintf(f, "\n");
    }
  } else {
    for (int64_t k = 0; k < n_vertices; k++) {
      std::fprintf(f, "VERTEX_SE2 %lld", (long long)vertex_key[k]);
      for (int i = 0; i < 3; i++) std::fprintf(f, fmt, vertex_pose[3 * k + i]);
      std::fprintf(f, "\n");
    }
    for (int64_t k = 0; k < n_edges; k++) {
      double info[9];
      information(edge_noise_kind[k], edge_noise + 9 * k, 3, info);
      std::fprintf(f, "EDGE_SE2 %lld %lld", (long long)edge_v1[k], (long long)edge_v2[k]);
      for (int i = 0; i < 3; i++) std::fprintf(f, fmt, edge_z[3 * k + i]);
      for (int i = 0; i < 3; i++) for (int j = i; j < 3; j++) std::fprintf(f, fmt, info[3 * i + j]);
      std::fprintf(f, "\n");
    }
  }
  if (std::fclose(f) != 0) { gtg_io::set_error("writeG2o: write failed"); return GTG_ERR_HIP; }
  return GTG_OK;
}

}  // extern "C"
