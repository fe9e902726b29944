// geom.h -- per-factor geometry of the LM hot path, written for one GPU lane per factor.
//
// Device-side re-derivation of exactly the formulas the reference evaluates inside every factor
// (SURVEY.md section 8(a) rows G1, F1-F4, R1); file:line citations are into /root/reference/gtsam.
// Everything is IEEE double, no fast-math, fused multiply-adds left to the compiler's default
// contraction (-ffp-contract=fast-honor-pragmas on hipcc): results agree with the reference to a
// few ulp, not bitwise (tolerances in tests/).
//
// GT_HD lets the same header be compiled by plain g++ for the CPU-side unit parity test of these
// formulas (tests/hostmath/, test infrastructure only -- the product has no CPU path).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define GT_HD inline
#endif

namespace gt {

constexpr double kEps = 2.220446049250313e-16;  // std::numeric_limits<double>::epsilon()
constexpr double kPi = 3.14159265358979323846;

// ---- tiny fixed-size helpers (row-major) ------------------------------------------------------
GT_HD void mat3_mul(const double* A, const double* B, double* C) {  // C = A*B
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
GT_HD void mat3_tmul(const double* A, const double* B, double* C) {  // C = A^T * B
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
GT_HD void mat3_vec(const double* A, const double* x, double* y) {
  for (int i = 0; i < 3; i++) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
GT_HD void mat3_tvec(const double* A, const double* x, double* y) {  // y = A^T x
  for (int i = 0; i < 3; i++) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}
GT_HD void skew3(double x, double y, double z, double* W) {  // skewSymmetric (base/Matrix.h)
  W[0] = 0; W[1] = -z; W[2] = y; W[3] = z; W[4] = 0; W[5] = -x; W[6] = -y; W[7] = x; W[8] = 0;
}

// ---- SO(3) -----------------------------------------------------------------------------------
// SO3::Expmap through so3::ExpmapFunctor (geometry/SO3.cpp:50-88,200-208).
GT_HD void so3_expmap(const double* w, double* R) {
  const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double W[9];
  skew3(w[0], w[1], w[2], W);
  if (theta2 <= kEps) {  // nearZero: I + W
    for (int i = 0; i < 9; i++) R[i] = W[i];
    R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
    return;
  }
  const double theta = sqrt(theta2);
  const double sin_theta = sin(theta);
  const double s2 = sin(theta / 2.0);
  const double one_minus_cos = 2.0 * s2 * s2;
  double K[9], KK[9];
  for (int i = 0; i < 9; i++) K[i] = W[i] / theta;
  mat3_mul(K, K, KK);
  for (int i = 0; i < 9; i++) R[i] = sin_theta * K[i] + one_minus_cos * KK[i];
  R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
}

// SO3::Logmap (geometry/SO3.cpp:247-323): near-pi branch by largest diagonal, acos branch,
// Taylor branch near the identity -- thresholds 1e-3 and -1e-6 as in the reference.
GT_HD void so3_logmap(const double* R, double* omega) {
  const double R11 = R[0], R12 = R[1], R13 = R[2];
  const double R21 = R[3], R22 = R[4], R23 = R[5];
  const double R31 = R[6], R32 = R[7], R33 = R[8];
  const double tr = R11 + R22 + R33;
  if (tr + 1.0 < 1e-3) {
    // axis a = the largest diagonal entry (ties as in the reference: 3 over 2 over 1), (b, c) the next two cyclically;
    // along a: 2 + 2 R_aa, along b: R_ab + R_ba, along c: R_ac + R_ca, antisymmetric part R_cb - R_bc
    double along_a, along_b, along_c, anti;
    int a;
    if (R33 > R22 && R33 > R11) { a = 2; anti = R21 - R12; along_a = 2.0 + 2.0 * R33; along_b = R31 + R13; along_c = R23 + R32; }
    else if (R22 > R11) { a = 1; anti = R13 - R31; along_a = 2.0 + 2.0 * R22; along_b = R23 + R32; along_c = R12 + R21; }
    else { a = 0; anti = R32 - R23; along_a = 2.0 + 2.0 * R11; along_b = R12 + R21; along_c = R31 + R13; }
    const double inv_root = 1 / sqrt(along_a);
    const double len = sqrt(along_a * along_a + along_b * along_b + along_c * along_c + anti * anti);
    const double sign = anti < 0 ? -1.0 : 1.0;
    const double angle = kPi - (2 * sign * anti) / len;
    const double k = 0.5 * inv_root * angle;
    const double wa = sign * k * along_a, wb = sign * k * along_b, wc = sign * k * along_c;
    if (a == 2) { omega[0] = wb; omega[1] = wc; omega[2] = wa; }
    else if (a == 1) { omega[0] = wc; omega[1] = wa; omega[2] = wb; }
    else { omega[0] = wa; omega[1] = wb; omega[2] = wc; }
  } else {
    double magnitude;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-6) {
      const double theta = acos((tr - 1.0) / 2.0);
      magnitude = theta / (2.0 * sin(theta));
    } else {
      magnitude = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0;
    }
    omega[0] = magnitude * (R32 - R23); omega[1] = magnitude * (R13 - R31); omega[2] = magnitude * (R21 - R12);
  }
}

// ---- SE(3): a pose is 12 doubles, R row-major then t ---------------------------------------
// Pose3::operator* (geometry/Pose3.h:114-116)
GT_HD void pose_compose(const double* A, const double* B, double* C) {
  double R[9], t[3];
  mat3_mul(A, B, R);
  mat3_vec(A, B + 9, t);
  for (int i = 0; i < 9; i++) C[i] = R[i];
  for (int i = 0; i < 3; i++) C[9 + i] = A[9 + i] + t[i];
}
// Pose3::inverse (geometry/Pose3.cpp:49-52)
GT_HD void pose_inverse(const double* A, double* B) {
  double nt[3] = {-A[9], -A[10], -A[11]}, t[3];
  mat3_tvec(A, nt, t);
  double R[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = A[3 * j + i];
  for (int i = 0; i < 9; i++) B[i] = R[i];
  for (int i = 0; i < 3; i++) B[9 + i] = t[i];
}
// inverse(A) * B without materialising the inverse: LieGroup::between (base/Lie.h:63-69)
GT_HD void pose_between(const double* A, const double* B, double* C) {
  double Ai[12];
  pose_inverse(A, Ai);
  pose_compose(Ai, B, C);
}
// Pose3::AdjointMap (geometry/Pose3.cpp:57-63): [R 0; [t]x R, R], 6x6 row-major
GT_HD void pose_adjoint(const double* T, double* Ad) {
  double S[9], A[9];
  skew3(T[9], T[10], T[11], S);
  mat3_mul(S, T, A);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Ad[6 * i + j] = T[3 * i + j];
      Ad[6 * i + 3 + j] = 0.0;
      Ad[6 * (3 + i) + j] = A[3 * i + j];
      Ad[6 * (3 + i) + 3 + j] = T[3 * i + j];
    }
}
// Pose3::Expmap (geometry/Pose3.cpp:169-185), xi = [omega; v]
GT_HD void pose_expmap(const double* xi, double* T) {
  const double* w = xi; const double* v = xi + 3;
  so3_expmap(w, T);
  const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (theta2 > kEps) {
    const double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
    const double c[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
    double Rc[3];
    mat3_vec(T, c, Rc);
    for (int i = 0; i < 3; i++) T[9 + i] = (c[i] - Rc[i] + w[i] * wv) / theta2;
  } else {
    T[9] = v[0]; T[10] = v[1]; T[11] = v[2];
  }
}
// Pose3::Logmap (geometry/Pose3.cpp:188-208)
GT_HD void pose_logmap(const double* T, double* xi) {
  double w[3];
  so3_logmap(T, w);
  const double* Tt = T + 9;
  const double t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
  if (t < 1e-10) {
    xi[3] = Tt[0]; xi[4] = Tt[1]; xi[5] = Tt[2];
  } else {
    double W[9], WT[3], WWT[3];
    skew3(w[0] / t, w[1] / t, w[2] / t, W);
    const double Tan = tan(0.5 * t);
    mat3_vec(W, Tt, WT);
    mat3_vec(W, WT, WWT);
    const double c = 1 - t / (2. * Tan);
    for (int i = 0; i < 3; i++) xi[3 + i] = Tt[i] - (0.5 * t) * WT[i] + c * WWT[i];
  }
}
// LieGroup::retract = compose(Expmap(xi)) (base/Lie.h:131-133; GTSAM_POSE3_EXPMAP default)
GT_HD void pose_retract(const double* T, const double* xi, double* out) {
  double E[12];
  pose_expmap(xi, E);
  pose_compose(T, E, out);
}
// LieGroup::localCoordinates = Logmap(between(g)) (base/Lie.h:136-138)
GT_HD void pose_local(const double* A, const double* B, double* xi) {
  double C[12];
  pose_between(A, B, C);
  pose_logmap(C, xi);
}

// ---- pinhole projection -----------------------------------------------------------------------
// PinholeBase::project2 (geometry/CalibratedCamera.cpp:116-135) with Pose3::transformTo
// (Pose3.cpp:371-388), Project (:88-94), Dpose (:27-34), Dpoint (:37-46).
// Returns false on a CheiralityException (q.z <= 0, GTSAM_THROW_CHEIRALITY_EXCEPTION default ON).
// Dpose 2x6, Dpoint 2x3 row-major; pass nullptr to skip derivatives.
GT_HD bool project2(const double* T, const double* pw, double* pn, double* Dpose, double* Dpoint) {
  const double dx[3] = {pw[0] - T[9], pw[1] - T[10], pw[2] - T[11]};
  double q[3];
  mat3_tvec(T, dx, q);
  if (q[2] <= 0) return false;
  const double d = 1.0 / q[2];
  const double u = q[0] * d, v = q[1] * d;
  pn[0] = u; pn[1] = v;
  if (Dpose) {
    const double uv = u * v, uu = u * u, vv = v * v;
    Dpose[0] = uv; Dpose[1] = -1 - uu; Dpose[2] = v; Dpose[3] = -d; Dpose[4] = 0; Dpose[5] = d * u;
    Dpose[6] = 1 + vv; Dpose[7] = -uv; Dpose[8] = -u; Dpose[9] = 0; Dpose[10] = -d; Dpose[11] = d * v;
  }
  if (Dpoint) {  // Rt(i,j) = R(j,i) = T[3*j+i]
    for (int j = 0; j < 3; j++) {
      Dpoint[j] = (T[3 * j + 0] - u * T[3 * j + 2]) * d;
      Dpoint[3 + j] = (T[3 * j + 1] - v * T[3 * j + 2]) * d;
    }
  }
  return true;
}

// PinholeCamera<Cal3Bundler>::project2 (geometry/PinholeCamera.h:230-248) = PinholeBaseK::_project
// (PinholePose.h:89-109) + Cal3Bundler::uncalibrate (Cal3Bundler.cpp:66-92).
// cam = pose(12), f, k1, k2, u0, v0.  Dcam 2x9 = [Dp*Dpose | Dcal], Dpoint 2x3 (row-major).
GT_HD bool sfm_project(const double* cam, const double* pw, double* pi, double* Dcam, double* Dpoint) {
  double pn[2], Dpose[12], Dpt[6];
  // (both derivative buffers or none: a conditionally selected pointer keeps the local arrays in scratch memory on the device)
  const bool want = Dcam || Dpoint;
  if (!(want ? project2(cam, pw, pn, Dpose, Dpt) : project2(cam, pw, pn, nullptr, nullptr))) return false;
  const double f = cam[12], k1 = cam[13], k2 = cam[14], u0 = cam[15], v0 = cam[16];
  const double x = pn[0], y = pn[1];
  const double r = x * x + y * y;
  const double g = 1. + (k1 + k2 * r) * r;
  const double u = g * x, v = g * y;
  pi[0] = u0 + f * u; pi[1] = v0 + f * v;
  if (Dcam || Dpoint) {
    const double a = 2. * (k1 + 2. * k2 * r);
    const double axx = a * x * x, axy = a * x * y, ayy = a * y * y;
    const double Dp[4] = {(g + axx) * f, axy * f, axy * f, (g + ayy) * f};
    if (Dcam) {
      const double rx = r * x, ry = r * y;
      _Pragma("unroll") for (int j = 0; j < 6; j++) {   // (unrolled: Dpose stays in registers instead of scratch memory)
        Dcam[j] = Dp[0] * Dpose[j] + Dp[1] * Dpose[6 + j];
        Dcam[9 + j] = Dp[2] * Dpose[j] + Dp[3] * Dpose[6 + j];
      }
      Dcam[6] = u; Dcam[7] = f * rx; Dcam[8] = f * r * rx;
      Dcam[15] = v; Dcam[16] = f * ry; Dcam[17] = f * r * ry;
    }
    if (Dpoint)
      _Pragma("unroll") for (int j = 0; j < 3; j++) {
        Dpoint[j] = Dp[0] * Dpt[j] + Dp[1] * Dpt[3 + j];
        Dpoint[3 + j] = Dp[2] * Dpt[j] + Dp[3] * Dpt[3 + j];
      }
  }
  return true;
}

// ---- triangulation of a smart factor's landmark (slam/SmartProjectionFactor.h:166-183 -> geometry/triangulation.h:697-752) ----
// Cal3Bundler::calibrate (geometry/Cal3Bundler.cpp:95-128): the intrinsic point of a pixel by the reference's fixed-point
// iteration (at most 10 rounds, stops when uncalibrate(pn) is within tol = 1e-5 pixels -- the same rounds, hence the same point).
// false where the reference throws "fails to converge".
GT_HD bool bundler_calibrate(double f, double k1, double k2, double u0, double v0, const double* pi, double* pn) {
  double px = (pi[0] - u0) / f, py = (pi[1] - v0) / f;
  const double ix = px, iy = py;
  int iteration = 0;
  do {
    const double rr = (px * px) + (py * py);
    const double g = (1 + k1 * rr + k2 * rr * rr);
    pn[0] = ix / g; pn[1] = iy / g;
    const double x = pn[0], y = pn[1], r = x * x + y * y, g2 = 1. + (k1 + k2 * r) * r;
    const double du = (u0 + f * (g2 * x)) - pi[0], dv = (v0 + f * (g2 * y)) - pi[1];
    if (sqrt(du * du + dv * dv) <= 1e-5) break;
    px = pn[0]; py = pn[1];
    iteration++;
  } while (iteration < 10);
  return iteration < 10;
}

// ---- a smart factor's landmark at infinity (slam/SmartProjectionFactor.h:356-371, :419-427) ------------------------------------------
// When the triangulation is not VALID the reference replaces the landmark by the direction of the FIRST measurement seen from the
// FIRST camera, a Unit3 (two degrees of freedom), and projects that into every camera: rotation-only factors.
// Unit3::basis (geometry/Unit3.cpp:73-135): b1 = normalize(n x axis), axis = the coordinate axis of the smallest |n_i| (ties: x, then
// y), b2 = n x b1.
GT_HD void unit3_basis(const double* n, double* b1, double* b2) {
  const double mx = fabs(n[0]), my = fabs(n[1]), mz = fabs(n[2]);
  double a[3] = {0.0, 0.0, 0.0};
  if (mx <= my && mx <= mz) a[0] = 1.0; else if (my <= mx && my <= mz) a[1] = 1.0; else a[2] = 1.0;
  const double c[3] = {n[1] * a[2] - n[2] * a[1], n[2] * a[0] - n[0] * a[2], n[0] * a[1] - n[1] * a[0]};
  const double l = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  b1[0] = c[0] / l; b1[1] = c[1] / l; b1[2] = c[2] / l;
  b2[0] = n[1] * b1[2] - n[2] * b1[1]; b2[1] = n[2] * b1[0] - n[0] * b1[2]; b2[2] = n[0] * b1[1] - n[1] * b1[0];
}
// PinholeBaseK::backprojectPointAtInfinity (geometry/PinholePose.h:164-168): Rot3::rotate(Unit3(calibrate(z), 1)); both Unit3
// constructors normalise (Unit3.cpp:36-38, Rot3.cpp:109-116).  false where Cal3Bundler::calibrate throws.
GT_HD bool sfm_backproject_at_infinity(const double* cam, const double* z, double* dir) {
  double pn[2];
  if (!bundler_calibrate(cam[12], cam[13], cam[14], cam[15], cam[16], z, pn)) return false;
  const double l = sqrt(pn[0] * pn[0] + pn[1] * pn[1] + 1.0);
  const double pc[3] = {pn[0] / l, pn[1] / l, 1.0 / l};
  double w[3];
  mat3_vec(cam, pc, w);
  const double lw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  dir[0] = w[0] / lw; dir[1] = w[1] / lw; dir[2] = w[2] / lw;
  return true;
}
// PinholeCamera<Cal3Bundler>::project2(Unit3) (geometry/PinholeCamera.h:251-254 -> PinholePose.h:89-109 ->
// CalibratedCamera.cpp:138-165): q = Unit3(R^T d) (Rot3.cpp:119-126), pn = (q_x, q_y) / q_z, then Cal3Bundler::uncalibrate.
//   Dpose = Duv B_q B_q^T [q]x on the rotation, ZERO on the translation;   Dpoint (2x2) = Duv B_q B_q^T R^T B_d
// with Duv = [1/q_z 0 -u/q_z; 0 1/q_z -v/q_z] and B the tangent bases.  Duv q = 0 and B_q B_q^T = I - q q^T, so the basis of q drops
// out: Dpose = Duv [q]x, Dpoint = Duv R^T B_d.  Dcam 2x9 = [Dp Dpose | Dcal]; Ddir 2x3 = [Dp Dpoint | 0]: the landmark keeps its
// 3-wide slot in the elimination, the third coordinate is not coupled to anything.
// false on a CheiralityException (q_z <= 0, CalibratedCamera.cpp:146-149) -- which the smart factor does NOT catch.
GT_HD bool sfm_project_at_infinity(const double* cam, const double* dir, double* pi, double* Dcam, double* Ddir) {
  double q[3];
  mat3_tvec(cam, dir, q);
  const double lq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  q[0] /= lq; q[1] /= lq; q[2] /= lq;
  if (q[2] <= 0) return false;
  const double d = 1.0 / q[2];
  const double x = q[0] * d, y = q[1] * d;
  const double f = cam[12], k1 = cam[13], k2 = cam[14], u0 = cam[15], v0 = cam[16];
  const double r = x * x + y * y;
  const double g = 1. + (k1 + k2 * r) * r;
  const double u = g * x, v = g * y;
  pi[0] = u0 + f * u; pi[1] = v0 + f * v;
  if (Dcam || Ddir) {
    const double a = 2. * (k1 + 2. * k2 * r);
    const double axx = a * x * x, axy = a * x * y, ayy = a * y * y;
    const double Dp[4] = {(g + axx) * f, axy * f, axy * f, (g + ayy) * f};
    const double Duv[6] = {d, 0.0, -x * d, 0.0, d, -y * d};
    if (Dcam) {
      double S[9];
      skew3(q[0], q[1], q[2], S);
      for (int j = 0; j < 3; j++) {
        const double r0 = Duv[0] * S[j] + Duv[1] * S[3 + j] + Duv[2] * S[6 + j];
        const double r1 = Duv[3] * S[j] + Duv[4] * S[3 + j] + Duv[5] * S[6 + j];
        Dcam[j] = Dp[0] * r0 + Dp[1] * r1; Dcam[9 + j] = Dp[2] * r0 + Dp[3] * r1;
        Dcam[3 + j] = 0.0; Dcam[12 + j] = 0.0;
      }
      const double rx = r * x, ry = r * y;
      Dcam[6] = u; Dcam[7] = f * rx; Dcam[8] = f * r * rx;
      Dcam[15] = v; Dcam[16] = f * ry; Dcam[17] = f * r * ry;
    }
    if (Ddir) {
      double B[2][3];
      unit3_basis(dir, B[0], B[1]);
      for (int j = 0; j < 2; j++) {
        double t[3];
        mat3_tvec(cam, B[j], t);                                       // R^T b_j
        const double r0 = Duv[0] * t[0] + Duv[2] * t[2], r1 = Duv[4] * t[1] + Duv[5] * t[2];
        Ddir[j] = Dp[0] * r0 + Dp[1] * r1; Ddir[3 + j] = Dp[2] * r0 + Dp[3] * r1;
      }
      Ddir[2] = 0.0; Ddir[5] = 0.0;
    }
  }
  return true;
}

// eigen-decomposition of a symmetric 4x4 matrix by cyclic Jacobi rotations: A -> diagonal, V = eigenvectors (columns)
GT_HD void jacobi_eig4(double (&A)[4][4], double (&V)[4][4]) {
  _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int j = 0; j < 4; j++) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 40; sweep++) {
    double off = 0.0, dia = 0.0;
    _Pragma("unroll") for (int p = 0; p < 4; p++) { dia += A[p][p] * A[p][p]; _Pragma("unroll") for (int q = p + 1; q < 4; q++) off += A[p][q] * A[p][q]; }
    if (off <= 1e-40 * dia || off == 0.0) break;
    _Pragma("unroll") for (int p = 0; p < 3; p++)
      _Pragma("unroll") for (int q = p + 1; q < 4; q++) {
        const double apq = A[p][q];
        if (apq == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        _Pragma("unroll") for (int k = 0; k < 4; k++) {      // A <- A J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq;
        }
        _Pragma("unroll") for (int k = 0; k < 4; k++) {      // A <- J^T A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk;
        }
        _Pragma("unroll") for (int k = 0; k < 4; k++) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq;
        }
      }
  }
}

enum { kTriValid = 0, kTriDegenerate = 1, kTriBehindCamera = 2, kTriOutlier = 3, kTriFarPoint = 4, kTriNoConvergence = 5,
       kTriCheiralityThrown = 6,   // enableEPI: the refinement linearised at a point behind a camera -- the reference throws (see triangulate_refine)
       kTriAtInfinity = 16 };   // flag on a failed status: this use of the factor replaces the landmark by a point at infinity

// ---- TriangulationParameters::enableEPI: gtsam::triangulateNonlinear (geometry/triangulation.h:211-221) ------------------------------
// The DLT point refined by the reference's own LevenbergMarquardtOptimizer on one TriangulationFactor per camera
// (slam/TriangulationFactor.h:121-136: h(x) - z through the camera's full projection, unit noise) with the parameters of
// triangulation.cpp:177-195 -- lambdaInitial 1, lambdaFactor 10 (fixed), at most 100 iterations, absoluteErrorTol 1.0, the rest
// defaults (relativeErrorTol 1e-5, lambdaUpperBound 1e5, minModelFidelity 1e-3, identity damping).  The state machine is that of
// LM.cpp:121-308 / NonlinearOptimizer.cpp:62-117 on one 3-dimensional variable; it stops after a step or two (absoluteErrorTol 1.0),
// so the RESULT depends on every accept / reject decision and is reproduced decision by decision.
//   tri_cost: sum over the cameras of |h - z|^2; with `H`: also A^T A (upper triangle: 00 01 02 11 12 22), A^T b with b = z - h.
//   A point behind a camera evaluates to the constant error (2 fx, 2 fx) (PinholeCamera.h:323-325) -- but LINEARISING there is a
//   CheiralityException nothing catches (TriangulationFactor::linearize projects without evaluateError's try / catch, :148-170):
//   returns false.
GT_HD bool tri_cost(int m, const int32_t* cams, const int64_t* val_off, const double* values, const double* z, const double* pt,
                    double* H, double* g, double* sumsq) {
  double f = 0.0;
  if (H) { for (int i = 0; i < 6; i++) H[i] = 0.0; g[0] = g[1] = g[2] = 0.0; }
  for (int k = 0; k < m; k++) {
    const double* c = values + val_off[cams[k]];
    double cam[17], pi[2], Dp[6];
    for (int j = 0; j < 17; j++) cam[j] = c[j];
    const bool front = H ? sfm_project(cam, pt, pi, nullptr, Dp) : sfm_project(cam, pt, pi, nullptr, nullptr);
    if (!front) {
      if (H) return false;
      f += 2.0 * (2.0 * cam[12]) * (2.0 * cam[12]);
      continue;
    }
    const double b0 = z[2 * k] - pi[0], b1 = z[2 * k + 1] - pi[1];
    f += b0 * b0 + b1 * b1;
    if (H) {
      H[0] += Dp[0] * Dp[0] + Dp[3] * Dp[3]; H[1] += Dp[0] * Dp[1] + Dp[3] * Dp[4]; H[2] += Dp[0] * Dp[2] + Dp[3] * Dp[5];
      H[3] += Dp[1] * Dp[1] + Dp[4] * Dp[4]; H[4] += Dp[1] * Dp[2] + Dp[4] * Dp[5]; H[5] += Dp[2] * Dp[2] + Dp[5] * Dp[5];
      g[0] += Dp[0] * b0 + Dp[3] * b1; g[1] += Dp[1] * b0 + Dp[4] * b1; g[2] += Dp[2] * b0 + Dp[5] * b1;
    }
  }
  *sumsq = f;
  return true;
}
// false: the reference throws (see above); the point is refined in place otherwise.
GT_HD bool triangulate_refine(int m, const int32_t* cams, const int64_t* val_off, const double* values, const double* z, double* pt) {
  double H[6], g[3], bb, f;
  tri_cost(m, cams, val_off, values, z, pt, nullptr, nullptr, &f);
  double err = 0.5 * f, lam = 1.0;
  int iterations = 0;
  if (err <= 0.0) return true;                                      // errorTol = 0 (NonlinearOptimizer.cpp:68-74)
  double new_error = err;
  for (;;) {
    const double current_error = new_error;
    if (!tri_cost(m, cams, val_off, values, z, pt, H, g, &bb)) return false;
    for (;;) {                                                      // tryLambda until a step is accepted or the search is given up
      // the one clique of the damped system: Eigen LLT of H + lambda I (choleskyPartial, base/cholesky.cpp:107-158) and its rank test
      // on the exponents of the last two pivots
      const double a00 = H[0] + lam, a11 = H[3] + lam, a22 = H[5] + lam;
      bool ok = a00 > 0.0;
      const double r00 = sqrt(a00), r01 = H[1] / r00, r02 = H[2] / r00;
      const double x11 = a11 - r01 * r01;
      ok = ok && x11 > 0.0;
      const double r11 = sqrt(x11), r12 = (H[4] - r01 * r02) / r11;
      const double x22 = a22 - r02 * r02 - r12 * r12;
      ok = ok && x22 > 0.0;
      const double r22 = sqrt(x22);
      if (ok) { int e2, e1; (void)frexp(r11, &e2); (void)frexp(r22, &e1); ok = e2 - e1 < 12; }
      bool step_ok = false, stop = false;
      double trial[3] = {0.0, 0.0, 0.0}, trial_err = 0.0;
      if (ok) {
        const double y0 = g[0] / r00, y1 = (g[1] - r01 * y0) / r11, y2 = (g[2] - r02 * y0 - r12 * y1) / r22;   // R^T y = g
        const double d2 = y2 / r22, d1 = (y1 - r12 * d2) / r11, d0 = (y0 - r01 * d1 - r02 * d2) / r00;      // R delta = y
        // linear.error(0) = 0.5 |b|^2; linear.error(delta) = 0.5 |A delta - b|^2 = 0.5 (|b|^2 - 2 g.delta + delta^T H delta), undamped
        const double Hd0 = H[0] * d0 + H[1] * d1 + H[2] * d2, Hd1 = H[1] * d0 + H[3] * d1 + H[4] * d2, Hd2 = H[2] * d0 + H[4] * d1 + H[5] * d2;
        const double old_lin = 0.5 * bb, new_lin = 0.5 * (bb - 2.0 * (g[0] * d0 + g[1] * d1 + g[2] * d2) + (d0 * Hd0 + d1 * Hd1 + d2 * Hd2));
        const double lin_change = old_lin - new_lin;
        if (lin_change >= 0) {
          trial[0] = pt[0] + d0; trial[1] = pt[1] + d1; trial[2] = pt[2] + d2;
          double ft;
          tri_cost(m, cams, val_off, values, z, trial, nullptr, nullptr, &ft);
          trial_err = 0.5 * ft;
          const double cost_change = err - trial_err;
          if (lin_change > kEps * old_lin) step_ok = cost_change / lin_change > 1e-3;
          if (fabs(cost_change) < 1e-5 * err) stop = true;
        }
      }
      if (step_ok) {
        lam = lam / 10.0;                                           // decreaseLambda, fixed factor; lambdaLowerBound 0
        pt[0] = trial[0]; pt[1] = trial[1]; pt[2] = trial[2]; err = trial_err;
        iterations++;
        break;
      } else if (!stop) {
        lam *= 10.0;                                                // increaseLambda
        if (lam >= 1e5) break;
      } else {
        break;
      }
    }
    new_error = err;
    // checkConvergence(relativeErrorTol 1e-5, absoluteErrorTol 1.0, errorTol 0) (NonlinearOptimizer.cpp:182-231)
    const double absolute_decrease = current_error - new_error, relative_decrease = absolute_decrease / current_error;
    const bool converged = new_error <= 0.0 || relative_decrease <= 1e-5 || absolute_decrease <= 1.0;
    if (!(iterations < 100 && !converged && current_error - current_error == 0.0)) break;     // (x - x == 0: finite)
  }
  return true;
}

// gtsam::triangulateSafe for PinholeCamera<Cal3Bundler> cameras (triangulation.h:697-752; useLOST = false):
//   undistort every measurement (calibrate with the camera's Cal3Bundler, uncalibrate with its pinhole part, :261-268),
//   DLT on the projection matrices K [R | t]^-1 (triangulation.cpp:27-57: rows x P_3 - P_1, y P_3 - P_2; the right singular vector
//   of the smallest singular value, rank = singular values above rank_tol -- here from the eigen-decomposition of A^T A),
//   rank < 3 -> DEGENERATE, a camera with the point not in front -> BEHIND_CAMERA (:536-541), then per camera the distance
//   threshold (FAR_POINT) and the largest reprojection error against the outlier threshold (OUTLIER).
// cams[k] -> values + val_off[cams[k]] = pose (R row-major, t), f, k1, k2, u0, v0; z = 2 per measurement.
GT_HD int smart_triangulate(int m, const int32_t* cams, const int64_t* val_off, const double* values, const double* z,
                            double rank_tol, double dist_thr, double outlier_thr, double* point, bool enable_epi = false) {
  point[0] = point[1] = point[2] = 0.0;
  if (m < 2) return kTriDegenerate;
  double M[4][4], V[4][4];
  _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int j = 0; j < 4; j++) M[i][j] = 0.0;
  for (int k = 0; k < m; k++) {
    const double* c = values + val_off[cams[k]];
    const double f = c[12], k1 = c[13], k2 = c[14], u0 = c[15], v0 = c[16];
    double pn[2];
    if (!bundler_calibrate(f, k1, k2, u0, v0, z + 2 * k, pn)) return kTriNoConvergence;
    const double zu[2] = {f * pn[0] + u0, f * pn[1] + v0};        // Cal3_S2(f, f, 0, u0, v0).uncalibrate
    // [R | t]^-1 = [R^T | -R^T t]; P = K [R^T | -R^T t], K = [f 0 u0; 0 f v0; 0 0 1]
    double W[3][4];
    _Pragma("unroll") for (int i = 0; i < 3; i++) {
      W[i][0] = c[0 + i]; W[i][1] = c[3 + i]; W[i][2] = c[6 + i];   // R^T(i, j) = R(j, i)
      W[i][3] = -(c[0 + i] * c[9] + c[3 + i] * c[10] + c[6 + i] * c[11]);
    }
    double r0[4], r1[4];
    _Pragma("unroll") for (int j = 0; j < 4; j++) {
      const double P0 = f * W[0][j] + u0 * W[2][j], P1 = f * W[1][j] + v0 * W[2][j], P2 = W[2][j];
      r0[j] = zu[0] * P2 - P0; r1[j] = zu[1] * P2 - P1;
    }
    _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int j = 0; j < 4; j++) M[i][j] += r0[i] * r0[j] + r1[i] * r1[j];
  }
  jacobi_eig4(M, V);
  int rank = 0, smallest = 0;
  _Pragma("unroll") for (int j = 0; j < 4; j++) {
    const double sv = sqrt(M[j][j] > 0.0 ? M[j][j] : 0.0);
    if (sv > rank_tol) rank++;
    if (M[j][j] < M[smallest][smallest]) smallest = j;
  }
  if (rank < 3) return kTriDegenerate;
  double v[4] = {0, 0, 0, 0};
  _Pragma("unroll") for (int j = 0; j < 4; j++) if (j == smallest) { v[0] = V[0][j]; v[1] = V[1][j]; v[2] = V[2][j]; v[3] = V[3][j]; }
  point[0] = v[0] / v[3]; point[1] = v[1] / v[3]; point[2] = v[2] / v[3];
  if (enable_epi && !triangulate_refine(m, cams, val_off, values, z, point)) return kTriCheiralityThrown;   // (triangulation.h:531-534)
  for (int k = 0; k < m; k++) {       // triangulatePoint3's cheirality check: every camera first
    const double* c = values + val_off[cams[k]];
    const double dx[3] = {point[0] - c[9], point[1] - c[10], point[2] - c[11]};
    double q[3];
    mat3_tvec(c, dx, q);
    if (!(q[2] > 0)) return kTriBehindCamera;
  }
  double max_err = 0.0;
  for (int k = 0; k < m; k++) {
    const double* c = values + val_off[cams[k]];
    if (dist_thr > 0) {
      const double dx = point[0] - c[9], dy = point[1] - c[10], dz = point[2] - c[11];
      if (sqrt(dx * dx + dy * dy + dz * dz) > dist_thr) return kTriFarPoint;
    }
    if (outlier_thr > 0) {
      double cam[17], pi[2];
      for (int j = 0; j < 17; j++) cam[j] = c[j];
      if (sfm_project(cam, point, pi, nullptr, nullptr)) {
        const double eu = pi[0] - z[2 * k], ev = pi[1] - z[2 * k + 1], e = sqrt(eu * eu + ev * ev);
        max_err = e > max_err ? e : max_err;
      }
    }
  }
  if (outlier_thr > 0 && max_err > outlier_thr) return kTriOutlier;
  return kTriValid;
}

// PinholeCamera<Cal3_S2>(pose, K).project (PinholePose.h:89-109) + Cal3_S2::uncalibrate (geometry/Cal3_S2.cpp:44-50), or, when the
// table entry carries distortion coefficients, + Cal3DS2_Base::uncalibrate (geometry/Cal3DS2_Base.cpp:93-132, derivative with
// respect to the intrinsic point D2dintrinsic :71-91).  K = fx, fy, s, u0, v0, k1, k2, p1, p2 (kCalibStride).  Dpose 2x6, Dpoint 2x3.
constexpr int kCalibStride = 9;
GT_HD bool s2_project(const double* T, const double* K, const double* pw, double* pi, double* Dpose,
                      double* Dpoint) {
  double pn[2], Dps[12], Dpt[6];
  const bool want = Dpose || Dpoint;
  if (!(want ? project2(T, pw, pn, Dps, Dpt) : project2(T, pw, pn, nullptr, nullptr))) return false;
  const double fx = K[0], fy = K[1], s = K[2], u0 = K[3], v0 = K[4];
  const double k1 = K[5], k2 = K[6], p1 = K[7], p2 = K[8];
  // d(pixel)/d(intrinsic point): DK for a Cal3_S2, DK * DR with distortion
  double d00 = fx, d01 = s, d10 = 0.0, d11 = fy;
  if (k1 == 0.0 && k2 == 0.0 && p1 == 0.0 && p2 == 0.0) {
    pi[0] = fx * pn[0] + s * pn[1] + u0;
    pi[1] = fy * pn[1] + v0;
  } else {
    const double x = pn[0], y = pn[1], xy = x * y, xx = x * x, yy = y * y;
    const double rr = xx + yy, r4 = rr * rr;
    const double g = 1. + k1 * rr + k2 * r4;
    const double dx = 2. * p1 * xy + p2 * (rr + 2. * xx);
    const double dy = 2. * p2 * xy + p1 * (rr + 2. * yy);
    const double pnx = g * x + dx, pny = g * y + dy;
    pi[0] = fx * pnx + s * pny + u0;
    pi[1] = fy * pny + v0;
    if (want) {
      const double drdx = 2. * x, drdy = 2. * y;
      const double dgdx = k1 * drdx + k2 * 2. * rr * drdx, dgdy = k1 * drdy + k2 * 2. * rr * drdy;
      const double dDxdx = 2. * p1 * y + p2 * (drdx + 4. * x), dDxdy = 2. * p1 * x + p2 * drdy;
      const double dDydx = 2. * p2 * y + p1 * drdx, dDydy = 2. * p2 * x + p1 * (drdy + 4. * y);
      const double r00 = g + x * dgdx + dDxdx, r01 = x * dgdy + dDxdy, r10 = y * dgdx + dDydx, r11 = g + y * dgdy + dDydy;
      d00 = fx * r00 + s * r10; d01 = fx * r01 + s * r11;
      d10 = 0.0 * r00 + fy * r10; d11 = 0.0 * r01 + fy * r11;
    }
  }
  if (Dpose)
    _Pragma("unroll") for (int j = 0; j < 6; j++) {
      Dpose[j] = d00 * Dps[j] + d01 * Dps[6 + j];
      Dpose[6 + j] = d10 * Dps[j] + d11 * Dps[6 + j];
    }
  if (Dpoint)
    _Pragma("unroll") for (int j = 0; j < 3; j++) {
      Dpoint[j] = d00 * Dpt[j] + d01 * Dpt[3 + j];
      Dpoint[3 + j] = d10 * Dpt[j] + d11 * Dpt[3 + j];
    }
  return true;
}

// ---- noise models (linear/NoiseModel.cpp) ------------------------------------------------------
enum { kNoiseUnit = 0, kNoiseIsotropic = 1, kNoiseDiagonal = 2, kNoiseGaussian = 3 };
// Device noise table entry data: ISOTROPIC {invsigma}; DIAGONAL invsigmas[dim]; GAUSSIAN R row-major.
// whiten a column vector / the columns of a row-major m x n matrix in place:
//   Unit no-op; Isotropic v*invsigma (:641-663); Diagonal v.*invsigmas (:311-325); Gaussian R*v (:163-181)
template <int M>
GT_HD void whiten_cols(int kind, const double* nd, double* A, int ncols) {
  if (kind == kNoiseUnit) return;
  if (kind == kNoiseIsotropic) {
    const double s = nd[0];
    for (int i = 0; i < M * ncols; i++) A[i] *= s;
  } else if (kind == kNoiseDiagonal) {
    for (int r = 0; r < M; r++) for (int c = 0; c < ncols; c++) A[r * ncols + c] *= nd[r];
  } else {
    for (int c = 0; c < ncols; c++) {
      double col[M];
      for (int r = 0; r < M; r++) col[r] = A[r * ncols + c];
      for (int r = 0; r < M; r++) {
        double acc = 0.0;
        for (int k = 0; k < M; k++) acc += nd[r * M + k] * col[k];
        A[r * ncols + c] = acc;
      }
    }
  }
}

// ---- Pose2 (geometry/Pose2.{h,cpp}, Rot2.{h,cpp}) --------------------------------------------------------------
// Stored as (x, y, theta); the reference keeps (c, s) instead of theta and multiplies rotations as complex numbers
// (Rot2.h:116-118: fromCosSin(c1 c2 - s1 s2, s1 c2 + c1 s2), re-normalised only when |c^2+s^2-1| > 1e-10,
// Rot2.cpp:56-64); theta() = atan2(s, c) (Rot2.h:186-188).  h = a^-1 b with
//   a^-1 = (R_a^T, R_a^T (-t_a))                       Pose2.cpp:201-203
//   p q  = (R_p R_q, t_p + R_p t_q)                    Pose2.h:131-133
// returned as (x, y, c, s).
GT_HD void rot2_normalize(double& c, double& s) {
  double scale = c * c + s * s;
  if (fabs(scale - 1.0) > 1e-10) { scale = 1.0 / sqrt(scale); c *= scale; s *= scale; }
}
GT_HD void pose2_between_cs(double xa, double ya, double ca, double sa, double xb, double yb, double cb, double sb,
                            double* h) {
  // inverse of a: rotation (ca, -sa), translation unrotate(-t_a)
  const double ix = ca * (-xa) + sa * (-ya), iy = -sa * (-xa) + ca * (-ya);
  double c = ca * cb - (-sa) * sb, s = (-sa) * cb + ca * sb;
  rot2_normalize(c, s);
  h[0] = ix + (ca * xb + sa * yb);        // t_inv + R_inv t_b   (Rot2::rotate with (c, s) = (ca, -sa), Rot2.cpp:100-106)
  h[1] = iy + (-sa * xb + ca * yb);
  h[2] = c; h[3] = s;
}
// Local(a, b) = ChartAtOrigin::Local(between(a, b)) = (x, y, theta) of a^-1 b  (Lie.h:136-138, Pose2.cpp:111-121,
// GTSAM_SLOW_BUT_CORRECT_EXPMAP off)
GT_HD void pose2_local(const double* a, const double* b, double* d) {
  double h[4];
  pose2_between_cs(a[0], a[1], cos(a[2]), sin(a[2]), b[0], b[1], cos(b[2]), sin(b[2]), h);
  d[0] = h[0]; d[1] = h[1]; d[2] = atan2(h[3], h[2]);
}
// Retract(a, v) = a * Pose2(v0, v1, v2)  (Lie.h:131-133, Pose2.cpp:99-109)
GT_HD void pose2_retract(const double* a, const double* v, double* y) {
  const double ca = cos(a[2]), sa = sin(a[2]), cv = cos(v[2]), sv = sin(v[2]);
  double c = ca * cv - sa * sv, s = sa * cv + ca * sv;
  rot2_normalize(c, s);
  y[0] = a[0] + (ca * v[0] + -sa * v[1]);
  y[1] = a[1] + (sa * v[0] + ca * v[1]);
  y[2] = atan2(s, c);
}

// traits<T>::Local(x, z) for the supported value types (PriorFactor.h:98-102):
// Pose3 Logmap(between); PinholeCamera [pose local; calib diff] (PinholeCamera.h:208-213,
// Cal3Bundler.h:150-152); Point3 z - x; Pose2 above.   vtype: 0 POSE3, 1 SFM_CAMERA, 2 POINT3, 3 POSE2.
GT_HD void value_local(int vtype, const double* x, const double* z, double* d) {
  if (vtype == 2) { d[0] = z[0] - x[0]; d[1] = z[1] - x[1]; d[2] = z[2] - x[2]; return; }
  if (vtype == 3) { pose2_local(x, z, d); return; }
  pose_local(x, z, d);
  if (vtype == 1) { d[6] = z[12] - x[12]; d[7] = z[13] - x[13]; d[8] = z[14] - x[14]; }
}
// traits<T>::Retract(x, d): Pose3 (Lie.h:131-133), PinholeCamera::retract (PinholeCamera.h:199-205)
// with Cal3Bundler::retract (Cal3Bundler.h:145-147), Point3 x + d, Pose2 above.
GT_HD void value_retract(int vtype, const double* x, const double* d, double* y) {
  if (vtype == 2) { y[0] = x[0] + d[0]; y[1] = x[1] + d[1]; y[2] = x[2] + d[2]; return; }
  if (vtype == 3) { pose2_retract(x, d, y); return; }
  pose_retract(x, d, y);
  if (vtype == 1) { y[12] = x[12] + d[6]; y[13] = x[13] + d[7]; y[14] = x[14] + d[8]; y[15] = x[15]; y[16] = x[16]; }
}

}  // namespace gt
