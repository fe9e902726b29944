// assemble.hip -- J^T J / J^T b block accumulation, landmark elimination (Schur complement) and
// back-substitution.
//
// What the reference does per lambda try inside eliminateMultifrontal with the Schur ordering
// (SURVEY.md section 8(a) rows S1-S4, H1, H2):
//   HessianFactor(factors, scatter) + updateHessian      linear/HessianFactor.cpp:239-252, JacobianFactor.cpp:563-598
//   choleskyPartial on every landmark clique (3 frontals) linear/HessianFactor.cpp:459-487, base/cholesky.cpp:107-158
//   separator HessianFactors summed into the camera cliques (= the Schur complement)
//   GaussianBayesTree::optimize back-substitution        linear/linearAlgorithms-inst.h:49-155
// is split here into a lambda-INVARIANT part done once per linearization (diagonal blocks, gradients,
// off-diagonal pose-pose blocks, Hessian diagonal) and a per-lambda part (damping, 3x3 landmark
// Cholesky, E = W L^-T, S = H_cc + lambda D - sum E E^T).  All sums run in a host-fixed order: no
// floating-point atomics, results are bit-reproducible run to run.
#include "factors.h"
#include "fused.h"
#include <string>

#include "kernels.h"
#include "recio.h"

namespace gt {

constexpr int kBlock = 256;
enum { INC_SFM = 0, INC_PROJ = 1, INC_BTW_A = 2, INC_BTW_B = 3, INC_PRIOR = 4 };

struct JTabs {
  const double *sfm_J, *proj_J, *bt_J, *pr_J;
  int64_t n_sfm;
};

// (A, rows, row stride, b) of one contribution to a reduced variable
__device__ __forceinline__ void contribution(const JTabs& t, int kind, int idx, int d, const double*& A,
                                             int& rows, const double*& b) {
  if (kind == INC_SFM) { const double* J = t.sfm_J + (int64_t)kSfmRec * idx; A = J; rows = 2; b = J + 24; }
  else if (kind == INC_PROJ) { const double* J = t.proj_J + (int64_t)kProjRec * idx; A = J; rows = 2; b = J + 18; }
  else if (kind == INC_BTW_A) { const double* J = t.bt_J + (int64_t)kBetweenRec * idx; A = J; rows = d; b = J + 72; }   // d = 6 Pose3, 3 Pose2
  else if (kind == INC_BTW_B) { const double* J = t.bt_J + (int64_t)kBetweenRec * idx; A = J + 36; rows = d; b = J + 72; }
  else { const double* J = t.pr_J + (int64_t)kPriorRec * idx; A = J; rows = d; b = J + 81; }
}

// observation o -> (Jc 2 x dc, Jp 2x3, b 2)
__device__ __forceinline__ void obs_rec(const JTabs& t, int64_t o, const double*& Jc, const double*& Jp,
                                        const double*& b, int& dc) {
  if (o < t.n_sfm) { const double* J = t.sfm_J + (int64_t)kSfmRec * o; Jc = J; Jp = J + 18; b = J + 24; dc = 9; }
  else { const double* J = t.proj_J + (int64_t)kProjRec * (o - t.n_sfm); Jc = J; Jp = J + 12; b = J + 18; dc = 6; }
}

// ---- lambda-invariant assembly --------------------------------------------------------------------
// One workgroup per reduced variable: H_dd = sum A^T A and g_d = sum A^T b over its contributions (whitened Jacobian
// rows of every factor touching it).  The sum over rows is a contraction, run on the FP64 matrix core with b as one
// more column: per step a lane loads one entry, the 64 lanes cover up to 4 rows of [A | b] of ONE contribution
// (contiguous in the factor's record), and C[i][j] = sum_q A[q][i] [A | b][q][j].  The 16 waves split the
// contribution list and are combined through LDS in wave order (deterministic).
typedef double v4f64a __attribute__((ext_vector_type(4)));
constexpr int kRedWaves = 16;   // waves per reduced variable: a 16-camera problem still has 256 wavefronts of work
__global__ __launch_bounds__(64 * kRedWaves) void k_red_diag(int32_t n_red_vars, const int64_t* __restrict__ inc_ptr,
    const int32_t* __restrict__ inc_kind, const int32_t* __restrict__ inc_idx, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, JTabs t, double* __restrict__ Hd, double* __restrict__ g,
    double* __restrict__ hdiag, int after_fused) {
  // after_fused: the GeneralSFM contributions are in Hd / g / hdiag already (k_cam_fused): skip them here and ADD the rest
  __shared__ double part[kRedWaves][256];
  const int r = blockIdx.x;
  if (r >= n_red_vars) return;
  const int d = red_dim[r];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const int nent = d * d + d;
  const int64_t beg = inc_ptr[r], end = inc_ptr[r + 1];
  v4f64a acc = {0.0, 0.0, 0.0, 0.0};
  for (int64_t k = beg + wave; k < end; k += kRedWaves) {
    const double* A; const double* b; int rows;
    if (after_fused && inc_kind[k] == INC_SFM) continue;   // (wave-uniform)
    contribution(t, inc_kind[k], inc_idx[k], d, A, rows, b);
    const int boff = (int)(b - A);
    for (int q0 = 0; q0 < rows; q0 += 4) {
      const int q = q0 + lk;
      const bool valid = q < rows && lr <= d;
      const int off = valid ? (lr < d ? q * d + lr : boff + q) : 0;   // branch-free: masked lanes re-read entry 0
      const double v = A[off];
      const double bv = valid ? v : 0.0;
      const double av = lr < d ? bv : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int rr = 0; rr < 4; rr++) part[wave][rr * 64 + lane] = acc[rr];   // C[row = lk + 4 rr][col = lr]
  __syncthreads();
  const int e = threadIdx.x;
  if (e < nent) {
    const int i = e < d * d ? e / d : e - d * d, j = e < d * d ? e % d : d;
    const int idx = (i >> 2) * 64 + 16 * (i & 3) + j;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kRedWaves; w++) s += part[w][idx];   // wave order: deterministic
    if (e < d * d) {
      if (after_fused) s += Hd[(int64_t)81 * r + e];
      Hd[(int64_t)81 * r + e] = s;
      if (i == j) hdiag[red_off[r] + i] = s;
    } else {
      if (after_fused) s += g[(int64_t)9 * r + i];
      g[(int64_t)9 * r + i] = s;
    }
  }
}

// ---- the same sums without stored records (fused.h; north_star: "one-factor-per-wavefront kernels ... the small dense Jacobian blocks
// reduced into J^T J / J^T r Hessian blocks staged in LDS").  One factor per LANE here, 64 per wavefront: a record is 2 rows of 13, and
// the reduction over the factors of a camera is what runs wavefront-wide, on the matrix core, out of the LDS image.
// k_cam_fused: the camera-sorted pass.  Workgroup (r, sp) takes the sp-th part of reduced variable r's contribution list; a wavefront
// takes 64 GeneralSFM entries at a time: every lane recomputes its factor's record [Jc | Jp | b] into the wavefront's LDS image (gather of
// the camera -- the same 17 doubles for the whole workgroup: L1 --, its point, its measurement), then the 128 rows [Jc | b] of the
// image are contracted 4 at a time: C += [Jc]^T [Jc | b], 32 MFMAs per 64 factors with both operands read from LDS (conflict free:
// the 64 lanes of a step read one contiguous 54-word window).  Wave partials combined in wave order, parts in part order: deterministic.
constexpr int kCamWaves = 4;
constexpr int kCamPartStride = 96;   // doubles per (variable, part) in the partial buffer (>= 9 * 9 + 9)
__global__ __launch_bounds__(64 * kCamWaves) void k_cam_fused(int32_t n_red_vars, int splits, const int64_t* __restrict__ inc_ptr,
    const CamPack* __restrict__ pack, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, SfmTabs t, double* __restrict__ Hd, double* __restrict__ g, double* __restrict__ hdiag,
    double* __restrict__ part) {
  typedef RecIO<kSfmRec> IO;
  __shared__ double img[kCamWaves][IO::LDS_DOUBLES];
  __shared__ double comb[kCamWaves][256];
  const int r = blockIdx.x / splits, sp = blockIdx.x - r * splits;
  if (r >= n_red_vars) return;
  const int d = red_dim[r];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const int64_t beg = inc_ptr[r], len = inc_ptr[r + 1] - beg;
  const int64_t s0 = beg + len * sp / splits, s1 = beg + len * (sp + 1) / splits;
  double* my = img[wave];
  double* rec = my + lane * IO::PITCH;
  const int col_off = lr < 9 ? lr : 24;                      // column lr of [Jc | b]: Jc[row][lr] at 9 row + lr, b[row] at 24 + row; lanes lr > 9 are masked
  v4f64a acc = {0.0, 0.0, 0.0, 0.0};
  for (int64_t base = s0 + 64 * wave; base < s1; base += 64 * kCamWaves) {
    const int64_t k = base + lane;
    CamPack e; e.cam_at = -1;
    if (k < s1) e = pack[k];
    if (e.cam_at >= 0) sfm_record_at(t, e.cam_at, e.pt_at, e.nz, e.z0, e.z1, rec);
    else
#pragma unroll
      for (int q = 0; q < kSfmRec; q++) rec[q] = 0.0;
    IO::wave_sync();
#pragma unroll 8
    for (int m = 0; m < 32; m++) {
      const int row = 4 * m + lk;
      const double* R = my + (row >> 1) * IO::PITCH;
      const double v = R[(lr < 9 ? 9 : 1) * (row & 1) + col_off];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lr < 9 ? v : 0.0, lr <= 9 ? v : 0.0, acc, 0, 0, 0);
    }
    IO::wave_sync();
  }
#pragma unroll
  for (int rr = 0; rr < 4; rr++) comb[wave][rr * 64 + lane] = acc[rr];   // C[row = lk + 4 rr][col = lr]
  __syncthreads();
  const int e = threadIdx.x, nent = d * d + d;
  if (e < nent) {
    const int i = e < d * d ? e / d : e - d * d, j = e < d * d ? e % d : 9;
    const int idx = (i >> 2) * 64 + 16 * (i & 3) + j;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kCamWaves; w++) s += comb[w][idx];
    if (splits > 1) part[(int64_t)blockIdx.x * kCamPartStride + e] = s;
    else if (e < d * d) { Hd[(int64_t)81 * r + e] = s; if (i == j) hdiag[red_off[r] + i] = s; }
    else g[(int64_t)9 * r + i] = s;
  }
}
// ... the parts of a variable added up in part order (graphs with few cameras and thousands of observations each)
__global__ __launch_bounds__(128) void k_cam_combine(int32_t n_red_vars, int splits, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, const double* __restrict__ part, double* __restrict__ Hd, double* __restrict__ g,
    double* __restrict__ hdiag) {
  const int r = blockIdx.x, e = threadIdx.x;
  if (r >= n_red_vars) return;
  const int d = red_dim[r];
  if (e >= d * d + d) return;
  double s = 0.0;
  for (int sp = 0; sp < splits; sp++) s += part[((int64_t)r * splits + sp) * kCamPartStride + e];
  if (e < d * d) { Hd[(int64_t)81 * r + e] = s; if (e / d == e % d) hdiag[red_off[r] + e / d] = s; }
  else g[(int64_t)9 * r + (e - d * d)] = s;
}

// k_lm_fused: the landmark-sorted pass.  A wavefront owns 64 consecutive landmarks -- and with them a contiguous stretch of the
// landmark -> observation list.  It walks that stretch 64 observations at a time: every LANE recomputes one observation's record into
// its row of the LDS image (all 64 lanes busy, whatever the track lengths), then every landmark's lane adds up ITS observations of the
// chunk in list order.  Same sums in the same order as k_lm_diag: V and gp are bit-identical to the stored-record form.  (First
// version, measured: one landmark per lane recomputing its records one after the other -- 250 us on the L1723 shape against 82 us for
// k_lm_diag: a lane with a 30-camera track kept 63 others waiting through 30 dependent gathers.)
__global__ __launch_bounds__(kBlock) void k_lm_fused(int32_t n_lm, const int64_t* __restrict__ obs_ptr,
    const int32_t* __restrict__ obs, const int64_t* __restrict__ pri_ptr, const int32_t* __restrict__ pri,
    JTabs t, SfmTabs st, double* __restrict__ V, double* __restrict__ gp) {
  typedef RecIO<kSfmRec> IO;
  __shared__ double img[kBlock / 64][IO::LDS_DOUBLES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* my = img[wave];
  const int64_t ngroups = ((int64_t)n_lm + 63) / 64, gstride = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t grp = blockIdx.x * (int64_t)(kBlock / 64) + wave; grp < ngroups; grp += gstride) {
    const int64_t l = grp * 64 + lane;
    const bool have = l < n_lm;
    const int64_t k0 = have ? obs_ptr[l] : 0, k1 = have ? obs_ptr[l + 1] : 0;
    // the stretch of the whole group: from the first landmark's first observation to the last landmark's last
    const int64_t lfirst = grp * 64, llast = (lfirst + 64 < n_lm ? lfirst + 64 : (int64_t)n_lm);
    const int64_t g0 = obs_ptr[lfirst], g1 = obs_ptr[llast];
    double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int64_t base = g0; base < g1; base += 64) {
      const int64_t k = base + lane;
      double* rec = my + lane * IO::PITCH;
      if (k < g1) {
        const int64_t o = obs[k];
        if (o < t.n_sfm) sfm_record(st, o, rec);
        else {   // a GenericProjectionFactor observation: its stored record's landmark part, into the same places of the row
          const double* J = t.proj_J + (int64_t)kProjRec * (o - t.n_sfm);
#pragma unroll
          for (int e = 0; e < 6; e++) rec[18 + e] = J[12 + e];
          rec[24] = J[18]; rec[25] = J[19];
        }
      }
      IO::wave_sync();
      // this lane's landmark: its observations inside [base, base + 64)
      const int64_t a = k0 > base ? k0 : base, b = k1 < base + 64 ? k1 : base + 64;
      for (int64_t kk = a; kk < b; kk++) {
        const double* R = my + (kk - base) * IO::PITCH;
        const double* Jp = R + 18;
        const double* bb = R + 24;
        for (int i = 0; i < 3; i++) {
          for (int j = 0; j < 3; j++) v[3 * i + j] += Jp[i] * Jp[j] + Jp[3 + i] * Jp[3 + j];
          g[i] += Jp[i] * bb[0] + Jp[3 + i] * bb[1];
        }
      }
      IO::wave_sync();
    }
    if (!have) continue;
    for (int64_t k = pri_ptr[l]; k < pri_ptr[l + 1]; k++) {
      const double* J = t.pr_J + (int64_t)kPriorRec * pri[k];
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) for (int q = 0; q < 3; q++) v[3 * i + j] += J[3 * q + i] * J[3 * q + j];
        for (int q = 0; q < 3; q++) g[i] += J[3 * q + i] * J[81 + q];
      }
    }
    for (int i = 0; i < 9; i++) V[9 * l + i] = v[i];
    for (int i = 0; i < 3; i++) gp[3 * l + i] = g[i];
  }
}

// One lane per landmark: V = sum Jp^T Jp (+ priors), gp = sum Jp^T b.
__global__ __launch_bounds__(kBlock) void k_lm_diag(int32_t n_lm, const int64_t* __restrict__ obs_ptr,
    const int32_t* __restrict__ obs, const int64_t* __restrict__ pri_ptr, const int32_t* __restrict__ pri,
    JTabs t, double* __restrict__ V, double* __restrict__ gp) {
  for (int64_t l = blockIdx.x * (int64_t)kBlock + threadIdx.x; l < n_lm; l += (int64_t)gridDim.x * kBlock) {
    double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int64_t k = obs_ptr[l]; k < obs_ptr[l + 1]; k++) {
      const double *Jc, *Jp, *b; int dc;
      obs_rec(t, obs[k], Jc, Jp, b, dc);
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) v[3 * i + j] += Jp[i] * Jp[j] + Jp[3 + i] * Jp[3 + j];
        g[i] += Jp[i] * b[0] + Jp[3 + i] * b[1];
      }
    }
    for (int64_t k = pri_ptr[l]; k < pri_ptr[l + 1]; k++) {
      const double* J = t.pr_J + (int64_t)kPriorRec * pri[k];
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) for (int q = 0; q < 3; q++) v[3 * i + j] += J[3 * q + i] * J[3 * q + j];
        for (int q = 0; q < 3; q++) g[i] += J[3 * q + i] * J[81 + q];
      }
    }
    for (int i = 0; i < 9; i++) V[9 * l + i] = v[i];
    for (int i = 0; i < 3; i++) gp[3 * l + i] = g[i];
  }
}

// One wavefront per off-diagonal pose-pose block (d x d, d = 6 Pose3 / 3 Pose2): sum over the BetweenFactors joining
// the pair.
__global__ __launch_bounds__(64) void k_hoff(int64_t n_blocks, const int64_t* __restrict__ ptr,
    const int32_t* __restrict__ fac, const int32_t* __restrict__ row, const int32_t* __restrict__ red_dim,
    const double* __restrict__ bt_J, double* __restrict__ Hoff) {
  const int64_t blk = blockIdx.x;
  if (blk >= n_blocks) return;
  const int d = red_dim[row[blk]];
  const int e = threadIdx.x;
  if (e >= d * d) return;
  const int i = e / d, j = e % d;
  double acc = 0.0;
  for (int64_t k = ptr[blk]; k < ptr[blk + 1]; k++) {
    const int code = fac[k];
    const int f = code & 0x3fffffff;
    const bool swap = (code >> 30) & 1;  // row variable is the factor's key2
    const double* J = bt_J + (int64_t)kBetweenRec * f;
    const double* X = swap ? J + 36 : J;
    const double* Y = swap ? J : J + 36;
    for (int q = 0; q < d; q++) acc += X[d * q + i] * Y[d * q + j];
  }
  Hoff[(int64_t)81 * blk + e] = acc;
}

// ---- per lambda -------------------------------------------------------------------------------------
// what the damping prior adds to a diagonal entry: JacobianFactor(key, A, 0, Isotropic::Sigma(dim,
// 1/sqrt(lambda))) with A = I or diag(sqrt(clamp(H_jj))) (internal/LevenbergMarquardtState.h:125-156,
// LM.cpp:293-299): whitened A' = A * invsigma, contribution A'^2.
__device__ __forceinline__ double damp_term(double hjj, double invsigma, int diag, double dmin, double dmax) {
  double a = 1.0;
  if (diag) a = sqrt(fmin(fmax(hjj, dmin), dmax));
  const double w = a * invsigma;
  return w * w;
}

// One lane per landmark: damped 3x3 Cholesky (Eigen LLT semantics + the exponent test of
// base/cholesky.cpp:144-157), L^-1 and y = L^-1 gp.
__global__ __launch_bounds__(kBlock) void k_point_factor(int32_t n_lm, const int32_t* __restrict__ owned,
    const double* __restrict__ V, const double* __restrict__ gp, double invsigma, int diag, double dmin,
    double dmax, double* __restrict__ Linv, double* __restrict__ y, double* __restrict__ fail,
    const int32_t* __restrict__ lm_smart, const int32_t* __restrict__ smart_status) {
  for (int64_t l = blockIdx.x * (int64_t)kBlock + threadIdx.x; l < n_lm; l += (int64_t)gridDim.x * kBlock) {
    if (!owned[l]) continue;
    const double* v = V + 9 * l;
    // The landmark of a smart factor is eliminated inside the factor: P = (E^T E)^-1 without damping (linearize() passes
    // lambda = 0, SmartProjectionFactor.h:322-331, CameraSet.h:325-343); one that did not triangulate contributes nothing.
    // A failed track under IGNORE_DEGENERACY / HANDLE_INFINITY is a point at infinity with TWO degrees of freedom (kTriAtInfinity,
    // factors.hip k_lin_smart_at_infinity): its E blocks have a zero third column, so row / column 3 of V is exactly zero; a 1 on
    // that diagonal entry leaves P = (E^T E)^-1 of the 2 x 2 block and an uncoupled third coordinate (y_3 = 0).
    const int sm = lm_smart ? lm_smart[l] : -1;
    const int st = sm >= 0 ? smart_status[sm] : 0;
    if (st != 0 && !(st & kTriAtInfinity)) {
      for (int k = 0; k < 9; k++) Linv[9 * l + k] = 0.0;
      y[3 * l] = 0.0; y[3 * l + 1] = 0.0; y[3 * l + 2] = 0.0;
      continue;
    }
    const double a00 = v[0] + (sm >= 0 ? 0.0 : damp_term(v[0], invsigma, diag, dmin, dmax));
    const double a11 = v[4] + (sm >= 0 ? 0.0 : damp_term(v[4], invsigma, diag, dmin, dmax));
    const double a22 = (st & kTriAtInfinity) ? 1.0 : v[8] + (sm >= 0 ? 0.0 : damp_term(v[8], invsigma, diag, dmin, dmax));
    const double a10 = v[3], a20 = v[6], a21 = v[7];
    bool bad = !(a00 > 0.0);
    const double l00 = sqrt(a00);
    const double l10 = a10 / l00, l20 = a20 / l00;
    const double x11 = a11 - l10 * l10;
    bad = bad || !(x11 > 0.0);
    const double l11 = sqrt(x11);
    const double l21 = (a21 - l20 * l10) / l11;
    const double x22 = a22 - l20 * l20 - l21 * l21;
    bad = bad || !(x22 > 0.0);
    const double l22 = sqrt(x22);
    int ex2, ex1;
    (void)frexp(l11, &ex2);
    (void)frexp(l22, &ex1);
    bad = bad || (sm < 0 && !(ex2 - ex1 < 12));   // (choleskyPartial's rank test belongs to the eliminated cliques, not to a smart factor's inverse)
    if (bad) *fail = 1.0;
    // inverse of the lower-triangular factor
    const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
    const double i10 = -l10 * i00 * i11;
    const double i21 = -l21 * i11 * i22;
    const double i20 = -(l20 * i00 + l21 * i10) * i22;
    double* Li = Linv + 9 * l;
    Li[0] = i00; Li[1] = 0; Li[2] = 0; Li[3] = i10; Li[4] = i11; Li[5] = 0; Li[6] = i20; Li[7] = i21; Li[8] = i22;
    const double* g = gp + 3 * l;
    // forward substitution for y (same operation order as a triangular solve, not Linv*g)
    const double y0 = g[0] / l00;
    const double y1 = (g[1] - l10 * y0) / l11;
    const double y2 = (g[2] - l20 * y0 - l21 * y1) / l22;
    y[3 * l] = y0; y[3 * l + 1] = y1; y[3 * l + 2] = y2;
  }
}

// E_o = Jc^T (Jp L^-T) for the observations [0, n) of one factor type, one observation per lane.  The Jacobian records
// are read and the 256-byte E slots written through the wavefront's LDS image (recio.h): 1 KiB contiguous per memory
// instruction in both directions.
// FUSED (GeneralSFM records of a graph without smart factors): the record is recomputed in the image instead of loaded (fused.h).
// Also out: w_o = E_o y_l (9 doubles per observation, zero padded), the observation's summand of the reduced right-hand side, stored at the
// observation's place in its camera's contribution list -- so that k_build_diag reads 72 contiguous bytes per entry instead of gathering the
// 216-byte E block and y (283 MB per try on the L1723 shape until round 4).
template <int REC, int DC, bool FUSED>
__global__ __launch_bounds__(kBlock) void k_obs_E(int64_t n, const double* __restrict__ J, SfmTabs t, const int32_t* __restrict__ obs_lm,
    const double* __restrict__ Linv, const double* __restrict__ ylm, double* __restrict__ E, double* __restrict__ W,
    const int32_t* __restrict__ wpos) {
  typedef RecIO<REC> IN;
  typedef RecIO<kEStride> OUT;
  __shared__ double img[kBlock / 64][OUT::LDS_DOUBLES > IN::LDS_DOUBLES ? OUT::LDS_DOUBLES : IN::LDS_DOUBLES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* my = img[wave];
  const int64_t nchunks = (n + 63) / 64, stride = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t ch = blockIdx.x * (int64_t)(kBlock / 64) + wave; ch < nchunks; ch += stride) {
    const int64_t o = ch * 64 + lane, left = n - ch * 64;
    const int nrec = left < 64 ? (int)left : 64;
    if constexpr (FUSED) { if (o < n) sfm_record(t, o, my + lane * IN::PITCH); }
    else IN::load(my, J + (int64_t)REC * ch * 64, nrec, lane);
    double Eo[kEStride], w[9];
#pragma unroll
    for (int k = 0; k < kEStride; k++) Eo[k] = 0.0;
#pragma unroll
    for (int i = 0; i < 9; i++) w[i] = 0.0;
    if (o < n) {
      const double* rec = my + lane * IN::PITCH;     // [Jc 2 x DC | Jp 2 x 3 | b 2]
      const int64_t lm = obs_lm[o];
      const double* Li = Linv + 9 * lm;
      const double* y = ylm + 3 * lm;
      double T[6];
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int m = 0; m < 3; m++) {
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k <= m; k++) acc += rec[2 * DC + 3 * r + k] * Li[3 * m + k];
          T[3 * r + m] = acc;
        }
#pragma unroll
      for (int i = 0; i < DC; i++)
#pragma unroll
        for (int m = 0; m < 3; m++) Eo[3 * i + m] = rec[i] * T[m] + rec[DC + i] * T[3 + m];
      const double y0 = y[0], y1 = y[1], y2 = y[2];
#pragma unroll
      for (int i = 0; i < DC; i++) w[i] = Eo[3 * i] * y0 + Eo[3 * i + 1] * y1 + Eo[3 * i + 2] * y2;
    }
    IN::wave_sync();   // every lane has read its record: the image becomes the E block
#pragma unroll
    for (int k = 0; k < kEStride; k++) my[lane * OUT::PITCH + k] = Eo[k];
    OUT::store(my, E + (int64_t)kEStride * ch * 64, nrec, lane);
    // w_o goes where k_build_diag reads it: to the observation's place in its camera's contribution list (wpos), so that the sum over a
    // camera's observations is a sequential read (in observation order it was a 72-byte gather per observation: 92 us for 49 MB)
    if (o < n) {
      double* wo = W + 9 * (int64_t)wpos[o];
#pragma unroll
      for (int i = 0; i < 9; i++) wo[i] = w[i];
    }
    IN::wave_sync();   // the image may be overwritten afterwards
  }
}

// v_o = Jp^T (Jc x_cam) for the back-substitution, one observation per lane, records through LDS as above
template <int REC, int DC, bool FUSED>
__global__ __launch_bounds__(kBlock) void k_obs_v(int64_t n, const double* __restrict__ J, SfmTabs t, const int32_t* __restrict__ obs_red,
    const int64_t* __restrict__ red_off, const double* __restrict__ x, double* __restrict__ v) {
  typedef RecIO<REC> IN;
  __shared__ double img[kBlock / 64][IN::LDS_DOUBLES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* my = img[wave];
  const int64_t nchunks = (n + 63) / 64, stride = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t ch = blockIdx.x * (int64_t)(kBlock / 64) + wave; ch < nchunks; ch += stride) {
    const int64_t o = ch * 64 + lane, left = n - ch * 64;
    if constexpr (FUSED) { if (o < n) sfm_record(t, o, my + lane * IN::PITCH); }     // (a lane reads its own record only)
    else IN::load(my, J + (int64_t)REC * ch * 64, left < 64 ? (int)left : 64, lane);
    if (o < n) {
      const double* rec = my + lane * IN::PITCH;
      const double* xr = x + red_off[obs_red[o]];
      double w0 = 0.0, w1 = 0.0;
#pragma unroll
      for (int i = 0; i < DC; i++) { w0 += rec[i] * xr[i]; w1 += rec[DC + i] * xr[i]; }
      const double* Jp = rec + 2 * DC;
      v[3 * o] = Jp[0] * w0 + Jp[3] * w1; v[3 * o + 1] = Jp[1] * w0 + Jp[4] * w1; v[3 * o + 2] = Jp[2] * w0 + Jp[5] * w1;
    }
  }
}

// One wavefront per reduced variable: damped diagonal block into S, rhs entries g - sum E y (the summands w_o = E_o y_l come from
// k_obs_E) into the extra row NP of S.
constexpr int kDiagWaves = 4;   // (one wavefront per variable until round 5: 1 723 wavefronts on 256 CUs walked ~6 dependent gathers each, 90 - 146 us)
__global__ __launch_bounds__(64 * kDiagWaves) void k_build_diag(int32_t n_red_vars, const int64_t* __restrict__ inc_ptr,
    const int32_t* __restrict__ inc_kind, const int32_t* __restrict__ inc_idx, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, int64_t n_sfm,
    const double* __restrict__ Hd, const double* __restrict__ g, const double* __restrict__ hdiag,
    const double* __restrict__ W, double invsigma,
    int diag, double dmin, double dmax, int add_damping, SMat S) {
  __shared__ double part[kDiagWaves][9];
  const int r = blockIdx.x;
  if (r >= n_red_vars) return;
  const int d = red_dim[r];
  const int64_t off = red_off[r];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < d * d) {
    const int i = tid / d, j = tid % d;
    double v = Hd[(int64_t)81 * r + tid];
    if (i == j && add_damping) v += damp_term(hdiag[off + i], invsigma, diag, dmin, dmax);
    if (double* q = S.at_stored(off + i, off + j)) *q = v;
  }
  // rhs: g - sum of the observations' summands, each thread every (64 kDiagWaves)-th entry of the variable's list, then lanes, then
  // wavefronts: a fixed order
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t k = inc_ptr[r] + tid; k < inc_ptr[r + 1]; k += 64 * kDiagWaves) {
    if (inc_kind[k] > INC_PROJ) continue;
    const double* wo = W + 9 * k;      // E_o y_l of the observation behind entry k, put there by k_obs_E
    for (int i = 0; i < d; i++) acc[i] += wo[i];
  }
  for (int i = 0; i < 9; i++)
    for (int s2 = 32; s2 > 0; s2 >>= 1) acc[i] += __shfl_down(acc[i], s2, 64);
  if (lane == 0)
    for (int i = 0; i < 9; i++) part[wave][i] = acc[i];
  __syncthreads();
  if (tid < d) {
    double sum = 0.0;
    for (int w = 0; w < kDiagWaves; w++) sum += part[w][tid];
    *S.at((int64_t)kTile * S.nt, off + tid) = g[(int64_t)9 * r + tid] - sum;   // rhs row
  }
}

__global__ __launch_bounds__(64) void k_scatter_hoff(int64_t n_blocks, const int32_t* __restrict__ row,
    const int32_t* __restrict__ col, const int32_t* __restrict__ red_dim, const int64_t* __restrict__ red_off,
    const double* __restrict__ Hoff, SMat S) {
  const int64_t blk = blockIdx.x;
  if (blk >= n_blocks) return;
  const int d = red_dim[row[blk]];
  if ((int)threadIdx.x >= d * d) return;
  const int i = threadIdx.x / d, j = threadIdx.x % d;
  if (double* q = S.at_stored(red_off[row[blk]] + i, red_off[col[blk]] + j)) *q = Hoff[(int64_t)81 * blk + threadIdx.x];
}

// One wavefront per block pair (a,b) of the reduced system: S_ab -= sum_t E_a(t) E_b(t)^T over the landmarks seen
// by both, on the FP64 matrix core: one v_mfma_f64_16x16x4 per term, the contraction index being the 3 landmark
// coordinates (k = 3 is a zero lane group; rows/columns >= d masked to zero); lane (row lr, k lk) supplies entry 3 lr + lk
// of the 9x3 block.  The term's slot indices are wave-uniform scalar loads.  Fixed summation order: deterministic.
// The kernel is bound by the memory system's rate for scattered 256-byte granules (12 M slot fetches on the L1723 shape,
// 89 M on Venice; rounds 2 - 5 measured six re-orderings of the same fetches without a gain).  Round 6: one load instruction
// fetches the four E slots of TWO terms with 16-byte lanes (lanes 0-15: a(t), 16-31: b(t), 32-47: a(t+1), 48-63: b(t+1);
// 16 lanes x 16 B = one 256-byte slot), the wavefront's 1 KB goes through its private LDS patch (LDS operations of one
// wavefront complete in program order: no barrier) and every lane picks its MFMA operands out of it.  Per two terms:
// 1 global_load_dwordx4 + 1 ds_write_b128 + 4 ds_read_b64 + 2 MFMA, against 4 global_load_dwordx2 + 2 MFMA with rounds
// 1 - 5's 8-byte lanes (one load = one slot): the same terms in the same order, S bit for bit, a quarter of the
// vector-memory instructions -- 0.829 -> 0.786 ms on L1723, 4.41 -> 4.04 ms on Venice (profiles/r06h_schur_wide_ab.txt).
typedef double v4f64s __attribute__((ext_vector_type(4)));
// Last session of round 6 (profiles/r06_schur_sweep.txt): with its occupancy cut by untouched dynamic LDS the kernel slows nearly in
// proportion -- 0.77 / 1.08 / 1.72 / 3.10 ms at 32 / 20 / 12 / 4 wavefronts per CU on L1723 -- so what it needs is round trips in flight.
// Now: kWidePairs = 8 load instructions in flight per wavefront (4 before), the loads of a block's LAST group clamped to its last term
// instead of a tail loop that fetched two terms per round trip (a block of 30 terms: 2 round trips, 6 before), every lane fetching the
// index of its own slot (no scalar index walk), and ONE 1 KB patch per wavefront used load after load -- LDS operations of a wavefront
// complete in program order.  Same terms in the same order: S bit for bit.  0.773 -> 0.70 ms (L1723), 4.02 -> 3.75 ms (Venice).
// Measured beside it and not kept: 6 / 12 / 16 loads in flight (0.72 / 0.77 / 0.90 ms: registers cost wavefronts), the two terms of a
// load on accumulators of their own (0.78: the dependent MFMA chain is not what a wavefront waits for), XCD-contiguous block ranges
// (0.73 / 4.27 ms, as in round 5).
#ifndef GT_SCHUR_WIDE
#define GT_SCHUR_WIDE 8
#endif

constexpr int kWidePairs = GT_SCHUR_WIDE;       // load instructions in flight per wavefront (2 terms each)
__global__ __launch_bounds__(256) void k_schur_pairs(int64_t n_pairs, const int32_t* __restrict__ prow,
    const int32_t* __restrict__ pcol, const int64_t* __restrict__ pptr, const int32_t* __restrict__ oa,
    const int32_t* __restrict__ ob, const int32_t* __restrict__ red_dim, const int64_t* __restrict__ red_off,
    const double* __restrict__ E, SMat S) {
  __shared__ double patch[4][4 * kEStride];      // [wavefront][4 slots]
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t p = blockIdx.x * (int64_t)4 + wv;
  if (p >= n_pairs) return;
  const int ra = prow[p], rb = pcol[p];
  const int da = red_dim[ra], db = red_dim[rb];
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int grp = lane >> 4, piece = lane & 15;      // as a loader: slot `grp` of the four, its doubles 2 piece, 2 piece + 1
  const int64_t k0 = pptr[p], k1 = pptr[p + 1];
  const bool ina = lr < da && lk < 3, inb = lr < db && lk < 3;
  const int ea = ina ? 3 * lr + lk : 0, eb = inb ? 3 * lr + lk : 0;
  v4f64s acc = {0.0, 0.0, 0.0, 0.0};
  double* my = patch[wv];
  // every lane fetches the index of ITS slot (lane groups 0 / 2: oa of the pair's first / second term, 1 / 3: ob): per-lane loads, all
  // kWidePairs of them in flight, then the kWidePairs slot loads -- no scalar branch between them (the wave-uniform form made hipcc walk the
  // four indices of a load under exec-mask branches, one scalar round trip after the other)
  const int32_t* __restrict__ my_idx = (grp & 1) ? ob : oa;
  const int64_t last = k1 - 1;
  for (int64_t t = k0; t < k1; t += 2 * kWidePairs) {
    int slot[kWidePairs];
#pragma unroll
    for (int u = 0; u < kWidePairs; u++) {
      const int64_t tt = t + 2 * u + (grp >> 1);
      slot[u] = my_idx[tt < last ? tt : last];    // (terms behind the block's last one: its slots once more -- an L1 hit, never multiplied)
    }
    double2 v[kWidePairs];
#pragma unroll
    for (int u = 0; u < kWidePairs; u++) v[u] = *reinterpret_cast<const double2*>(E + kEStride * (int64_t)slot[u] + 2 * piece);
    // all kWidePairs loads are issued before the first one is waited for: hipcc sinks each load to its use otherwise (one round trip after
    // the other); the empty asm statements make every loaded value live HERE, in the straight-line code behind the last load
#pragma unroll
    for (int u = 0; u < kWidePairs; u++) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y));
    // (the patch traffic is unconditional -- it keeps every load of the group in the straight-line code in front of it: with the whole
    // step under the "term exists" test hipcc sank each load to its use, one in flight at a time -- only the MFMAs are skipped)
#pragma unroll
    for (int u = 0; u < kWidePairs; u++) {
      *reinterpret_cast<double2*>(&my[2 * lane]) = v[u];
      const double a0 = my[ea], b0 = my[kEStride + eb], a1 = my[2 * kEStride + ea], b1 = my[3 * kEStride + eb];
      if (t + 2 * u < k1) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ina ? a0 : 0.0, inb ? b0 : 0.0, acc, 0, 0, 0);
      if (t + 2 * u + 1 < k1) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ina ? a1 : 0.0, inb ? b1 : 0.0, acc, 0, 0, 0);
    }
  }

  const int64_t oa_ = red_off[ra], ob_ = red_off[rb];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = lk + 4 * r;
    if (row < da && lr < db) if (double* q = S.at_stored(oa_ + row, ob_ + lr)) *q -= acc[r];
  }
}

// The same contraction for graphs with few cameras and thousands of common landmarks per pair (Dubrovnik-16 shape: 136
// pairs, ~2 000 terms each): one workgroup per pair, its 16 waves take every 16th group of 4 terms, partial tiles are
// combined through LDS in wave order (deterministic).
constexpr int kPairWaves = 16;
__global__ __launch_bounds__(64 * kPairWaves) void k_schur_pairs_heavy(int64_t n_pairs, const int32_t* __restrict__ prow,
    const int32_t* __restrict__ pcol, const int64_t* __restrict__ pptr, const int32_t* __restrict__ oa,
    const int32_t* __restrict__ ob, const int32_t* __restrict__ red_dim, const int64_t* __restrict__ red_off,
    const double* __restrict__ E, SMat S) {
  __shared__ double part[kPairWaves][256];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t p = blockIdx.x;
  const int ra = prow[p], rb = pcol[p];
  const int da = red_dim[ra], db = red_dim[rb];
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int64_t k0 = pptr[p], k1 = pptr[p + 1];
  const bool ina = lr < da && lk < 3, inb = lr < db && lk < 3;
  const int ea = ina ? 3 * lr + lk : 0, eb = inb ? 3 * lr + lk : 0;
  v4f64s acc = {0.0, 0.0, 0.0, 0.0};
  for (int64_t t = k0 + 4 * wv; t < k1; t += 4 * kPairWaves) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (t + u < k1) {
        const double av = E[kEStride * (int64_t)oa[t + u] + ea], bv = E[kEStride * (int64_t)ob[t + u] + eb];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ina ? av : 0.0, inb ? bv : 0.0, acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) part[wv][r * 64 + lane] = acc[r];
  __syncthreads();
  if (wv == 0) {
    const int64_t oa_ = red_off[ra], ob_ = red_off[rb];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = lk + 4 * r;
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < kPairWaves; w++) sum += part[w][r * 64 + lane];
      if (row < da && lr < db) if (double* q = S.at_stored(oa_ + row, ob_ + lr)) *q -= sum;
    }
  }
}

// identity on the padded rows / columns of S (tail padding to the tile size, alignment gaps between the parts)
__global__ void k_pad_diag(SMat S, const int64_t* __restrict__ pad, int64_t npad) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < npad) { const int64_t i = pad[t]; *S.at(i, i) = 1.0; }
}

// ---- back-substitution ---------------------------------------------------------------------------------
// delta_p = L^-T (y - sum_obs E^T x_cam)   (x_F = R^-1 (d - S x_S)).  With E = Jc^T Jp L^-T the sum is
// L^-1 sum_obs Jp^T (Jc x_cam): k_obs_v forms the summands from the Jacobian records (coalesced, one observation per
// lane), this kernel (one landmark per lane) adds them in the landmark's observation order and finishes.
__global__ __launch_bounds__(kBlock) void k_backsub_lm(int32_t n_lm, const int32_t* __restrict__ owned,
    const int64_t* __restrict__ obs_ptr, const int32_t* __restrict__ obs, const double* __restrict__ v,
    const double* __restrict__ Linv, const double* __restrict__ ylm, double* __restrict__ dlm) {
  for (int64_t l = blockIdx.x * (int64_t)kBlock + threadIdx.x; l < n_lm; l += (int64_t)gridDim.x * kBlock) {
    if (!owned[l]) { dlm[3 * l] = 0; dlm[3 * l + 1] = 0; dlm[3 * l + 2] = 0; continue; }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t k = obs_ptr[l]; k < obs_ptr[l + 1]; k++) {
      const double* vo = v + 3 * (int64_t)obs[k];
      s0 += vo[0]; s1 += vo[1]; s2 += vo[2];
    }
    const double* Li = Linv + 9 * l;     // row-major lower-triangular L^-1
    const double a0 = ylm[3 * l] - Li[0] * s0;
    const double a1 = ylm[3 * l + 1] - (Li[3] * s0 + Li[4] * s1);
    const double a2 = ylm[3 * l + 2] - (Li[6] * s0 + Li[7] * s1 + Li[8] * s2);
    dlm[3 * l] = Li[0] * a0 + Li[3] * a1 + Li[6] * a2;
    dlm[3 * l + 1] = Li[4] * a1 + Li[7] * a2;
    dlm[3 * l + 2] = Li[8] * a2;
  }
}

__global__ __launch_bounds__(kBlock) void k_scatter_delta(int32_t n_vars, const int32_t* __restrict__ lm_index,
    const int32_t* __restrict__ red_index, const int32_t* __restrict__ red_dim, const int64_t* __restrict__ red_off,
    const int64_t* __restrict__ dim_off, const double* __restrict__ x, const double* __restrict__ dlm,
    double* __restrict__ delta) {
  for (int64_t v = blockIdx.x * (int64_t)kBlock + threadIdx.x; v < n_vars; v += (int64_t)gridDim.x * kBlock) {
    double* d = delta + dim_off[v];
    const int l = lm_index[v];
    if (l >= 0) { d[0] = dlm[3 * l]; d[1] = dlm[3 * l + 1]; d[2] = dlm[3 * l + 2]; }
    else { const int r = red_index[v]; const double* xr = x + red_off[r]; for (int i = 0; i < red_dim[r]; i++) d[i] = xr[i]; }
  }
}

// ---- launchers ------------------------------------------------------------------------------------------
static inline int grid1(int64_t n) { int64_t b = (n + kBlock - 1) / kBlock; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }
static JTabs jtabs(gtg_context& c) { return JTabs{c.f.sfm_J.p, c.f.proj_J.p, c.f.between_J.p, c.f.prior_J.p, c.f.n_sfm}; }

// observation -> its place in its camera's contribution list (once per graph, behind the incidence lists)
__global__ __launch_bounds__(kBlock) void k_obs_wpos(int64_t n_inc, const int32_t* __restrict__ inc_kind, const int32_t* __restrict__ inc_idx,
                                                     int64_t n_sfm, int32_t* __restrict__ wpos) {
  for (int64_t k = blockIdx.x * (int64_t)kBlock + threadIdx.x; k < n_inc; k += (int64_t)gridDim.x * kBlock) {
    const int kind = inc_kind[k];
    if (kind == INC_SFM) wpos[inc_idx[k]] = (int32_t)k;
    else if (kind == INC_PROJ) wpos[n_sfm + inc_idx[k]] = (int32_t)k;
  }
}
// once per graph: the camera-sorted contribution lists' GeneralSFM entries packed in list order (fused.h::CamPack)
__global__ __launch_bounds__(kBlock) void k_cam_pack(int64_t n_inc, const int32_t* __restrict__ inc_kind, const int32_t* __restrict__ inc_idx,
    const int32_t* __restrict__ cam_at, const int32_t* __restrict__ pt_at, const int32_t* __restrict__ nz, const double* __restrict__ z,
    CamPack* __restrict__ pack) {
  for (int64_t k = blockIdx.x * (int64_t)kBlock + threadIdx.x; k < n_inc; k += (int64_t)gridDim.x * kBlock) {   // (grid1 caps the grid)
    CamPack e; e.cam_at = -1; e.pt_at = 0; e.nz = 0; e.pad = 0; e.z0 = 0.0; e.z1 = 0.0;
    if (inc_kind[k] == INC_SFM) {
      const int64_t i = inc_idx[k];
      e.cam_at = cam_at[i]; e.pt_at = pt_at[i]; e.nz = nz[i]; e.z0 = z[2 * i]; e.z1 = z[2 * i + 1];
    }
    pack[k] = e;
  }
}
void launch_obs_wpos(gtg_context& c, int64_t n_inc) {
  if (n_inc > 0 && c.n_obs > 0)
    hipLaunchKernelGGL(k_obs_wpos, dim3(grid1(n_inc)), dim3(kBlock), 0, c.stream, n_inc, c.red_inc_kind.p, c.red_inc_idx.p, c.f.n_sfm, c.obs_wpos.p);
  if (c.fused_sfm) {
    static_assert(sizeof(CamPack) == 4 * sizeof(double), "CamPack is stored in a buffer of doubles");
    c.cam_pack.alloc(4 * (size_t)std::max<int64_t>(n_inc, 1));
    if (n_inc > 0)
      hipLaunchKernelGGL(k_cam_pack, dim3(grid1(n_inc)), dim3(kBlock), 0, c.stream, n_inc, c.red_inc_kind.p, c.red_inc_idx.p, c.f.sfm_cam_at.p,
                         c.f.sfm_point_at.p, c.f.sfm_noise.p, c.f.sfm_z.p, reinterpret_cast<CamPack*>(c.cam_pack.p));
  }
  check_hip(hipGetLastError(), "obs_wpos");
}

void launch_assemble(gtg_context& c) {
  JTabs t = jtabs(c);
  if (c.n_red_vars && c.fused_sfm) {
    // several workgroups per camera where there are few cameras (a 16-camera graph has thousands of observations per camera)
    const int splits = c.n_red_vars >= 512 ? 1 : std::min(16, (1024 + c.n_red_vars - 1) / c.n_red_vars);
    if (splits > 1 && (int64_t)c.cam_part.n != (int64_t)c.n_red_vars * splits * kCamPartStride) c.cam_part.alloc((size_t)c.n_red_vars * splits * kCamPartStride);
    hipLaunchKernelGGL(k_cam_fused, dim3((unsigned)(c.n_red_vars * splits)), dim3(64 * kCamWaves), 0, c.stream, c.n_red_vars, splits, c.red_inc_ptr.p,
                       reinterpret_cast<const CamPack*>(c.cam_pack.p), c.red_dim.p, c.red_off.p, sfm_tabs(c), c.Hd.p, c.gred0.p, c.hdiag_red.p, c.cam_part.p);
    if (splits > 1)
      hipLaunchKernelGGL(k_cam_combine, dim3((unsigned)c.n_red_vars), dim3(128), 0, c.stream, c.n_red_vars, splits, c.red_dim.p, c.red_off.p,
                         c.cam_part.p, c.Hd.p, c.gred0.p, c.hdiag_red.p);
    if (c.f.n_proj + c.f.n_between + c.f.n_prior > 0)     // the other factor types' contributions on top
      hipLaunchKernelGGL(k_red_diag, dim3(c.n_red_vars), dim3(64 * kRedWaves), 0, c.stream, c.n_red_vars, c.red_inc_ptr.p,
                         c.red_inc_kind.p, c.red_inc_idx.p, c.red_dim.p, c.red_off.p, t, c.Hd.p, c.gred0.p, c.hdiag_red.p, 1);
  } else if (c.n_red_vars)
    hipLaunchKernelGGL(k_red_diag, dim3(c.n_red_vars), dim3(64 * kRedWaves), 0, c.stream, c.n_red_vars, c.red_inc_ptr.p,
                       c.red_inc_kind.p, c.red_inc_idx.p, c.red_dim.p, c.red_off.p, t, c.Hd.p, c.gred0.p, c.hdiag_red.p, 0);
  if (c.n_lm && c.fused_sfm)
    hipLaunchKernelGGL(k_lm_fused, dim3(grid1(c.n_lm)), dim3(kBlock), 0, c.stream, c.n_lm, c.lm_obs_ptr.p, c.lm_obs.p,
                       c.lm_pri_ptr.p, c.lm_pri.p, t, sfm_tabs(c), c.V.p, c.gp.p);
  else if (c.n_lm)
    hipLaunchKernelGGL(k_lm_diag, dim3(grid1(c.n_lm)), dim3(kBlock), 0, c.stream, c.n_lm, c.lm_obs_ptr.p, c.lm_obs.p,
                       c.lm_pri_ptr.p, c.lm_pri.p, t, c.V.p, c.gp.p);
  if (c.n_hoff)
    hipLaunchKernelGGL(k_hoff, dim3((unsigned)c.n_hoff), dim3(64), 0, c.stream, c.n_hoff, c.hoff_ptr.p, c.hoff_fac.p,
                       c.hoff_row.p, c.red_dim.p, c.f.between_J.p, c.Hoff.p);
  check_hip(hipGetLastError(), "assemble");
}

static inline double inv_sigma(double lambda) {
  // CachedModel(dim, 1.0 / std::sqrt(lambda)) -> Isotropic: invsigma_ = 1.0 / sigma (LMState.h:117-121)
  const double sigma = 1.0 / std::sqrt(lambda);
  return 1.0 / sigma;
}

// ---- smart factors: what differs from an explicit landmark -------------------------------------------------------------
// (1) The Hessian factor of a smart factor is the Schur complement of its point, so hessianDiagonal() -- what diagonal damping
// scales with (LevenbergMarquardtOptimizer.cpp:281-299) -- sees diag(F^T F - F^T E P E^T F) for its cameras, not diag(F^T F):
// one lane per camera / pose subtracts the squared row norms of E_o = Jc^T Jp L^-T over its smart observations (in list order).
__global__ __launch_bounds__(kBlock) void k_smart_hdiag(int32_t n_red_vars, const int64_t* __restrict__ inc_ptr, const int32_t* __restrict__ inc_kind,
    const int32_t* __restrict__ inc_idx, const int32_t* __restrict__ red_dim, const int64_t* __restrict__ red_off,
    const int32_t* __restrict__ sfm_smart, const int32_t* __restrict__ status, const double* __restrict__ E, double* __restrict__ hdiag) {
  // one workgroup per camera / pose: the incidence list dealt round-robin to the lanes, partial sums combined in lane order
  // (a lane per camera took 50 ms per linearisation on a 150-camera scene with 1 500 measurements each)
  __shared__ double part[kBlock][9];
  const int r = blockIdx.x;
  if (r >= n_red_vars) return;
  const int d = red_dim[r];
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t k = inc_ptr[r] + threadIdx.x; k < inc_ptr[r + 1]; k += kBlock) {
    if (inc_kind[k] != 0) continue;                      // GeneralSFM observations only
    const int64_t o = inc_idx[k];
    const int sm = sfm_smart[o];
    if (sm < 0 || (status[sm] != 0 && !(status[sm] & kTriAtInfinity))) continue;
    const double* Eo = E + kEStride * o;
    for (int i = 0; i < d; i++) acc[i] += Eo[3 * i] * Eo[3 * i] + Eo[3 * i + 1] * Eo[3 * i + 1] + Eo[3 * i + 2] * Eo[3 * i + 2];
  }
  for (int i = 0; i < 9; i++) part[threadIdx.x][i] = acc[i];
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) for (int i = 0; i < 9; i++) part[threadIdx.x][i] += part[threadIdx.x + s][i];
    __syncthreads();
  }
  if ((int)threadIdx.x < d) hdiag[red_off[r] + threadIdx.x] -= part[0][threadIdx.x];
}
// (2) The constant of that Hessian factor is b^T b (CameraSet.h:224), not b^T b - |L^-1 E^T b|^2: linear.error(delta) of the
// reference lies 0.5 |y_l|^2 per contributing smart landmark above the error of the explicit system at the optimal point update.
// A JacobianFactorQ / JacobianFactorSVD (JACOBIAN_Q, JACOBIAN_SVD) is Q [F | b] with the projector Q = I - E P E^T instead: its
// constant IS b^T Q b, so there linear.error(0) lies 0.5 |y_l|^2 BELOW 0.5 |b|^2 and linear.error(delta) needs no correction.
__global__ __launch_bounds__(kBlock) void k_smart_lin1(int32_t n_lm, const int32_t* __restrict__ owned, const int32_t* __restrict__ lm_smart,
                                                       const int32_t* __restrict__ status, const double* __restrict__ params, const double* __restrict__ ylm, double* __restrict__ scalars) {
  __shared__ double sm_[2][kBlock];
  double acc_h = 0.0, acc_j = 0.0;
  for (int l = threadIdx.x; l < n_lm; l += kBlock) {
    const int s = lm_smart[l];
    if (s < 0 || !owned[l] || (status[s] != 0 && !(status[s] & kTriAtInfinity))) continue;
    const double yy = ylm[3 * l] * ylm[3 * l] + ylm[3 * l + 1] * ylm[3 * l + 1] + ylm[3 * l + 2] * ylm[3 * l + 2];
    if (params[8 * s + 5] == 0.0) acc_h += yy; else acc_j += yy;
  }
  sm_[0][threadIdx.x] = acc_h; sm_[1][threadIdx.x] = acc_j;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sm_[0][threadIdx.x] += sm_[0][threadIdx.x + s]; sm_[1][threadIdx.x] += sm_[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { scalars[SC_LIN1] += 0.5 * sm_[0][0]; scalars[SC_LIN0] -= 0.5 * sm_[1][0]; }
}

void launch_smart_hdiag(gtg_context& c) {
  if (!c.n_smart) return;
  hipLaunchKernelGGL(k_smart_hdiag, dim3((unsigned)std::max(c.n_red_vars, 1)), dim3(kBlock), 0, c.stream, c.n_red_vars, c.red_inc_ptr.p, c.red_inc_kind.p,
                     c.red_inc_idx.p, c.red_dim.p, c.red_off.p, c.sfm_smart.p, c.smart_lin_status.p, c.E.p, c.hdiag_red.p);
  check_hip(hipGetLastError(), "smart_hdiag");
}
void launch_smart_lin1(gtg_context& c) {
  if (!c.n_smart) return;
  hipLaunchKernelGGL(k_smart_lin1, dim3(1), dim3(kBlock), 0, c.stream, c.n_lm, c.lm_owned.p, c.lm_smart.p, c.smart_lin_status.p, c.smart_params.p, c.ylm.p, c.scalars.p);
  check_hip(hipGetLastError(), "smart_lin1");
}

void launch_point_eliminate(gtg_context& c, double lambda, int diag, double dmin, double dmax) {
  if (!c.n_lm) return;
  const double is = inv_sigma(lambda);
  hipLaunchKernelGGL(k_point_factor, dim3(grid1(c.n_lm)), dim3(kBlock), 0, c.stream, c.n_lm, c.lm_owned.p, c.V.p,
                     c.gp.p, is, diag, dmin, dmax, c.Linv.p, c.ylm.p, c.scalars.p + SC_FAIL, c.n_smart ? c.lm_smart.p : nullptr, c.smart_lin_status.p);
  const SfmTabs st = sfm_tabs(c);
  if (c.f.n_sfm && c.fused_sfm)
    hipLaunchKernelGGL((k_obs_E<kSfmRec, 9, true>), dim3(grid1(c.f.n_sfm / 4 + 1)), dim3(kBlock), 0, c.stream, c.f.n_sfm, c.f.sfm_J.p, st,
                       c.obs_lm.p, c.Linv.p, c.ylm.p, c.E.p, c.wobs.p, c.obs_wpos.p);
  else if (c.f.n_sfm)
    hipLaunchKernelGGL((k_obs_E<kSfmRec, 9, false>), dim3(grid1(c.f.n_sfm / 4 + 1)), dim3(kBlock), 0, c.stream, c.f.n_sfm, c.f.sfm_J.p, st,
                       c.obs_lm.p, c.Linv.p, c.ylm.p, c.E.p, c.wobs.p, c.obs_wpos.p);
  if (c.f.n_proj)
    hipLaunchKernelGGL((k_obs_E<kProjRec, 6, false>), dim3(grid1(c.f.n_proj / 4 + 1)), dim3(kBlock), 0, c.stream, c.f.n_proj,
                       c.f.proj_J.p, st, c.obs_lm.p + c.f.n_sfm, c.Linv.p, c.ylm.p, c.E.p + (int64_t)kEStride * c.f.n_sfm, c.wobs.p, c.obs_wpos.p + c.f.n_sfm);
  check_hip(hipGetLastError(), "point_eliminate");
}

void launch_build_reduced(gtg_context& c, double lambda, int diag, double dmin, double dmax) {
  const double is = inv_sigma(lambda);
  const SMat S = smat(c);
  launch_zero_tiles(c, S, c.plan);
  if (c.n_red_vars)
    hipLaunchKernelGGL(k_build_diag, dim3(c.n_red_vars), dim3(64 * kDiagWaves), 0, c.stream, c.n_red_vars, c.red_inc_ptr.p,
                       c.red_inc_kind.p, c.red_inc_idx.p, c.red_dim.p, c.red_off.p, c.f.n_sfm, c.Hd.p,
                       c.gred0.p, c.hdiag_red.p, c.wobs.p, is, diag, dmin, dmax, c.shard == 0 ? 1 : 0, S);
  if (c.n_hoff)
    hipLaunchKernelGGL(k_scatter_hoff, dim3((unsigned)c.n_hoff), dim3(64), 0, c.stream, c.n_hoff, c.hoff_row.p,
                       c.hoff_col.p, c.red_dim.p, c.red_off.p, c.Hoff.p, S);
  if (c.n_pairs && c.n_pair_terms > 512 * c.n_pairs)   // few pairs, thousands of terms each
    hipLaunchKernelGGL(k_schur_pairs_heavy, dim3((unsigned)c.n_pairs), dim3(64 * kPairWaves), 0, c.stream, c.n_pairs, c.pair_row.p,
                       c.pair_col.p, c.pair_ptr.p, c.pair_oa.p, c.pair_ob.p, c.red_dim.p, c.red_off.p, c.E.p, S);
  else if (c.n_pairs)
    hipLaunchKernelGGL(k_schur_pairs, dim3((unsigned)((c.n_pairs + 3) / 4)), dim3(256), 0, c.stream, c.n_pairs, c.pair_row.p,
                       c.pair_col.p, c.pair_ptr.p, c.pair_oa.p, c.pair_ob.p, c.red_dim.p, c.red_off.p, c.E.p, S);
  const int64_t npad = (int64_t)c.h_pad_index.size();
  if (npad > 0 && c.shard == 0)
    hipLaunchKernelGGL(k_pad_diag, dim3((unsigned)((npad + 63) / 64)), dim3(64), 0, c.stream, S, c.pad_index.p, npad);
  check_hip(hipGetLastError(), "build_reduced");
}

void launch_back_substitute(gtg_context& c) {
  const SfmTabs st = sfm_tabs(c);
  if (c.f.n_sfm && c.n_lm && c.fused_sfm)
    hipLaunchKernelGGL((k_obs_v<kSfmRec, 9, true>), dim3(grid1(c.f.n_sfm / 4 + 1)), dim3(kBlock), 0, c.stream, c.f.n_sfm, c.f.sfm_J.p, st,
                       c.obs_red.p, c.red_off.p, c.xred.p, c.vobs.p);
  else if (c.f.n_sfm && c.n_lm)
    hipLaunchKernelGGL((k_obs_v<kSfmRec, 9, false>), dim3(grid1(c.f.n_sfm / 4 + 1)), dim3(kBlock), 0, c.stream, c.f.n_sfm, c.f.sfm_J.p, st,
                       c.obs_red.p, c.red_off.p, c.xred.p, c.vobs.p);
  if (c.f.n_proj && c.n_lm)
    hipLaunchKernelGGL((k_obs_v<kProjRec, 6, false>), dim3(grid1(c.f.n_proj / 4 + 1)), dim3(kBlock), 0, c.stream, c.f.n_proj,
                       c.f.proj_J.p, st, c.obs_red.p + c.f.n_sfm, c.red_off.p, c.xred.p, c.vobs.p + 3 * c.f.n_sfm);
  if (c.n_lm)
    hipLaunchKernelGGL(k_backsub_lm, dim3(grid1(c.n_lm)), dim3(kBlock), 0, c.stream, c.n_lm, c.lm_owned.p,
                       c.lm_obs_ptr.p, c.lm_obs.p, c.vobs.p, c.Linv.p, c.ylm.p, c.delta_lm.p);
  check_hip(hipGetLastError(), "back_substitute");
}

void launch_scatter_delta(gtg_context& c) {
  hipLaunchKernelGGL(k_scatter_delta, dim3(grid1(c.n_vars)), dim3(kBlock), 0, c.stream, c.n_vars, c.lm_index.p,
                     c.red_index.p, c.red_dim.p, c.red_off.p, c.dim_off.p, c.xred.p, c.delta_lm.p, c.delta.p);
  check_hip(hipGetLastError(), "scatter_delta");
}

// gtg_prewarm: this unit's kernels (kernels.h)
static void prewarm_assemble(int) {
  prewarm_kernels({(const void*)k_red_diag, (const void*)k_cam_fused, (const void*)k_cam_combine, (const void*)k_lm_fused, (const void*)k_lm_diag, (const void*)k_hoff,
                   (const void*)k_point_factor, (const void*)k_obs_E<kSfmRec, 9, true>, (const void*)k_obs_E<kSfmRec, 9, false>, (const void*)k_obs_E<kProjRec, 6, false>,
                   (const void*)k_obs_v<kSfmRec, 9, true>, (const void*)k_obs_v<kSfmRec, 9, false>, (const void*)k_obs_v<kProjRec, 6, false>, (const void*)k_build_diag,
                   (const void*)k_scatter_hoff, (const void*)k_schur_pairs, (const void*)k_schur_pairs_heavy, (const void*)k_pad_diag, (const void*)k_backsub_lm,
                   (const void*)k_scatter_delta, (const void*)k_obs_wpos, (const void*)k_cam_pack, (const void*)k_smart_hdiag, (const void*)k_smart_lin1});
}
static PrewarmUnit prewarm_assemble_registered(prewarm_assemble);

}  // namespace gt
