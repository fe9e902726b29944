// pcg.hip -- block-Jacobi preconditioned conjugate gradients on the IMPLICIT Schur complement of the landmarks.
//
// The alternative to the Cholesky of the reduced system for problems whose camera count makes the n^3 factorisation the
// wall.  The algorithm is the reference's preconditionedConjugateGradient
// (linear/ConjugateGradientSolver.h:106-169: split preconditioning r = L^-1 (b - A x), p = L^-T r, stop when
// |r|^2 <= max(epsilon_abs, epsilon_rel^2 |r0|^2), PCGSolver.cpp:51-64) with a block-Jacobi preconditioner
// (linear/Preconditioner.cpp, BlockJacobiPreconditioner: Cholesky factors of the diagonal blocks), applied to
//
//     S x = b,   S = H_cc + lambda D - sum_landmarks E_l E_l^T,   b = g_c - sum E_l y_l
//
// WITHOUT forming S (the implicit Schur factor of slam/RegularImplicitSchurFactor.h): one product is
//     w_o = E_o^T x_cam(o)          one observation per lane        (E read once, coalesced through LDS)
//     y_l = sum_{o in l} w_o        one landmark per lane
//     out_r = (H_rr + damping) x_r - sum_{o in r} E_o y_l(o) + sum_{between factors (r,s)} A_r^T A_s x_s
//                                   four wavefronts per camera / pose (E read a second time)
// i.e. three streaming kernels, 0.35 GB of traffic for the L1723 shape (676 773 observations x 256-byte E slots, twice).
// The CG scalars stay on the device (no host round trip inside a batch of iterations); the preconditioner is kept as
// the inverse factors L_r^-1 and applied one lane per (variable, row).  All reductions run in a fixed order.
// Vectors live in the layout of the reduced system (offset red_off[r], length NP, alignment gaps stay zero).
#include <cmath>
#include <limits>
#include <stdexcept>

#include "factors.h"
#include "kernels.h"
#include "recio.h"

namespace gt {

namespace {
constexpr int kB = 256;
constexpr int kMaxPart = 2048;
enum { INC_SFM = 0, INC_PROJ = 1, INC_BTW_A = 2, INC_BTW_B = 3, INC_PRIOR = 4 };

inline int grid_n(int64_t n) { int64_t b = (n + kB - 1) / kB; return (int)(b < 1 ? 1 : (b > kMaxPart ? kMaxPart : b)); }

__device__ __forceinline__ double damp(double hjj, double invsigma, int diag, double dmin, double dmax) {
  double a = 1.0;
  if (diag) a = sqrt(fmin(fmax(hjj, dmin), dmax));
  const double w = a * invsigma;
  return w * w;
}

// Loads the 9x3 block of one E slot (rows >= d read as zero) with 16-byte loads.
__device__ __forceinline__ void load_E(const double* __restrict__ Eo, int d, double e[27]) {
  typedef double v2f64 __attribute__((ext_vector_type(2)));
  const v2f64* E2 = reinterpret_cast<const v2f64*>(Eo);
  double raw[28];
#pragma unroll
  for (int q = 0; q < 14; q++) { const v2f64 t = E2[q]; raw[2 * q] = t.x; raw[2 * q + 1] = t.y; }
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int cc = 0; cc < 3; cc++) e[3 * i + cc] = i < d ? raw[3 * i + cc] : 0.0;
}

// Per reduced variable (one workgroup of kSetupWaves wavefronts, the incident observations dealt round-robin to the
// lanes): rhs b_r = g_r - sum_o E_o y_l(o) and block D_r = H_rr + damping - sum_o E_o E_o^T (k_pcg_setup; on a sharded graph
// both are this shard's partial sums -- damping and the unit padding only on the first shard -- and are all-reduced before the
// next step), then its Cholesky factor L_r and the block-Jacobi preconditioner in the form it is applied in: M_r = L_r^-1
// (lower, row-major, stride 9, zero padded) (k_pcg_factor, in place).
constexpr int kSetupWaves = 4;
__global__ __launch_bounds__(64 * kSetupWaves) void k_pcg_setup(int32_t n_red_vars, const int64_t* __restrict__ inc_ptr,
    const int32_t* __restrict__ inc_kind, const int32_t* __restrict__ inc_idx, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, const int32_t* __restrict__ obs_lm, int64_t n_sfm,
    const double* __restrict__ Hd, const double* __restrict__ g, const double* __restrict__ hdiag,
    const double* __restrict__ E, const double* __restrict__ ylm, double invsigma, int diag, double dmin, double dmax,
    int first, double* __restrict__ b, double* __restrict__ Mbj) {
  __shared__ double red[kSetupWaves][54];
  const int r = blockIdx.x;
  if (r >= n_red_vars) return;
  const int d = red_dim[r];
  const int64_t off = red_off[r];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double acc[54];      // lower triangle of sum E E^T (entry i (i + 1) / 2 + j), then the 9 entries of sum E y
#pragma unroll
  for (int q = 0; q < 54; q++) acc[q] = 0.0;
  for (int64_t k = inc_ptr[r] + tid; k < inc_ptr[r + 1]; k += 64 * kSetupWaves) {
    const int kind = inc_kind[k];
    if (kind > INC_PROJ) continue;
    const int64_t o = kind == INC_SFM ? (int64_t)inc_idx[k] : n_sfm + inc_idx[k];
    const double* y = ylm + 3 * (int64_t)obs_lm[o];
    const double y0 = y[0], y1 = y[1], y2 = y[2];
    double e[27];
    load_E(E + kEStride * o, d, e);
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) acc[i * (i + 1) / 2 + j] += e[3 * i] * e[3 * j] + e[3 * i + 1] * e[3 * j + 1] + e[3 * i + 2] * e[3 * j + 2];
      acc[45 + i] += e[3 * i] * y0 + e[3 * i + 1] * y1 + e[3 * i + 2] * y2;
    }
  }
#pragma unroll
  for (int q = 0; q < 54; q++) {
    for (int sft = 32; sft > 0; sft >>= 1) acc[q] += __shfl_down(acc[q], sft, 64);
    if (lane == 0) red[wave][q] = acc[q];
  }
  __syncthreads();
  if (tid < 54) { double t = red[0][tid]; for (int w = 1; w < kSetupWaves; w++) t += red[w][tid]; red[0][tid] = t; }
  __syncthreads();
  if (tid < 81) {
    const int i = tid / 9, j = tid % 9;
    double v = (i == j && first) ? 1.0 : 0.0;
    if (i < d && j < d) {
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      v = Hd[(int64_t)81 * r + i * d + j] + ((i == j && first) ? damp(hdiag[off + i], invsigma, diag, dmin, dmax) : 0.0) - red[0][hi * (hi + 1) / 2 + lo];
    }
    Mbj[(int64_t)81 * r + tid] = v;
  } else if (tid >= 128 && tid < 128 + d) {
    const int i = tid - 128;
    b[off + i] = g[(int64_t)9 * r + i] - red[0][45 + i];
  }
}

__global__ __launch_bounds__(128) void k_pcg_factor(int32_t n_red_vars, const int32_t* __restrict__ red_dim, double* __restrict__ Mbj,
                                                    double* __restrict__ fail) {
  __shared__ double D[81], Ls[81];
  const int r = blockIdx.x, tid = threadIdx.x;
  if (r >= n_red_vars) return;
  const int d = red_dim[r];
  if (tid < 81) { D[tid] = Mbj[(int64_t)81 * r + tid]; Ls[tid] = 0.0; }
  __syncthreads();
  if (tid == 0) {   // d <= 9: serial LLT (Eigen semantics: a non-positive pivot is a failure)
    bool bad = false;
    for (int j = 0; j < d; j++) {
      double pv = D[j * 9 + j];
      for (int m = 0; m < j; m++) pv -= Ls[j * 9 + m] * Ls[j * 9 + m];
      if (!(pv > 0.0)) { bad = true; pv = 1.0; }
      const double ljj = sqrt(pv);
      Ls[j * 9 + j] = ljj;
      for (int i = j + 1; i < d; i++) {
        double v = D[i * 9 + j];
        for (int m = 0; m < j; m++) v -= Ls[i * 9 + m] * Ls[j * 9 + m];
        Ls[i * 9 + j] = v / ljj;
      }
    }
    if (bad) *fail = 1.0;
  }
  __syncthreads();
  if (tid < 9) {    // column tid of L^-1 by forward substitution on a unit vector
    const int j = tid;
    double z[9];
    for (int i = 0; i < 9; i++) z[i] = 0.0;
    if (j < d) {
      z[j] = 1.0 / Ls[j * 9 + j];
      for (int i = j + 1; i < d; i++) {
        double v = 0.0;
        for (int m = j; m < i; m++) v -= Ls[i * 9 + m] * z[m];
        z[i] = v / Ls[i * 9 + i];
      }
    }
    for (int i = 0; i < 9; i++) Mbj[(int64_t)81 * r + i * 9 + j] = z[i];
  }
}

// The CG scalars live on the device so that a batch of iterations is enqueued without a host round trip:
//   st[ST_GAMMA] |r|^2, st[ST_ALPHA], st[ST_BETA], st[ST_THRESHOLD], st[ST_K] number of the NEXT iteration (1-based, as the
//   reference's loop counter), st[ST_DONE] != 0 once the loop condition of ConjugateGradientSolver.h:128-129 fails (every
//   kernel of a later, already enqueued iteration then returns at once), st[ST_GAMMA0].
enum { ST_GAMMA = 0, ST_ALPHA = 1, ST_BETA = 2, ST_THRESHOLD = 3, ST_K = 4, ST_DONE = 5, ST_GAMMA0 = 6, ST_COUNT = 8 };

__device__ __forceinline__ double block_sum(double acc, double* sm) {   // fixed-order sum over the kB lanes of a workgroup
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kB / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
  return sm[0];
}

// The preconditioner kernels run one lane per (reduced variable, row): M = L^-1 is zero padded to 9 x 9, so a row of
// M x or M^T x is nine independent multiply-adds (a triangular solve per lane was a chain of 45 dependent loads).
__device__ __forceinline__ void load_vec(const double* __restrict__ x, int d, double v[9]) {
#pragma unroll
  for (int j = 0; j < 9; j++) v[j] = j < d ? x[j] : 0.0;
}
__device__ __forceinline__ double row_of_M(const double* __restrict__ M, int i, const double v[9]) {     // (M v)_i
  double t = 0.0;
#pragma unroll
  for (int j = 0; j < 9; j++) t += M[i * 9 + j] * v[j];
  return t;
}
__device__ __forceinline__ double row_of_Mt(const double* __restrict__ M, int i, const double v[9]) {    // (M^T v)_i
  double t = 0.0;
#pragma unroll
  for (int j = 0; j < 9; j++) t += M[j * 9 + i] * v[j];
  return t;
}

// start (x0 = 0): r = L^-1 b, p = L^-T r, partial sums of |r|^2
__global__ __launch_bounds__(kB) void k_pcg_start(int32_t n_red_vars, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, const double* __restrict__ Mbj, const double* __restrict__ b, double* __restrict__ r,
    double* __restrict__ p, double* __restrict__ partials) {
  __shared__ double sm[kB];
  double acc = 0.0;
  for (int64_t gl = blockIdx.x * (int64_t)kB + threadIdx.x; gl < 9 * (int64_t)n_red_vars; gl += (int64_t)gridDim.x * kB) {
    const int64_t v = gl / 9;
    const int i = (int)(gl - 9 * v), d = red_dim[v];
    if (i >= d) continue;
    const int64_t off = red_off[v];
    const double* M = Mbj + 81 * v;
    double bv[9], rv[9];
    load_vec(b + off, d, bv);
#pragma unroll
    for (int j = 0; j < 9; j++) rv[j] = row_of_M(M, j, bv);     // the whole L^-1 b of this variable (p needs all of it)
    double ri = 0.0;
#pragma unroll
    for (int j = 0; j < 9; j++) ri = j == i ? rv[j] : ri;
    r[off + i] = ri;
    p[off + i] = row_of_Mt(M, i, rv);
    acc += ri * ri;
  }
  const double t = block_sum(acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// one workgroup: sum of n partials in a fixed order, then the scalar bookkeeping of the loop
//   mode 0: gamma0 = gamma = sum, threshold, k = 1, done            (ConjugateGradientSolver.h:119-126)
//   mode 1: alpha = gamma / sum (sum = p . A p)                      (:131-132)
//   mode 2: beta = sum / gamma, gamma = sum, k += 1, done            (:136-139 and the loop condition :128-129)
__global__ __launch_bounds__(kB) void k_pcg_scalar(int mode, const double* __restrict__ partials, int n, double* __restrict__ st,
    double max_iterations, double min_iterations, double epsilon_rel, double epsilon_abs) {
  __shared__ double sm[kB];
  if (mode != 0 && st[ST_DONE] != 0.0) return;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += kB) acc += partials[i];
  const double sum = block_sum(acc, sm);
  if (threadIdx.x != 0) return;
  if (mode == 1) { st[ST_ALPHA] = st[ST_GAMMA] / sum; return; }
  double k;
  if (mode == 0) {
    st[ST_GAMMA0] = sum;
    st[ST_THRESHOLD] = fmax(epsilon_abs, epsilon_rel * epsilon_rel * sum);
    k = 1.0;
  } else {
    st[ST_BETA] = sum / st[ST_GAMMA];
    k = st[ST_K] + 1.0;
  }
  st[ST_GAMMA] = sum;
  st[ST_K] = k;
  const bool go = k <= max_iterations && (sum > st[ST_THRESHOLD] || k <= min_iterations) && isfinite(sum);
  st[ST_DONE] = go ? 0.0 : 1.0;
}

// x += alpha p, r -= alpha L^-1 q, partial sums of the new |r|^2
__global__ __launch_bounds__(kB) void k_pcg_update_xr(int32_t n_red_vars, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, const double* __restrict__ Mbj, const double* __restrict__ st, const double* __restrict__ p,
    const double* __restrict__ q, double* __restrict__ x, double* __restrict__ r, double* __restrict__ partials) {
  __shared__ double sm[kB];
  if (st[ST_DONE] != 0.0) return;
  const double alpha = st[ST_ALPHA];
  double acc = 0.0;
  for (int64_t gl = blockIdx.x * (int64_t)kB + threadIdx.x; gl < 9 * (int64_t)n_red_vars; gl += (int64_t)gridDim.x * kB) {
    const int64_t v = gl / 9;
    const int i = (int)(gl - 9 * v), d = red_dim[v];
    if (i >= d) continue;
    const int64_t off = red_off[v];
    double qv[9];
    load_vec(q + off, d, qv);
    const double t = row_of_M(Mbj + 81 * v, i, qv);
    x[off + i] += alpha * p[off + i];
    const double ri = r[off + i] - alpha * t;
    r[off + i] = ri;
    acc += ri * ri;
  }
  const double t = block_sum(acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// p = L^-T r + beta p
__global__ __launch_bounds__(kB) void k_pcg_update_p(int32_t n_red_vars, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, const double* __restrict__ Mbj, const double* __restrict__ st, const double* __restrict__ r,
    double* __restrict__ p) {
  if (st[ST_DONE] != 0.0) return;
  const double beta = st[ST_BETA];
  for (int64_t gl = blockIdx.x * (int64_t)kB + threadIdx.x; gl < 9 * (int64_t)n_red_vars; gl += (int64_t)gridDim.x * kB) {
    const int64_t v = gl / 9;
    const int i = (int)(gl - 9 * v), d = red_dim[v];
    if (i >= d) continue;
    const int64_t off = red_off[v];
    double rv[9];
    load_vec(r + off, d, rv);
    p[off + i] = row_of_Mt(Mbj + 81 * v, i, rv) + beta * p[off + i];
  }
}

// w_o = E_o^T x_cam(o): one observation per lane, the wavefront's 64 E slots through the LDS image
__global__ __launch_bounds__(kB) void k_pcg_obs(int64_t n, int dc, const double* __restrict__ E, const int32_t* __restrict__ obs_red,
    const int64_t* __restrict__ red_off, const double* __restrict__ st, const double* __restrict__ x, double* __restrict__ w) {
  typedef RecIO<kEStride> IN;
  if (st[ST_DONE] != 0.0) return;
  __shared__ double img[kB / 64][IN::LDS_DOUBLES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* my = img[wave];
  const int64_t nchunks = (n + 63) / 64, stride = (int64_t)gridDim.x * (kB / 64);
  for (int64_t ch = blockIdx.x * (int64_t)(kB / 64) + wave; ch < nchunks; ch += stride) {
    const int64_t o = ch * 64 + lane, left = n - ch * 64;
    IN::load(my, E + (int64_t)kEStride * ch * 64, left < 64 ? (int)left : 64, lane);
    if (o < n) {
      const double* Eo = my + lane * IN::PITCH;
      const double* xr = x + red_off[obs_red[o]];
      double w0 = 0.0, w1 = 0.0, w2 = 0.0;
      for (int i = 0; i < dc; i++) { w0 += Eo[3 * i] * xr[i]; w1 += Eo[3 * i + 1] * xr[i]; w2 += Eo[3 * i + 2] * xr[i]; }
      w[3 * o] = w0; w[3 * o + 1] = w1; w[3 * o + 2] = w2;
    }
  }
}

__global__ __launch_bounds__(kB) void k_pcg_lm(int32_t n_lm, const int64_t* __restrict__ obs_ptr, const int32_t* __restrict__ obs,
    const double* __restrict__ st, const double* __restrict__ w, double* __restrict__ y) {
  if (st[ST_DONE] != 0.0) return;
  for (int64_t l = blockIdx.x * (int64_t)kB + threadIdx.x; l < n_lm; l += (int64_t)gridDim.x * kB) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const int64_t k1 = obs_ptr[l + 1];
    for (int64_t k = obs_ptr[l]; k < k1; k += 4) {        // four independent gathers in flight, summed in list order
      int64_t o[4];
      double t[4][3];
#pragma unroll
      for (int u = 0; u < 4; u++) o[u] = k + u < k1 ? (int64_t)obs[k + u] : -1;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double* wo = w + 3 * (o[u] < 0 ? 0 : o[u]);
        t[u][0] = wo[0]; t[u][1] = wo[1]; t[u][2] = wo[2];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) if (o[u] >= 0) { s0 += t[u][0]; s1 += t[u][1]; s2 += t[u][2]; }
    }
    y[3 * l] = s0; y[3 * l + 1] = s1; y[3 * l + 2] = s2;
  }
}

struct ApplyArgs {
  const int64_t* inc_ptr; const int32_t *inc_kind, *inc_idx, *red_dim; const int64_t* red_off;
  const int32_t *obs_lm, *red_index, *bt_v1, *bt_v2; int64_t n_sfm;
  const double *Hd, *hdiag, *E, *bt_J;
};
// out_r = (H_rr + damping) x_r - sum_o E_o y_l(o) + sum_between A_r^T A_s x_s and partials[r] = x_r . out_r ;
// one workgroup of kApplyWaves wavefronts per reduced variable (the incidence list dealt round-robin to the lanes; a
// camera of the L1723 shape has 394 observations on average, thousands at most)
constexpr int kApplyWaves = 4;
__global__ __launch_bounds__(64 * kApplyWaves) void k_pcg_apply(int32_t n_red_vars, ApplyArgs a, double invsigma, int diag, double dmin,
    double dmax, int first, const double* __restrict__ st, const double* __restrict__ ylm, const double* __restrict__ x,
    double* __restrict__ out, double* __restrict__ partials) {
  __shared__ double red[kApplyWaves][9];
  __shared__ double pq[9];
  const int r = blockIdx.x;
  if (r >= n_red_vars || st[ST_DONE] != 0.0) return;
  const int d = a.red_dim[r];
  const int64_t off = a.red_off[r];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t k = a.inc_ptr[r] + tid; k < a.inc_ptr[r + 1]; k += 64 * kApplyWaves) {
    const int kind = a.inc_kind[k];
    if (kind <= INC_PROJ) {
      const int64_t o = kind == INC_SFM ? (int64_t)a.inc_idx[k] : a.n_sfm + a.inc_idx[k];
      const double* y = ylm + 3 * (int64_t)a.obs_lm[o];
      const double y0 = y[0], y1 = y[1], y2 = y[2];
      double e[27];
      load_E(a.E + kEStride * o, d, e);
#pragma unroll
      for (int i = 0; i < 9; i++) acc[i] -= e[3 * i] * y0 + e[3 * i + 1] * y1 + e[3 * i + 2] * y2;
    } else if (kind == INC_BTW_A || kind == INC_BTW_B) {
      const int f = a.inc_idx[k];
      const double* J = a.bt_J + (int64_t)kBetweenRec * f;
      const double* Ar = kind == INC_BTW_A ? J : J + 36;
      const double* As = kind == INC_BTW_A ? J + 36 : J;
      const int vs = kind == INC_BTW_A ? a.bt_v2[f] : a.bt_v1[f];
      const double* xs = x + a.red_off[a.red_index[vs]];
      double t[9];
      for (int q = 0; q < d; q++) { double sacc = 0.0; for (int m = 0; m < d; m++) sacc += As[q * d + m] * xs[m]; t[q] = sacc; }
      for (int i = 0; i < d; i++) { double sacc = 0.0; for (int q = 0; q < d; q++) sacc += Ar[q * d + i] * t[q]; acc[i] += sacc; }
    }
  }
#pragma unroll
  for (int i = 0; i < 9; i++) {
    for (int sft = 32; sft > 0; sft >>= 1) acc[i] += __shfl_down(acc[i], sft, 64);
    if (lane == 0) red[wave][i] = acc[i];
  }
  __syncthreads();
  if (tid < 9) {
    double sacc = 0.0;
    if (tid < d) {
      const int i = tid;
      for (int w = 0; w < kApplyWaves; w++) sacc += red[w][i];
      if (first) sacc += damp(a.hdiag[off + i], invsigma, diag, dmin, dmax) * x[off + i];
      const double* H = a.Hd + (int64_t)81 * r;
      for (int j = 0; j < d; j++) sacc += H[i * d + j] * x[off + j];
      out[off + i] = sacc;
      sacc *= x[off + i];
    }
    pq[tid] = sacc;
  }
  __syncthreads();
  if (tid == 0) { double t = 0.0; for (int i = 0; i < 9; i++) t += pq[i]; partials[r] = t; }
}

// sharded: x . out per reduced variable once the partial products have been summed over the shards (same order as k_pcg_apply)
__global__ __launch_bounds__(kB) void k_pcg_dot(int32_t n_red_vars, const int32_t* __restrict__ red_dim, const int64_t* __restrict__ red_off,
                                                const double* __restrict__ st, const double* __restrict__ x, const double* __restrict__ out,
                                                double* __restrict__ partials) {
  if (st[ST_DONE] != 0.0) return;
  for (int64_t r = blockIdx.x * (int64_t)kB + threadIdx.x; r < n_red_vars; r += (int64_t)gridDim.x * kB) {
    const int d = red_dim[r];
    const int64_t off = red_off[r];
    double t = 0.0;
    for (int i = 0; i < d; i++) t += out[off + i] * x[off + i];
    partials[r] = t;
  }
}
}  // namespace

// Solves the reduced system into c.xred.  Returns the number of CG iterations; *gamma0 / *gamma = |r|^2 of the
// preconditioned residual at the start / end.  Seven launches per iteration, enqueued in batches of kBatch iterations with one
// read-back of the scalars per batch (the iterations enqueued past convergence return at once): the result is exactly
// that of checking the loop condition on the host every iteration.
// Sharded graph (landmarks dealt to the shards, every shard holds all cameras): a shard's factors give PARTIAL sums of b, of the
// diagonal blocks and of every product S p, so these are all-reduced (one exchange of NP doubles per product: 124 KB for the
// L1723 / Venice shapes, next to 0.35 GB / n_shards of streaming per shard); everything else -- the preconditioner, the vector
// updates, the scalars -- is computed redundantly and identically on every shard, so the shards stay in lock step without a
// further exchange and leave the loop in the same iteration.
int launch_pcg(gtg_context& c, double lambda, int diag, double dmin, double dmax, int max_iterations, int min_iterations,
               double epsilon_rel, double epsilon_abs, double* gamma0, double* gamma_end) {
  const bool sharded = c.n_shards > 1;
  const int first = c.shard == 0 ? 1 : 0;
  constexpr int kBatch = 8;
  const int NP = c.NP, nrv = c.n_red_vars;
  const double is = std::sqrt(lambda);   // 1 / sigma with sigma = 1 / sqrt(lambda) (LMState.h:117-121)
  hipStream_t s = c.stream;
  const size_t n_part = (size_t)std::max(nrv, kMaxPart);
  const size_t n_vec = 4 * (size_t)NP + ST_COUNT + n_part;
  if (c.pcg_vec.n != n_vec) c.pcg_vec.alloc(n_vec);
  if ((int64_t)c.pcg_bj.n != 81 * (int64_t)std::max(nrv, 1)) c.pcg_bj.alloc(81 * (size_t)std::max(nrv, 1));
  if ((int64_t)c.pcg_y.n != 3 * (int64_t)std::max(c.n_lm, 1)) c.pcg_y.alloc(3 * (size_t)std::max(c.n_lm, 1));
  double *x = c.xred.p, *r = c.pcg_vec.p, *p = r + NP, *q = p + NP, *b = q + NP, *st = b + NP, *partials = st + ST_COUNT;
  check_hip(hipMemsetAsync(c.xred.p, 0, sizeof(double) * NP, s), "memset");
  check_hip(hipMemsetAsync(c.pcg_vec.p, 0, sizeof(double) * n_vec, s), "memset");
  const int gv = grid_n(9 * (int64_t)nrv);
  hipLaunchKernelGGL(k_pcg_setup, dim3(std::max(nrv, 1)), dim3(64 * kSetupWaves), 0, s, nrv, c.red_inc_ptr.p, c.red_inc_kind.p, c.red_inc_idx.p,
                     c.red_dim.p, c.red_off.p, c.obs_lm.p, c.f.n_sfm, c.Hd.p, c.gred0.p, c.hdiag_red.p, c.E.p, c.ylm.p, is, diag,
                     dmin, dmax, first, b, c.pcg_bj.p);
  if (sharded) { exchange_sum(c, b, NP); exchange_sum(c, c.pcg_bj.p, 81 * (int64_t)nrv); }
  hipLaunchKernelGGL(k_pcg_factor, dim3(std::max(nrv, 1)), dim3(128), 0, s, nrv, c.red_dim.p, c.pcg_bj.p, c.scalars.p + SC_FAIL);
  ApplyArgs aa{c.red_inc_ptr.p, c.red_inc_kind.p, c.red_inc_idx.p, c.red_dim.p, c.red_off.p, c.obs_lm.p, c.red_index.p,
               c.f.between_v1.p, c.f.between_v2.p, c.f.n_sfm, c.Hd.p, c.hdiag_red.p, c.E.p, c.f.between_J.p};
  const double dmax_it = (double)max_iterations, dmin_it = (double)min_iterations;
  // x0 = 0: the residual is b (ConjugateGradientSolver.h:113-117)
  hipLaunchKernelGGL(k_pcg_start, dim3(gv), dim3(kB), 0, s, nrv, c.red_dim.p, c.red_off.p, c.pcg_bj.p, b, r, p, partials);
  hipLaunchKernelGGL(k_pcg_scalar, dim3(1), dim3(kB), 0, s, 0, partials, gv, st, dmax_it, dmin_it, epsilon_rel, epsilon_abs);
  double h[ST_COUNT] = {0};
  auto iteration = [&]() {
    if (c.f.n_sfm)                                   // q = S p
      hipLaunchKernelGGL(k_pcg_obs, dim3(grid_n(c.f.n_sfm / 4 + 1)), dim3(kB), 0, s, c.f.n_sfm, 9, c.E.p, c.obs_red.p, c.red_off.p, st, p, c.vobs.p);
    if (c.f.n_proj)
      hipLaunchKernelGGL(k_pcg_obs, dim3(grid_n(c.f.n_proj / 4 + 1)), dim3(kB), 0, s, c.f.n_proj, 6, c.E.p + (int64_t)kEStride * c.f.n_sfm,
                         c.obs_red.p + c.f.n_sfm, c.red_off.p, st, p, c.vobs.p + 3 * c.f.n_sfm);
    if (c.n_lm)
      hipLaunchKernelGGL(k_pcg_lm, dim3(grid_n(c.n_lm)), dim3(kB), 0, s, c.n_lm, c.lm_obs_ptr.p, c.lm_obs.p, st, c.vobs.p, c.pcg_y.p);
    hipLaunchKernelGGL(k_pcg_apply, dim3(std::max(nrv, 1)), dim3(64 * kApplyWaves), 0, s, nrv, aa, is, diag, dmin, dmax, first, st, c.pcg_y.p, p, q, partials);
    if (sharded) {
      exchange_sum(c, q, NP);
      hipLaunchKernelGGL(k_pcg_dot, dim3(grid_n(nrv)), dim3(kB), 0, s, nrv, c.red_dim.p, c.red_off.p, st, p, q, partials);
    }
    hipLaunchKernelGGL(k_pcg_scalar, dim3(1), dim3(kB), 0, s, 1, partials, nrv, st, dmax_it, dmin_it, epsilon_rel, epsilon_abs);   // alpha
    hipLaunchKernelGGL(k_pcg_update_xr, dim3(gv), dim3(kB), 0, s, nrv, c.red_dim.p, c.red_off.p, c.pcg_bj.p, st, p, q, x, r, partials);
    hipLaunchKernelGGL(k_pcg_scalar, dim3(1), dim3(kB), 0, s, 2, partials, gv, st, dmax_it, dmin_it, epsilon_rel, epsilon_abs);    // beta, gamma, k, done
    hipLaunchKernelGGL(k_pcg_update_p, dim3(gv), dim3(kB), 0, s, nrv, c.red_dim.p, c.red_off.p, c.pcg_bj.p, st, r, p);
  };
  for (int issued = 0;; ) {
    check_hip(hipMemcpyAsync(h, st, sizeof(h), hipMemcpyDeviceToHost, s), "D2H");
    check_hip(hipStreamSynchronize(s), "sync");
    if (h[ST_DONE] != 0.0 || issued >= max_iterations) break;
    for (int i = 0; i < kBatch && issued < max_iterations; i++, issued++) iteration();
  }
  *gamma0 = h[ST_GAMMA0];
  *gamma_end = h[ST_GAMMA];
  check_hip(hipGetLastError(), "pcg");
  return (int)h[ST_K] - 1;
}

// gtg_prewarm: loads this unit's code object (the solver is opt-in: its other kernels pay their first launch)
static void prewarm_pcg(int) { prewarm_kernels({(const void*)k_pcg_setup}); }
static PrewarmUnit prewarm_pcg_registered(prewarm_pcg);

}  // namespace gt
