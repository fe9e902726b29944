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
//                                   one wavefront per camera / pose (E read a second time)
// i.e. three streaming kernels, 0.35 GB of traffic for the L1723 shape.  All reductions run in a fixed order.
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

// per reduced variable (one wavefront): rhs b_r = g_r - sum_o E_o y_l(o), block D_r = H_rr + damping - sum_o E_o E_o^T,
// its Cholesky factor L_r (lower, row-major, stride 81) = the block-Jacobi preconditioner
__global__ __launch_bounds__(64) void k_pcg_setup(int32_t n_red_vars, const int64_t* __restrict__ inc_ptr,
    const int32_t* __restrict__ inc_kind, const int32_t* __restrict__ inc_idx, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, const int32_t* __restrict__ obs_lm, int64_t n_sfm,
    const double* __restrict__ Hd, const double* __restrict__ g, const double* __restrict__ hdiag,
    const double* __restrict__ E, const double* __restrict__ ylm, double invsigma, int diag, double dmin, double dmax,
    double* __restrict__ b, double* __restrict__ Lbj, double* __restrict__ fail) {
  __shared__ double D[81];
  const int r = blockIdx.x;
  if (r >= n_red_vars) return;
  const int d = red_dim[r];
  const int64_t off = red_off[r];
  const int lane = threadIdx.x;
  const int e0 = lane, e1 = lane + 64;
  const int i0 = e0 / d, j0 = e0 % d, i1 = e1 / d, j1 = e1 % d;
  double s0 = 0.0, s1 = 0.0;            // entries e0, e1 of sum_o E_o E_o^T (every lane walks the whole list)
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t k = inc_ptr[r]; k < inc_ptr[r + 1]; k++) {
    const int kind = inc_kind[k];
    if (kind > INC_PROJ) continue;
    const int64_t o = kind == INC_SFM ? (int64_t)inc_idx[k] : n_sfm + inc_idx[k];
    const double* Eo = E + kEStride * o;
    if (e0 < d * d) s0 += Eo[3 * i0] * Eo[3 * j0] + Eo[3 * i0 + 1] * Eo[3 * j0 + 1] + Eo[3 * i0 + 2] * Eo[3 * j0 + 2];
    if (e1 < d * d) s1 += Eo[3 * i1] * Eo[3 * j1] + Eo[3 * i1 + 1] * Eo[3 * j1 + 1] + Eo[3 * i1 + 2] * Eo[3 * j1 + 2];
    if (lane < d) {
      const double* y = ylm + 3 * (int64_t)obs_lm[o];
      acc[0] += Eo[3 * lane] * y[0] + Eo[3 * lane + 1] * y[1] + Eo[3 * lane + 2] * y[2];
    }
  }
  if (lane < d) b[off + lane] = g[(int64_t)9 * r + lane] - acc[0];
  if (e0 < d * d) D[e0] = Hd[(int64_t)81 * r + e0] + (i0 == j0 ? damp(hdiag[off + i0], invsigma, diag, dmin, dmax) : 0.0) - s0;
  if (e1 < d * d) D[e1] = Hd[(int64_t)81 * r + e1] + (i1 == j1 ? damp(hdiag[off + i1], invsigma, diag, dmin, dmax) : 0.0) - s1;
  __syncthreads();
  if (lane == 0) {   // d <= 9: serial LLT (Eigen semantics: a non-positive pivot is a failure)
    double* L = Lbj + (int64_t)81 * r;
    bool bad = false;
    for (int j = 0; j < d; j++) {
      double p = D[j * d + j];
      for (int m = 0; m < j; m++) p -= L[j * d + m] * L[j * d + m];
      if (!(p > 0.0)) { bad = true; p = 1.0; }
      const double ljj = sqrt(p);
      L[j * d + j] = ljj;
      for (int i = j + 1; i < d; i++) {
        double v = D[i * d + j];
        for (int m = 0; m < j; m++) v -= L[i * d + m] * L[j * d + m];
        L[i * d + j] = v / ljj;
      }
      for (int c = j + 1; c < d; c++) L[j * d + c] = 0.0;
    }
    if (bad) *fail = 1.0;
  }
}

// out_r = L_r^-1 in_r (mode 0) or L_r^-T in_r (mode 1); one lane per reduced variable
__global__ __launch_bounds__(kB) void k_pcg_precond(int32_t n_red_vars, int mode, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, const double* __restrict__ Lbj, const double* __restrict__ in,
    double* __restrict__ out) {
  for (int64_t r = blockIdx.x * (int64_t)kB + threadIdx.x; r < n_red_vars; r += (int64_t)gridDim.x * kB) {
    const int d = red_dim[r];
    const double* L = Lbj + 81 * r;
    const double* x = in + red_off[r];
    double* y = out + red_off[r];
    double v[9];
    if (mode == 0) {
      for (int i = 0; i < d; i++) {
        double s = x[i];
        for (int m = 0; m < i; m++) s -= L[i * d + m] * v[m];
        v[i] = s / L[i * d + i];
      }
    } else {
      for (int i = d - 1; i >= 0; i--) {
        double s = x[i];
        for (int m = i + 1; m < d; m++) s -= L[m * d + i] * v[m];
        v[i] = s / L[i * d + i];
      }
    }
    for (int i = 0; i < d; i++) y[i] = v[i];
  }
}

// w_o = E_o^T x_cam(o): one observation per lane, the wavefront's 64 E slots through the LDS image
__global__ __launch_bounds__(kB) void k_pcg_obs(int64_t n, int dc, const double* __restrict__ E, const int32_t* __restrict__ obs_red,
    const int64_t* __restrict__ red_off, const double* __restrict__ x, double* __restrict__ w) {
  typedef RecIO<kEStride> IN;
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
    const double* __restrict__ w, double* __restrict__ y) {
  for (int64_t l = blockIdx.x * (int64_t)kB + threadIdx.x; l < n_lm; l += (int64_t)gridDim.x * kB) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t k = obs_ptr[l]; k < obs_ptr[l + 1]; k++) { const double* wo = w + 3 * (int64_t)obs[k]; s0 += wo[0]; s1 += wo[1]; s2 += wo[2]; }
    y[3 * l] = s0; y[3 * l + 1] = s1; y[3 * l + 2] = s2;
  }
}

struct ApplyArgs {
  const int64_t* inc_ptr; const int32_t *inc_kind, *inc_idx, *red_dim; const int64_t* red_off;
  const int32_t *obs_lm, *red_index, *bt_v1, *bt_v2; int64_t n_sfm;
  const double *Hd, *hdiag, *E, *bt_J;
};
// out_r = (H_rr + damping) x_r - sum_o E_o y_l(o) + sum_between A_r^T A_s x_s ; one wavefront per reduced variable
__global__ __launch_bounds__(64) void k_pcg_apply(int32_t n_red_vars, ApplyArgs a, double invsigma, int diag, double dmin,
    double dmax, const double* __restrict__ ylm, const double* __restrict__ x, double* __restrict__ out) {
  const int r = blockIdx.x;
  if (r >= n_red_vars) return;
  const int d = a.red_dim[r];
  const int64_t off = a.red_off[r];
  const int lane = threadIdx.x;
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t k = a.inc_ptr[r] + lane; k < a.inc_ptr[r + 1]; k += 64) {
    const int kind = a.inc_kind[k];
    if (kind <= INC_PROJ) {
      const int64_t o = kind == INC_SFM ? (int64_t)a.inc_idx[k] : a.n_sfm + a.inc_idx[k];
      const double* Eo = a.E + kEStride * o;
      const double* y = ylm + 3 * (int64_t)a.obs_lm[o];
      for (int i = 0; i < d; i++) acc[i] -= Eo[3 * i] * y[0] + Eo[3 * i + 1] * y[1] + Eo[3 * i + 2] * y[2];
    } else if (kind == INC_BTW_A || kind == INC_BTW_B) {
      const int f = a.inc_idx[k];
      const double* J = a.bt_J + (int64_t)kBetweenRec * f;
      const double* Ar = kind == INC_BTW_A ? J : J + 36;
      const double* As = kind == INC_BTW_A ? J + 36 : J;
      const int vs = kind == INC_BTW_A ? a.bt_v2[f] : a.bt_v1[f];
      const double* xs = x + a.red_off[a.red_index[vs]];
      double t[9];
      for (int q = 0; q < d; q++) { double s = 0.0; for (int m = 0; m < d; m++) s += As[q * d + m] * xs[m]; t[q] = s; }
      for (int i = 0; i < d; i++) { double s = 0.0; for (int q = 0; q < d; q++) s += Ar[q * d + i] * t[q]; acc[i] += s; }
    }
  }
  for (int i = 0; i < 9; i++)
    for (int s = 32; s > 0; s >>= 1) acc[i] += __shfl_down(acc[i], s, 64);
  if (lane == 0) {
    const double* H = a.Hd + (int64_t)81 * r;
    for (int i = 0; i < d; i++) {
      double s = acc[i] + damp(a.hdiag[off + i], invsigma, diag, dmin, dmax) * x[off + i];
      for (int j = 0; j < d; j++) s += H[i * d + j] * x[off + j];
      out[off + i] = s;
    }
  }
}

// y = a x + b y  /  partial sums of x . y, fixed order
__global__ __launch_bounds__(kB) void k_pcg_axpby(int64_t n, double a, const double* __restrict__ x, double b, double* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)kB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kB) y[i] = a * x[i] + b * y[i];
}
__global__ __launch_bounds__(kB) void k_pcg_dot(int64_t n, const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ partials) {
  __shared__ double sm[kB];
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)kB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kB) acc += x[i] * y[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kB / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) partials[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(kB) void k_pcg_dot_final(const double* __restrict__ partials, int n, double* __restrict__ out) {
  __shared__ double sm[kB];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += kB) acc += partials[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kB / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) *out = sm[0];
}
}  // namespace

// Solves the reduced system into c.xred.  Returns the number of CG iterations; *gamma0 / *gamma = |r|^2 of the
// preconditioned residual at the start / end.
int launch_pcg(gtg_context& c, double lambda, int diag, double dmin, double dmax, int max_iterations, int min_iterations,
               double epsilon_rel, double epsilon_abs, double* gamma0, double* gamma_end) {
  if (c.n_shards > 1) throw std::invalid_argument("the PCG solver does not support a sharded graph (one exchange per product)");
  const int NP = c.NP, nrv = c.n_red_vars;
  const double is = std::sqrt(lambda);   // 1 / sigma with sigma = 1 / sqrt(lambda) (LMState.h:117-121)
  hipStream_t s = c.stream;
  if ((int64_t)c.pcg_vec.n != 5 * (int64_t)NP) { c.pcg_vec.alloc(5 * (size_t)NP); }
  if ((int64_t)c.pcg_bj.n != 81 * (int64_t)std::max(nrv, 1)) c.pcg_bj.alloc(81 * (size_t)std::max(nrv, 1));
  if ((int64_t)c.pcg_y.n != 3 * (int64_t)std::max(c.n_lm, 1)) c.pcg_y.alloc(3 * (size_t)std::max(c.n_lm, 1));
  double *x = c.xred.p, *r = c.pcg_vec.p, *p = r + NP, *q1 = p + NP, *q2 = q1 + NP, *b = q2 + NP;
  check_hip(hipMemsetAsync(c.xred.p, 0, sizeof(double) * NP, s), "memset");
  check_hip(hipMemsetAsync(c.pcg_vec.p, 0, sizeof(double) * 5 * (size_t)NP, s), "memset");
  hipLaunchKernelGGL(k_pcg_setup, dim3(std::max(nrv, 1)), dim3(64), 0, s, nrv, c.red_inc_ptr.p, c.red_inc_kind.p, c.red_inc_idx.p,
                     c.red_dim.p, c.red_off.p, c.obs_lm.p, c.f.n_sfm, c.Hd.p, c.gred0.p, c.hdiag_red.p, c.E.p, c.ylm.p, is, diag,
                     dmin, dmax, b, c.pcg_bj.p, c.scalars.p + SC_FAIL);
  ApplyArgs aa{c.red_inc_ptr.p, c.red_inc_kind.p, c.red_inc_idx.p, c.red_dim.p, c.red_off.p, c.obs_lm.p, c.red_index.p,
               c.f.between_v1.p, c.f.between_v2.p, c.f.n_sfm, c.Hd.p, c.hdiag_red.p, c.E.p, c.f.between_J.p};
  auto multiply = [&](const double* in, double* out) {   // out = S in
    if (c.f.n_sfm)
      hipLaunchKernelGGL(k_pcg_obs, dim3(grid_n(c.f.n_sfm / 4 + 1)), dim3(kB), 0, s, c.f.n_sfm, 9, c.E.p, c.obs_red.p, c.red_off.p, in, c.vobs.p);
    if (c.f.n_proj)
      hipLaunchKernelGGL(k_pcg_obs, dim3(grid_n(c.f.n_proj / 4 + 1)), dim3(kB), 0, s, c.f.n_proj, 6, c.E.p + (int64_t)kEStride * c.f.n_sfm,
                         c.obs_red.p + c.f.n_sfm, c.red_off.p, in, c.vobs.p + 3 * c.f.n_sfm);
    if (c.n_lm)
      hipLaunchKernelGGL(k_pcg_lm, dim3(grid_n(c.n_lm)), dim3(kB), 0, s, c.n_lm, c.lm_obs_ptr.p, c.lm_obs.p, c.vobs.p, c.pcg_y.p);
    hipLaunchKernelGGL(k_pcg_apply, dim3(std::max(nrv, 1)), dim3(64), 0, s, nrv, aa, is, diag, dmin, dmax, c.pcg_y.p, in, out);
  };
  auto precond = [&](int mode, const double* in, double* out) {
    hipLaunchKernelGGL(k_pcg_precond, dim3(grid_n(nrv)), dim3(kB), 0, s, nrv, mode, c.red_dim.p, c.red_off.p, c.pcg_bj.p, in, out);
  };
  auto dot = [&](const double* u, const double* v) {
    const int g = grid_n(NP);
    hipLaunchKernelGGL(k_pcg_dot, dim3(g), dim3(kB), 0, s, (int64_t)NP, u, v, c.partials.p);
    hipLaunchKernelGGL(k_pcg_dot_final, dim3(1), dim3(kB), 0, s, c.partials.p, g, c.scalars.p + SC_COUNT - 1);
    double h = 0.0;
    check_hip(hipMemcpyAsync(&h, c.scalars.p + SC_COUNT - 1, sizeof(double), hipMemcpyDeviceToHost, s), "D2H");
    check_hip(hipStreamSynchronize(s), "sync");
    return h;
  };
  auto axpby = [&](double a, const double* u, double bb, double* v) {
    hipLaunchKernelGGL(k_pcg_axpby, dim3(grid_n(NP)), dim3(kB), 0, s, (int64_t)NP, a, u, bb, v);
  };
  // x0 = 0: q1 = b - A x0 = b (ConjugateGradientSolver.h:113-117)
  precond(0, b, r);
  precond(1, r, p);
  double gamma = dot(r, r);
  *gamma0 = gamma;
  const double threshold = std::max(epsilon_abs, epsilon_rel * epsilon_rel * gamma);
  int k = 1;
  for (; k <= max_iterations && (gamma > threshold || k <= min_iterations); k++) {
    if (!std::isfinite(gamma)) break;
    multiply(p, q1);                                 // q1 = A p
    const double alpha = gamma / dot(p, q1);         // alpha = gamma / (p' A p)
    axpby(alpha, p, 1.0, x);                         // x += alpha p
    precond(0, q1, q2);                              // q2 = L^-1 q1
    axpby(-alpha, q2, 1.0, r);                       // r -= alpha q2
    const double prev = gamma;
    gamma = dot(r, r);
    const double beta = gamma / prev;
    precond(1, r, q1);                               // q1 = L^-T r
    axpby(1.0, q1, beta, p);                         // p = q1 + beta p
  }
  *gamma_end = gamma;
  check_hip(hipGetLastError(), "pcg");
  return k - 1;
}

}  // namespace gt
