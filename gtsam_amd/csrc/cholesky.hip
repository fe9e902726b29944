// cholesky.hip -- dense FP64 Cholesky of the reduced (camera / pose) system + triangular solves.
//
// This is the frontal-matrix kernel of the path: what gtsam::choleskyPartial (base/cholesky.cpp:107-158:
// Eigen LLT + TRSM + SYRK) does on the root clique(s) of the reduced camera system, and
// GaussianBayesTree::optimize (linear/linearAlgorithms-inst.h:49-155) does for the back-substitution.
//
// Layout: S row-major, lower triangle, ld = NP (multiple of 128), followed by one extra 128-row tile
// whose row 0 holds the right-hand side: carrying it through TRSM + trailing updates performs the
// forward solve L y = g for free (the reference does the same with its augmented [H g; g^T f] matrix,
// HessianFactor.cpp:239-252).
//
// Right-looking, tile = 128:
//   k_potrf_inv   one workgroup: factor the 128x128 diagonal tile in LDS and invert the factor
//   k_gemm_abt    FP64 MFMA (v_mfma_f64_16x16x4_f64) 128x128x128 tile products, used for both
//                 TRSM (X = A * Linv^T) and the SYRK/GEMM trailing update (C -= X_I X_J^T)
// MFMA is used only here (dense contraction); everything else on the path is HBM-bound.
#include "kernels.h"

namespace gt {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int T = kTile;        // 128
constexpr int LP = T + 1;       // LDS pitch of the diagonal tile (odd -> conflict-free column access)

// ---- diagonal tile: L = chol(A), Dinv = L^-1 ------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf_inv(double* __restrict__ S, int NP, int k, double* __restrict__ Dinv,
                                                   double* __restrict__ fail) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* A = reinterpret_cast<double*>(smem_raw);   // [T][LP]
  double* dinv = A + T * LP;                          // [T]
  const int tid = threadIdx.x;
  double* tile = S + ((int64_t)k * T) * NP + (int64_t)k * T;
  for (int e = tid; e < T * T; e += 256) {
    const int i = e / T, j = e % T;
    A[i * LP + j] = tile[(int64_t)i * NP + j];
  }
  __syncthreads();
  // right-looking unblocked Cholesky on the lower triangle (Eigen LLT semantics: pivot <= 0 fails)
  for (int j = 0; j < T; j++) {
    if (tid == 0) {
      const double p = A[j * LP + j];
      if (!(p > 0.0)) *fail = 1.0;
      const double d = sqrt(p);
      A[j * LP + j] = d;
      dinv[j] = 1.0 / d;
    }
    __syncthreads();
    const double dj = dinv[j];
    for (int i = j + 1 + tid; i < T; i += 256) A[i * LP + j] *= dj;
    __syncthreads();
    // trailing update: A[i][c] -= A[i][j] * A[c][j] for j < c <= i
    const int m = T - 1 - j;  // rows/cols remaining
    for (int e = tid; e < m * m; e += 256) {
      const int i = j + 1 + e / m, c = j + 1 + e % m;
      if (c <= i) A[i * LP + c] -= A[i * LP + j] * A[c * LP + j];
    }
    __syncthreads();
  }
  // inverse of the lower-triangular factor, one column per thread, stored transposed in the upper
  // triangle of A (X[c][i], i > c); diagonal in dinv.  Loops are wave-uniform (predicated) so that
  // A[i][j] is a broadcast read and A[c][j] (pitch 129) is conflict-free.
  if (tid < T) {
    const int c = tid;
    for (int i = 1; i < T; i++) {
      double acc = (i > c) ? A[i * LP + c] * dinv[c] : 0.0;
      for (int j = 1; j < i; j++) {
        const double l = A[i * LP + j];
        const double x = A[c * LP + j];
        if (j > c) acc += l * x;
      }
      if (i > c) A[c * LP + i] = -acc * dinv[i];
    }
  }
  __syncthreads();
  for (int e = tid; e < T * T; e += 256) {
    const int i = e / T, j = e % T;
    tile[(int64_t)i * NP + j] = (j <= i) ? A[i * LP + j] : 0.0;
    Dinv[e] = (j < i) ? A[j * LP + i] : (j == i ? dinv[i] : 0.0);
  }
}

// ---- 128x128x128 tile product on the FP64 matrix cores ----------------------------------------------
// C_tile (op)= Apanel[128 x 128] * Bpanel[128 x 128]^T, both panels row-major with the contraction
// index contiguous.  4 wavefronts in 2x2, each owns 64x64 = 4x4 MFMA tiles of 16x16.
// v_mfma_f64_16x16x4_f64 operand layout (cdna_hip_programming.md section 3): lane l supplies
// A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15]; result reg j holds C[row = (l>>4) + 4j][col = l&15].
constexpr int KC = 32;          // contraction chunk staged through LDS
constexpr int PP = KC + 2;      // LDS pitch (doubles): (2*PP) % 64 == 4 -> the 16 rows x 2 k of a half-wave hit 32 distinct banks

enum { MODE_TRSM = 0, MODE_SYRK = 1 };

template <int MODE>
__global__ __launch_bounds__(256) void k_gemm_abt(double* __restrict__ S, int NP, int k,
                                                  const double* __restrict__ Dinv) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* As = reinterpret_cast<double*>(smem_raw);
  double* Bs = As + T * PP;
  int I, J;
  if (MODE == MODE_TRSM) { I = k + 1 + blockIdx.x; J = k; }
  else {
    J = k + 1 + blockIdx.y; I = k + 1 + blockIdx.x;
    if (I < J) return;
  }
  const double* Ap = S + ((int64_t)I * T) * NP + (int64_t)k * T;                          // L(I,k) / A(I,k)
  const double* Bp = (MODE == MODE_TRSM) ? Dinv : S + ((int64_t)J * T) * NP + (int64_t)k * T;
  const int ldb = (MODE == MODE_TRSM) ? T : NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;

  v4f64 acc[4][4];
  for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) acc[a][b] = (v4f64){0.0, 0.0, 0.0, 0.0};

  for (int k0 = 0; k0 < T; k0 += KC) {
    __syncthreads();
    // stage a 128 x KC chunk of each panel: 16-byte pieces, 16 consecutive lanes cover one 256 B row
    for (int p = tid; p < T * (KC / 2); p += 256) {
      const int row = p / (KC / 2), pc = p % (KC / 2);
      const double2 va = *reinterpret_cast<const double2*>(Ap + (int64_t)row * NP + k0 + 2 * pc);
      const double2 vb = *reinterpret_cast<const double2*>(Bp + (int64_t)row * ldb + k0 + 2 * pc);
      *reinterpret_cast<double2*>(&As[row * PP + 2 * pc]) = va;
      *reinterpret_cast<double2*>(&Bs[row * PP + 2 * pc]) = vb;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KC; kk += 4) {
      double a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        a[t] = As[(wr * 64 + t * 16 + lr) * PP + kk + lk];
        b[t] = Bs[(wc * 64 + t * 16 + lr) * PP + kk + lk];
      }
#pragma unroll
      for (int ti = 0; ti < 4; ti++)
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
          acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
    }
  }
  __syncthreads();  // TRSM overwrites the A panel in place: every wave must be done reading it
  double* C = S + ((int64_t)I * T) * NP + (int64_t)J * T;
#pragma unroll
  for (int ti = 0; ti < 4; ti++)
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = wr * 64 + ti * 16 + lk + 4 * r;
        const int col = wc * 64 + tj * 16 + lr;
        double* cp = C + (int64_t)row * NP + col;
        if (MODE == MODE_TRSM) *cp = acc[ti][tj][r];
        else *cp -= acc[ti][tj][r];
      }
}

void launch_cholesky(gtg_context& c, double* S, int NP, int extra_rows, double* Dinv, double* fail) {
  const int nt = NP / T, ne = extra_rows / T;
  const size_t smem = sizeof(double) * (T * LP + T);
  const size_t gsmem = sizeof(double) * 2 * T * PP;
  static bool attr_set = false;
  if (!attr_set) {
    check_hip(hipFuncSetAttribute((const void*)k_potrf_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem), "smem attr");
    check_hip(hipFuncSetAttribute((const void*)k_gemm_abt<MODE_TRSM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gsmem), "smem attr");
    check_hip(hipFuncSetAttribute((const void*)k_gemm_abt<MODE_SYRK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gsmem), "smem attr");
    attr_set = true;
  }
  for (int k = 0; k < nt; k++) {
    double* Dk = Dinv + (size_t)k * T * T;
    hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(256), smem, c.stream, S, NP, k, Dk, fail);
    const int rows_below = (nt - 1 - k) + ne;
    if (rows_below > 0)
      hipLaunchKernelGGL(k_gemm_abt<MODE_TRSM>, dim3(rows_below), dim3(256), gsmem, c.stream, S, NP, k, Dk);
    const int cols = nt - 1 - k;
    if (cols > 0)
      hipLaunchKernelGGL(k_gemm_abt<MODE_SYRK>, dim3(rows_below, cols), dim3(256), gsmem, c.stream, S, NP, k, Dk);
  }
  check_hip(hipGetLastError(), "cholesky");
}

// ---- backward solve L^T x = y -------------------------------------------------------------------------
// Step k (descending): every workgroup recomputes x_k = Dinv_k^T y_k (128x128 GEMV, trivial) and then
// updates its 256-column slice of y:  y[j] -= sum_r L(k*128 + r, j) x_k[r]  for j < k*128.
__global__ __launch_bounds__(256) void k_bwd_step(const double* __restrict__ S, int NP, int k,
                                                  const double* __restrict__ Dinv, double* __restrict__ y,
                                                  double* __restrict__ x) {
  __shared__ double xs[T];
  __shared__ double ys[T];
  const int tid = threadIdx.x;
  if (tid < T) ys[tid] = y[k * T + tid];
  __syncthreads();
  if (tid < T) {
    double acc = 0.0;   // x_k[c] = sum_{i >= c} Dinv[i][c] * y_k[i]
    for (int i = tid; i < T; i++) acc += Dinv[i * T + tid] * ys[i];
    xs[tid] = acc;
    if (blockIdx.x == 0) x[k * T + tid] = acc;
  }
  __syncthreads();
  const int j = blockIdx.x * 256 + tid;
  if (j < k * T) {
    const double* Lr = S + ((int64_t)k * T) * NP + j;
    double acc = 0.0;
    for (int r = 0; r < T; r++) acc += Lr[(int64_t)r * NP] * xs[r];
    y[j] -= acc;
  }
}

void launch_backward_solve(gtg_context& c, double* S, int NP, double* x) {
  const int nt = NP / T;
  double* y = S + (int64_t)NP * NP;  // rhs row (extra tile, row 0) now holds y = L^-1 g
  for (int k = nt - 1; k >= 0; k--) {
    const int blocks = k == 0 ? 1 : (k * T + 255) / 256;
    hipLaunchKernelGGL(k_bwd_step, dim3(blocks), dim3(256), 0, c.stream, S, NP, k, c.Dinv.p + (size_t)k * T * T, y, x);
  }
  check_hip(hipGetLastError(), "backward_solve");
}

}  // namespace gt
