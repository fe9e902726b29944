// tools/potrf_micro_probe.hip -- probe for a register-resident micro-panel pivot chain of the 32x32 diagonal block (round 6).
//
// The shipped chain wavefront (chol_device.h::PotrfStep) publishes every column through LDS and costs 250-300 cycles per pivot (bare
// readlane -> rcp -> fma chain: 90); its followers replay every pivot from LDS and finish 1.2-1.7 us after it.  Here ONE wavefront keeps
// 64 ROWS (lane = row: lanes 0-31 the rows of the diagonal block D, lanes 32-63 the rows of the identity, which turn into L^-T) with all
// 32 columns in registers.  Pivots run inside micro-panels of W columns with v_readlane broadcasts only (no LDS on the chain); after a
// micro-panel its rank-W update is applied to the remaining columns, the operands again by v_readlane.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/potrf_micro_probe.hip -o tools/potrf_micro_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
constexpr int SB = 32;
typedef __attribute__((address_space(3))) volatile double* lds_vdouble_p;
__device__ __forceinline__ double readlane_f64(double v, int l) {
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], l); u.i[1] = __builtin_amdgcn_readlane(u.i[1], l); return u.d;
}
__device__ __forceinline__ double rcp_nr(double p) {
  double x = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, x, 1.0); x = __builtin_fma(x, e, x);
  e = __builtin_fma(-p, x, 1.0); x = __builtin_fma(x, e, x);
  return x;
}
__device__ __forceinline__ double rsqrt_nr(double p) {
  double r = __builtin_amdgcn_rsq(p);
  const double h = 0.5 * p;
  r = r * __builtin_fma(-h, r * r, 1.5);
  r = r * __builtin_fma(-h, r * r, 1.5);
  return r;
}
#define PIN(x) asm volatile("" : "+v"(x))

// MODE bits: 1 = sched_barrier after every pivot step, 2 = update loop column-major (c outer) instead of k outer
template <int W, int MODE>
__device__ __forceinline__ void chain_panel(double (&x)[SB], lds_vdouble_p rinvs) {
#pragma unroll
  for (int m = 0; m < SB / W; m++) {
    double v[W];
#pragma unroll
    for (int j = 0; j < W; j++) {
      const int J = W * m + j;
      const double piv = readlane_f64(x[J], J);
      const double rinv = rcp_nr(piv);
      if (j + 1 < W) {
        const double s1 = readlane_f64(x[J], J + 1);
        x[J + 1] = __builtin_fma(-(x[J] * s1), rinv, x[J + 1]);
      }
      const double u = x[J] * rinv;
      v[j] = u;
#pragma unroll
      for (int c = j + 2; c < W; c++) x[W * m + c] = __builtin_fma(-u, readlane_f64(x[J], W * m + c), x[W * m + c]);
      rinvs[J] = rinv;
      if constexpr ((MODE & 1) != 0) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr ((MODE & 2) == 0) {
#pragma unroll
      for (int k = 0; k < W; k++)
#pragma unroll
        for (int c = W * (m + 1); c < SB; c++) x[c] = __builtin_fma(-v[k], readlane_f64(x[W * m + k], c), x[c]);
    } else {
#pragma unroll
      for (int c = W * (m + 1); c < SB; c++)
#pragma unroll
        for (int k = 0; k < W; k++) x[c] = __builtin_fma(-v[k], readlane_f64(x[W * m + k], c), x[c]);
    }
    if constexpr ((MODE & 1) != 0) __builtin_amdgcn_sched_barrier(0);
  }
}

template <int W, int MODE>
__global__ __launch_bounds__(64) void probe(const double* A, double* out, long long* cyc) {
  __shared__ double rinvs[SB];
  __shared__ double rs[SB];
  const int lane = threadIdx.x, i = lane & 31;
  double x[SB];
#pragma unroll
  for (int c = 0; c < SB; c++) x[c] = lane < 32 ? A[i * SB + c] : (c == i ? 1.0 : 0.0);
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  chain_panel<W, MODE>(x, (lds_vdouble_p)rinvs);
  __builtin_amdgcn_s_waitcnt(0);
  const long long t1 = __builtin_amdgcn_s_memtime();
  // scale: r_c = sqrt(1 / piv_c)
  const double rv = rinvs[i];
  rs[i] = rv * rsqrt_nr(rv);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < SB; c++) x[c] *= rs[c];
  __builtin_amdgcn_s_waitcnt(0);
  const long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
#pragma unroll
  for (int c = 0; c < SB; c++) out[lane * SB + c] = x[c];
}

static std::vector<double> g_L, g_X;
template <int W, int MODE> void run(const double* dA, double* dO, long long* dC, const char* what) {
  long long c[2] = {0, 0};
  for (int r = 0; r < 3; r++) { hipLaunchKernelGGL((probe<W, MODE>), dim3(1), dim3(64), 0, 0, dA, dO, dC); hipDeviceSynchronize(); }
  hipMemcpy(c, dC, 16, hipMemcpyDeviceToHost);
  std::vector<double> o(64 * SB);
  hipMemcpy(o.data(), dO, o.size() * 8, hipMemcpyDeviceToHost);
  double eL = 0, eX = 0;
  for (int i = 0; i < SB; i++)
    for (int j = 0; j < SB; j++) {
      if (j <= i) eL = std::fmax(eL, std::fabs(o[i * SB + j] - g_L[i * SB + j]));
      eX = std::fmax(eX, std::fabs(o[(32 + i) * SB + j] - g_X[i * SB + j]));
    }
  printf("W %2d MODE %d  %-40s %6lld cycles = %5.1f per pivot, scale %4lld   |dL| %.2e |dLinvT| %.2e\n", W, MODE, what, c[0], c[0] / 32.0, c[1], eL, eX);
}
int main() {
  std::vector<double> A(SB * SB);
  for (int i = 0; i < SB; i++) for (int j = 0; j < SB; j++) A[i * SB + j] = (i == j ? 40.0 : 0.0) + 1.0 / (1 + i + j);
  // host reference: L (lower) and X = L^-T
  g_L.assign(SB * SB, 0.0); g_X.assign(SB * SB, 0.0);
  for (int j = 0; j < SB; j++) {
    double d = A[j * SB + j];
    for (int k = 0; k < j; k++) d -= g_L[j * SB + k] * g_L[j * SB + k];
    g_L[j * SB + j] = std::sqrt(d);
    for (int i = j + 1; i < SB; i++) {
      double s = A[i * SB + j];
      for (int k = 0; k < j; k++) s -= g_L[i * SB + k] * g_L[j * SB + k];
      g_L[i * SB + j] = s / g_L[j * SB + j];
    }
  }
  // X = L^-T: solve x L^T = e_i  (row i of X), x[c] = (e_i[c] - sum_{k<c} x[k] L[c][k]) / L[c][c]
  for (int i = 0; i < SB; i++)
    for (int c = 0; c < SB; c++) {
      double s = (c == i) ? 1.0 : 0.0;
      for (int k = 0; k < c; k++) s -= g_X[i * SB + k] * g_L[c * SB + k];
      g_X[i * SB + c] = s / g_L[c * SB + c];
    }
  double *dA, *dO; long long* dC;
  hipMalloc(&dA, SB * SB * 8); hipMalloc(&dO, 64 * SB * 8); hipMalloc(&dC, 16);
  hipMemcpy(dA, A.data(), SB * SB * 8, hipMemcpyHostToDevice);
  run<8, 0>(dA, dO, dC, "W=8, k-outer update");
  run<8, 1>(dA, dO, dC, "W=8, sched barriers");
  run<8, 2>(dA, dO, dC, "W=8, c-outer update");
  run<8, 3>(dA, dO, dC, "W=8, c-outer + sched barriers");
  run<4, 0>(dA, dO, dC, "W=4, k-outer update");
  run<4, 1>(dA, dO, dC, "W=4, sched barriers");
  run<16, 0>(dA, dO, dC, "W=16, k-outer update");
  run<16, 1>(dA, dO, dC, "W=16, sched barriers");
  run<32, 0>(dA, dO, dC, "W=32 (plain right-looking, readlane)");
  return 0;
}
