// tools/mfma_f64_occupancy.hip -- how many wavefronts per SIMD and how many independent accumulation chains per wavefront does
// v_mfma_f64_16x16x4_f64 need to fill the matrix pipe?  One workgroup per CU (256 workgroups), W wavefronts each, every wavefront runs
// CH independent chains of dependent MFMAs.  Prints cycles per MFMA per SIMD (s_memtime) and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(1024) void k(double* out, int iters, long long* cyc) {
  v4f64 acc[CH];
  for (int i = 0; i < CH; i++) acc[i] = (v4f64){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CH; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < CH; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int CH>
void run(int waves) {
  const int iters = 2000, blocks = 256;
  double* out; long long* cyc; hipMalloc(&out, sizeof(double) * blocks * 1024); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(64 * waves), 0, 0, out, 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mfma_per_wave = (double)CH * iters, per_simd = mfma_per_wave * waves / 4.0;
  std::printf("waves/CU %2d (%.2f per SIMD)  chains %d: %7.1f cycles per MFMA per wavefront, %6.1f per SIMD, %6.2f TFLOP/s\n", waves, waves / 4.0, CH,
              c / mfma_per_wave, c / (per_simd < mfma_per_wave ? mfma_per_wave : per_simd), 2048.0 * mfma_per_wave * waves * blocks / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {4, 8, 12, 16}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
  return 0;
}
