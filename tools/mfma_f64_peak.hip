// tools/mfma_f64_peak.hip -- micro-ceiling of v_mfma_f64_16x16x4_f64 on the chip ("measure a
// v_mfma_f64_16x16x4_f64 micro-ceiling first").  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, long long* cyc) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4f64){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
void run(int blocks_per_cu, const char* name) {
  const int iters = 20000, blocks = 256 * blocks_per_cu;
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * 256); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 100, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  const double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4.0 * blocks;
  std::printf("%s: %d blocks/CU, %d acc: %.2f TFLOP/s, %.1f s_memtime ticks per MFMA per wave (%.3f ms)\n", name, blocks_per_cu, NACC,
              flops / (ms * 1e-3) / 1e12, (double)hc / ((double)NACC * iters), ms);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<4>(1, "f64 16x16x4"); run<8>(1, "f64 16x16x4"); run<16>(1, "f64 16x16x4"); run<8>(2, "f64 16x16x4"); run<4>(4, "f64 16x16x4");
  return 0;
}
