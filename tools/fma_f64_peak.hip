// tools/fma_f64_peak.hip -- micro-ceilings: v_fma_f64 (VALU) alone, and VALU waves co-resident with MFMA waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
// mode 0: all waves VALU; mode 1: all waves MFMA; mode 2: even waves MFMA, odd waves VALU
template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool mfma = MODE == 1 || (MODE == 2 && (wave & 1) == 0);
  double s = 0;
  if (mfma) {
    v4f64 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (v4f64){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double acc[32];
    for (int i = 0; i < 32; i++) acc[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-7;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 32; i++) acc[i] = __builtin_fma(acc[i], a, b);
    }
    for (int i = 0; i < 32; i++) s += acc[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name) {
  const int iters = 4000, blocks = 256 * 2;   // 8 waves per block, 2 blocks per CU = 4 waves per SIMD
  double* out; hipMalloc(&out, sizeof(double) * blocks * 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = 8.0 * blocks;
  const double f_mfma = 2.0 * 16 * 16 * 4 * 8 * iters, f_valu = 2.0 * 64 * 32 * 8 * iters;
  double flops = MODE == 0 ? waves * f_valu : MODE == 1 ? waves * f_mfma : 0.5 * waves * (f_valu + f_mfma);
  std::printf("%s: %.2f TFLOP/s total (%.3f ms)", name, flops / (ms * 1e-3) / 1e12, ms);
  if (MODE == 2) std::printf("  [mfma part %.2f, valu part %.2f TFLOP/s]", 0.5 * waves * f_mfma / (ms * 1e-3) / 1e12, 0.5 * waves * f_valu / (ms * 1e-3) / 1e12);
  std::printf("\n");
  hipFree(out);
}
int main() { run<0>("VALU v_fma_f64 only"); run<1>("MFMA f64 16x16x4 only"); run<2>("half MFMA waves + half VALU waves"); return 0; }
