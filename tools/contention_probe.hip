// tools/contention_probe.hip -- stand-alone probe for DESIGN.md section 8, item 2: how much does a short FP64-MFMA kernel on
// the serial chain (the thin update: 8.7 us alone, 25.8 us measured inside the factorisation) lose to a bulk MFMA kernel
// that fills the chip, and which of the cheap remedies gives it back?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/contention_probe.hip -o /tmp/contention_probe && /tmp/contention_probe
//
// Stand-ins with the launch shapes of the real kernels (512 threads = 8 wavefronts, each a chain of v_mfma_f64_16x16x4):
//   chain kernel: 120 workgroups x ~9 us     bulk kernel: 2048 workgroups, 64 KB of LDS each (2 per CU) x ~85 us
// Cases: chain alone | chain launched 20 us into the bulk kernel (today) | the same with the chain on a high-priority stream
// and s_setprio(3) | bulk limited to one workgroup per CU (96 KB of LDS) | bulk on a CU-masked stream leaving 32 CUs free.
// Reported per case: time from the chain kernel's launch to its completion (events on its stream), and the bulk kernel's time.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int PRIO>
__global__ __launch_bounds__(512) void k_mfma(double* out, int iters) {
  extern __shared__ char lds[];
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  v4f64 acc[8];
  for (int i = 0; i < 8; i++) acc[i] = (v4f64){0, 0, 0, 0};
  const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) lds[threadIdx.x] = 1;                    // keep the LDS allocation and the accumulators alive
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

static float timed(hipStream_t s, hipEvent_t a, hipEvent_t b) { float ms; CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b)); (void)s; return ms * 1e3f; }

int main() {
  double* out; CHECK(hipMalloc(&out, sizeof(double) * 4096 * 512));
  CHECK(hipFuncSetAttribute((const void*)k_mfma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  CHECK(hipFuncSetAttribute((const void*)k_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  int lo = 0, hi = 0; CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t sc, scp, sb, sbm;
  CHECK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CHECK(hipStreamCreateWithPriority(&scp, hipStreamNonBlocking, hi));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  std::vector<uint32_t> mask((ncu + 31) / 32, 0);
  for (int cu = 0; cu < ncu; cu++) if (cu % 8 != 7) mask[cu / 32] |= 1u << (cu % 32);      // one CU in eight stays free of the bulk kernel
  CHECK(hipExtStreamCreateWithCUMask(&sbm, (uint32_t)mask.size(), mask.data()));
  hipEvent_t c0, c1, b0, b1; CHECK(hipEventCreate(&c0)); CHECK(hipEventCreate(&c1)); CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1));
  // calibrate the iteration counts to the durations of the real kernels
  auto alone = [&](int grid, int smem, int iters) {
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
      CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(c0, sc));
      hipLaunchKernelGGL(k_mfma<0>, dim3(grid), dim3(512), smem, sc, out, iters);
      CHECK(hipEventRecord(c1, sc)); best = std::min(best, timed(sc, c0, c1));
    }
    return best;
  };
  int it_chain = 200, it_bulk = 400;
  for (int k = 0; k < 6; k++) { const float t = alone(120, 32768, it_chain); it_chain = std::max(8, (int)(it_chain * 9.0f / t)); }
  for (int k = 0; k < 6; k++) { const float t = alone(2048, 65536, it_bulk); it_bulk = std::max(8, (int)(it_bulk * 85.0f / t)); }
  std::printf("{\"compute_units\": %d, \"chain_iters\": %d, \"bulk_iters\": %d, \"chain_alone_us\": %.2f, \"bulk_alone_us\": %.2f", ncu, it_chain, it_bulk,
              alone(120, 32768, it_chain), alone(2048, 65536, it_bulk));
  struct Case { const char* name; hipStream_t chain; bool prio; hipStream_t bulk; int bulk_smem; };
  const Case cases[] = {{"today", sc, false, sb, 65536}, {"priority_stream_and_setprio", scp, true, sb, 65536},
                        {"bulk_one_workgroup_per_cu", sc, false, sb, 98304}, {"bulk_cu_masked_7_of_8", sc, false, sbm, 65536},
                        {"priority_and_bulk_one_per_cu", scp, true, sb, 98304}};
  for (const Case& cs : cases) {
    float best_chain = 1e30f, bulk_us = 0;
    for (int r = 0; r < 7; r++) {
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(b0, cs.bulk));
      hipLaunchKernelGGL(k_mfma<0>, dim3(2048), dim3(512), cs.bulk_smem, cs.bulk, out, it_bulk);
      CHECK(hipEventRecord(b1, cs.bulk));
      // let the bulk kernel get going (~20 us of host time), then the chain kernel
      { const auto w0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count() < 20.0) {} }
      CHECK(hipEventRecord(c0, cs.chain));
      if (cs.prio) hipLaunchKernelGGL(k_mfma<1>, dim3(120), dim3(512), 32768, cs.chain, out + 2048 * 512, it_chain);
      else hipLaunchKernelGGL(k_mfma<0>, dim3(120), dim3(512), 32768, cs.chain, out + 2048 * 512, it_chain);
      CHECK(hipEventRecord(c1, cs.chain));
      const float tc = timed(cs.chain, c0, c1); const float tb = timed(cs.bulk, b0, b1);
      if (tc < best_chain) { best_chain = tc; bulk_us = tb; }
    }
    std::printf(", \"%s\": {\"chain_us\": %.2f, \"bulk_us\": %.2f}", cs.name, best_chain, bulk_us);
  }
  std::printf("}\n");
  return 0;
}
