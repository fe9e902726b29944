// Probe: what bounds one pivot step of the chain wavefront in k_potrf128?  Runs the 32-step elimination of a 32x32 SPD
// block with parts ablated and prints cycles per pivot.  hipcc --offload-arch=gfx950 -O3 -o tools/potrf_chain_probe.bin
// ABL bits: 1 no rest-update FMAs, 2 no LDS line write/read, 4 one Newton step, 8 no permlane (own = a[cJ]),
//           16 no s1 readlane, 32 no rinvs/prog stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int SB = 32;
typedef __attribute__((address_space(3))) volatile double* lds_vdouble_p;
typedef __attribute__((address_space(3))) volatile int* lds_vint_p;
__device__ __forceinline__ double readlane_f64(double v, int l) {
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], l); u.i[1] = __builtin_amdgcn_readlane(u.i[1], l); return u.d;
}
template <int H> __device__ __forceinline__ double half_bcast(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[H], (int)a[H]);
}
template <int ABL> __device__ __forceinline__ double rcp_nr(double p) {
  double x = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, x, 1.0); x = __builtin_fma(x, e, x);
  if (!(ABL & 4)) { e = __builtin_fma(-p, x, 1.0); x = __builtin_fma(x, e, x); }
  return x;
}
constexpr int kLineTrash = SB * SB;
template <int J, int ABL> struct Step {
  static __device__ __forceinline__ void run(double (&a)[16], const double (&cj)[16], double* lines, lds_vdouble_p rinvs,
                                             lds_vint_p prog, int lane, int i, int h) {
    constexpr int hJ = J & 1, cJ = J >> 1;
    const double piv = readlane_f64(a[cJ], J + 32 * hJ);
    const double rinv = rcp_nr<ABL>(piv);
    const double own = (ABL & 8) ? a[cJ] : half_bcast<hJ>(a[cJ]);
    double cn[16];
    if constexpr (J + 1 < SB) {
      constexpr int hN = (J + 1) & 1, cN = (J + 1) >> 1;
      const double s1 = (ABL & 16) ? 0.5 : readlane_f64(a[cJ], J + 1 + 32 * hJ);
      a[cN] = __builtin_fma(-((h == hN) ? own * s1 : 0.0), rinv, a[cN]);
      double* line = lines + (J + 1) * SB;
      if (!(ABL & 2)) {
        *(lds_vdouble_p)((h == hN) ? line + 16 * (i & 1) + (i >> 1) : lines + kLineTrash + lane) = a[cN];
#pragma unroll
        for (int cl = cN; cl < 16; cl++) cn[cl] = line[16 * h + cl];
      } else {
#pragma unroll
        for (int cl = cN; cl < 16; cl++) cn[cl] = cj[cl] * 0.999;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const double u = own * rinv;
    if (!(ABL & 1)) {
      if constexpr ((J & 1) == 1 && J + 1 < SB) a[cJ + 1] = __builtin_fma(-((h == 1) ? u : 0.0), cj[cJ + 1], a[cJ + 1]);
      constexpr int c0 = (J & 1) ? cJ + 2 : cJ + 1;
#pragma unroll
      for (int cl = c0; cl < 16; cl++) a[cl] = __builtin_fma(-u, cj[cl], a[cl]);
    }
#pragma unroll
    for (int cl = cJ + 1; cl < 16; cl++) asm volatile("" : "+v"(a[cl]));
    if (!(ABL & 32)) { rinvs[J] = rinv; if constexpr ((J & 3) == 3) *prog = J + 1; }
    Step<J + 1, ABL>::run(a, cn, lines, rinvs, prog, lane, i, h);
  }
};
template <int ABL> struct Step<SB, ABL> {
  static __device__ __forceinline__ void run(double (&)[16], const double (&)[16], double*, lds_vdouble_p, lds_vint_p, int, int, int) {}
};
template <int ABL> __global__ __launch_bounds__(64) void probe(const double* A, double* out, long long* cyc) {
  __shared__ double lines[SB * SB + 64]; __shared__ double rinvs[SB]; __shared__ int prog[2];
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  double a[16], c0[16];
  for (int cl = 0; cl < 16; cl++) a[cl] = A[i * SB + 2 * cl + h];
  const int pos = 16 * (i & 1) + (i >> 1);
  *(lds_vdouble_p)((h == 0) ? lines + pos : lines + kLineTrash + lane) = a[0];
  for (int cl = 0; cl < 16; cl++) c0[cl] = lines[16 * h + cl];
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  Step<0, ABL>::run(a, c0, lines, (lds_vdouble_p)rinvs, (lds_vint_p)prog, lane, i, h);
  __builtin_amdgcn_s_waitcnt(0);
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) *cyc = t1 - t0;
  for (int cl = 0; cl < 16; cl++) out[i * SB + 2 * cl + h] = a[cl];
}
template <int ABL> void run(const double* dA, double* dO, long long* dC, const char* what) {
  long long c = 0;
  for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(probe<ABL>, dim3(1), dim3(64), 0, 0, dA, dO, dC); hipDeviceSynchronize(); }
  hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
  printf("ABL %2d  %-52s %6lld cycles = %5.1f per pivot\n", ABL, what, c, c / 32.0);
}
int main() {
  std::vector<double> A(SB * SB);
  for (int i = 0; i < SB; i++) for (int j = 0; j < SB; j++) A[i * SB + j] = (i == j ? 40.0 : 0.0) + 1.0 / (1 + i + j);
  double *dA, *dO; long long* dC;
  hipMalloc(&dA, SB * SB * 8); hipMalloc(&dO, SB * SB * 8); hipMalloc(&dC, 8);
  hipMemcpy(dA, A.data(), SB * SB * 8, hipMemcpyHostToDevice);
  run<0>(dA, dO, dC, "full step");
  run<1>(dA, dO, dC, "no rest-update FMAs");
  run<2>(dA, dO, dC, "no LDS line write/read");
  run<3>(dA, dO, dC, "no FMAs, no LDS");
  run<4>(dA, dO, dC, "one Newton step");
  run<8>(dA, dO, dC, "no permlane");
  run<16>(dA, dO, dC, "no s1 readlane");
  run<32>(dA, dO, dC, "no rinvs/prog stores");
  run<63>(dA, dO, dC, "pivot readlane + rcp + 1 NR + fma only");
  run<59>(dA, dO, dC, "same with 2 Newton steps");
  return 0;
}
