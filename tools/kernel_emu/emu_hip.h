// tools/kernel_emu/emu_hip.h -- just enough of the HIP / amdgcn device vocabulary to run ONE workgroup of a kernel on host threads: one
// std::thread per work-item, __syncthreads = a barrier over the workgroup, the wave collectives (__shfl_up, __ballot, the f64 16x16x4
// MFMA) = a rendezvous of the 64 threads of a wavefront through an exchange buffer.  Development tool (CPU tests of kernel LOGIC --
// indexing, barriers, who owns what -- before the kernel sees hardware); nothing of the product includes it.  Compile with clang++
// (ext_vector_type), -std=c++20 -pthread.
//
// A kernel is emulated faithfully as far as it keeps to what the hardware requires anyway: wave-uniform control flow around the
// collectives, every thread of the workgroup at every barrier.  A violation shows up as a hang (the tests run it under a timeout).
#pragma once
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

namespace emu {
struct Idx { unsigned x = 0, y = 0, z = 0; };
struct Wave {
  std::barrier<> bar{64};
  double da[64], db[64];
  long long li[64];
};
struct Group {
  explicit Group(int nthreads) : bar(nthreads), waves((nthreads + 63) / 64) {}
  std::barrier<> bar;
  std::vector<Wave> waves;
};
inline thread_local Idx t_idx, b_idx;
inline thread_local Group* t_group = nullptr;
inline Wave& wave() { return t_group->waves[t_idx.x >> 6]; }
inline int lane() { return (int)(t_idx.x & 63); }

// run `body` as one workgroup of `nthreads` work-items (a multiple of 64) with blockIdx.x = block
inline void run_workgroup(int nthreads, unsigned block, const std::function<void()>& body) {
  Group g(nthreads);
  std::vector<std::thread> th;
  th.reserve(nthreads);
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&, t] { t_idx.x = (unsigned)t; b_idx.x = block; t_group = &g; body(); g.bar.arrive_and_drop(); });
  for (auto& x : th) x.join();
}
}  // namespace emu

#define threadIdx (emu::t_idx)
#define blockIdx (emu::b_idx)
#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct double2 { double x, y; };

inline void __syncthreads() { emu::t_group->bar.arrive_and_wait(); }
inline int __shfl_up(int v, int delta, int /*width*/) {
  emu::Wave& w = emu::wave(); const int l = emu::lane();
  w.li[l] = v; w.bar.arrive_and_wait();
  const int r = l >= delta ? (int)w.li[l - delta] : v;
  w.bar.arrive_and_wait();
  return r;
}
inline unsigned long long __ballot(bool pred) {
  emu::Wave& w = emu::wave(); const int l = emu::lane();
  w.li[l] = pred ? 1 : 0; w.bar.arrive_and_wait();
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++) if (w.li[i]) m |= 1ull << i;
  w.bar.arrive_and_wait();
  return m;
}
inline int __popcll(unsigned long long m) { return __builtin_popcountll(m); }

// v_mfma_f64_16x16x4f64: lane l supplies A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15]; accumulator register r of lane l
// is C[row = (l >> 4) + 4 r][col = l & 15].  The sum over k as a chain of fused multiply-adds in k order (the hardware's internal order
// is not documented; the tests that use this compare against a reference formed the same way, or with a tolerance).
typedef double emu_v4d __attribute__((ext_vector_type(4)));
inline emu_v4d emu_mfma_f64_16x16x4(double a, double b, emu_v4d c) {
  emu::Wave& w = emu::wave(); const int l = emu::lane();
  w.da[l] = a; w.db[l] = b; w.bar.arrive_and_wait();
  emu_v4d d = c;
  for (int r = 0; r < 4; r++) {
    const int row = (l >> 4) + 4 * r, col = l & 15;
    double acc = c[r];
    for (int k = 0; k < 4; k++) acc = std::fma(w.da[16 * k + row], w.db[16 * k + col], acc);
    d[r] = acc;
  }
  w.bar.arrive_and_wait();
  return d;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) emu_mfma_f64_16x16x4((a), (b), (c))
#define __builtin_amdgcn_readfirstlane(v) (v)   // (only ever applied to wave-uniform values in the emulated kernels)
