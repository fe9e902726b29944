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

// ---- the rest of the vocabulary of csrc/chol_device.h (the diagonal-tile body) ---------------------------------------------------------
#define GT_KERNEL_EMU 1
#define GT_PIN(x) ((void)0)
#define GT_DRAIN_STORES() ((void)0)
#define GT_LDS_VOLATILE(T) volatile T*
#define GT_WAVE_SYNC() (__atomic_thread_fence(__ATOMIC_SEQ_CST), emu::wave().bar.arrive_and_wait())
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
template <class T, class V> inline void emu_atomic_store(T* p, V v) { T x = (T)v; __atomic_store(p, &x, __ATOMIC_SEQ_CST); }
template <class T> inline T emu_atomic_load(const T* p) { T x; __atomic_load(const_cast<T*>(p), &x, __ATOMIC_SEQ_CST); return x; }
template <class T, class V> inline T emu_atomic_exchange(T* p, V v) { T x = (T)v, old; __atomic_exchange(p, &x, &old, __ATOMIC_SEQ_CST); return old; }
#define __hip_atomic_store(p, v, order, scope) emu_atomic_store((p), (v))
#define __hip_atomic_load(p, order, scope) emu_atomic_load((p))
#define __hip_atomic_exchange(p, v, order, scope) emu_atomic_exchange((p), (v))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) std::this_thread::yield()
// a fence at "wavefront" scope orders LDS traffic between the lanes of ONE wavefront: in order on the hardware, a rendezvous here
inline void emu_fence(const char* scope) {
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  if (scope[0] == 'w' && scope[1] == 'a') emu::wave().bar.arrive_and_wait();
}
#define __builtin_amdgcn_fence(order, scope) emu_fence(scope)
#define __builtin_amdgcn_s_memtime() 0ll
#define __builtin_amdgcn_ballot_w64(p) __ballot(p)
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt(x))
inline int __double2loint(double v) { long long b; std::memcpy(&b, &v, 8); return (int)(b & 0xffffffffll); }
inline int __double2hiint(double v) { long long b; std::memcpy(&b, &v, 8); return (int)(b >> 32); }
inline double __hiloint2double(int hi, int lo) { long long b = ((long long)hi << 32) | (unsigned)lo; double v; std::memcpy(&v, &b, 8); return v; }
inline long long __double_as_longlong(double v) { long long b; std::memcpy(&b, &v, 8); return b; }
inline double __longlong_as_double(long long b) { double v; std::memcpy(&v, &b, 8); return v; }
// v_readlane_b32: the value of lane `l` (wave-uniform) to every lane
inline int emu_readlane(int v, int l) {
  emu::Wave& w = emu::wave(); const int me = emu::lane();
  w.li[me] = v; w.bar.arrive_and_wait();
  const int r = (int)w.li[l & 63];
  w.bar.arrive_and_wait();
  return r;
}
#define __builtin_amdgcn_readlane(v, l) emu_readlane((v), (l))
// v_permlane32_swap(a, b): lanes 32-63 of the first operand <-> lanes 0-31 of the second; returns {new first, new second}
struct emu_u2 { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
inline emu_u2 emu_permlane32_swap(unsigned a, unsigned b) {
  emu::Wave& w = emu::wave(); const int me = emu::lane();
  w.li[me] = ((long long)a << 32) | b; w.bar.arrive_and_wait();
  emu_u2 r;
  if (me < 32) { r.v[0] = a; r.v[1] = (unsigned)(w.li[me + 32] >> 32); }                 // second's lower half <- first's upper half
  else { r.v[0] = (unsigned)(w.li[me - 32] & 0xffffffffll); r.v[1] = b; }                // first's upper half <- second's lower half
  w.bar.arrive_and_wait();
  return r;
}
#define __builtin_amdgcn_permlane32_swap(a, b, x, y) emu_permlane32_swap((a), (b))
// ds_bpermute_b32: lane reads the value of lane addr / 4
inline int emu_ds_bpermute(int addr, int v) {
  emu::Wave& w = emu::wave(); const int me = emu::lane();
  w.li[me] = v; w.bar.arrive_and_wait();
  const int r = (int)w.li[(addr >> 2) & 63];
  w.bar.arrive_and_wait();
  return r;
}
#define __builtin_amdgcn_ds_bpermute(addr, v) emu_ds_bpermute((addr), (v))

// ---- the rest of the vocabulary of csrc/chol_dataflow.hip (the two persistent kernels of the dataflow Cholesky) ------------------------
#include <chrono>
#define GT_XCC_ID(x) ((x) = 0)
#define GT_HW_ID(x) ((x) = 0)
typedef const void* gptr_t;
typedef void* lptr_t;
// global_load_lds_dwordx4: lane l moves 16 bytes from ITS global address to (wave-uniform LDS address) + 16 l
inline void emu_global_load_lds(gptr_t g, lptr_t l, int bytes) { std::memcpy(static_cast<char*>(l) + (size_t)emu::lane() * bytes, g, (size_t)bytes); }
#define __builtin_amdgcn_global_load_lds(g, l, bytes, off, aux) emu_global_load_lds((g), (l), (bytes))
// the 100 MHz constant clock, slowed down 100 000 times: the kernels' 20 ms wait bounds must not fire at emulation speed
inline long long wall_clock64() { return (long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 1000000); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomicCAS(int* p, int expect, int v) { __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return expect; }
template <class T, class V> inline T emu_atomic_fetch_max(T* p, V v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (old < (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
#define __hip_atomic_fetch_max(p, v, order, scope) emu_atomic_fetch_max((p), (v))
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_SEQ_CST)
// "am I the first lane": every thread says yes -- the single-lane read-modify-write polls are idempotent, and the v_readfirstlane that
// follows them is the identity here
#define __builtin_amdgcn_mbcnt_lo(a, b) 0u
#define __builtin_amdgcn_mbcnt_hi(a, b) 0u
