// recio.h -- wave-cooperative record I/O.
//
// The per-factor kernels are one-factor-per-lane: a lane produces / consumes a small record (26 doubles for an SFM
// factor's whitened [A1 | A2 | b]).  Written straight from the lanes, every store instruction of a wavefront touches
// 64 different cache lines (stride = record size): the kernels were bound by memory-instruction issue at a fraction
// of the HBM rate.  Here the 64 records of a wavefront, which are contiguous in global memory, travel through a
// per-wave LDS image: lanes access their own record there (odd pitch: at most 2-way bank conflicts), and the
// wavefront copies the whole 64 x REC block with 16-byte accesses, 1 KiB contiguous per instruction.
// Same-wave LDS traffic is in order; a wave-level fence/barrier orders the two phases for the compiler.
#pragma once
#include <hip/hip_runtime.h>

namespace gt {

template <int REC>
struct RecIO {
  static_assert(REC % 2 == 0, "records are copied in 16-byte pieces");
  static constexpr int PITCH = REC + 1;
  static constexpr int LDS_DOUBLES = 64 * PITCH;   // per wavefront

  static __device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // LDS image -> global: g points at the wavefront's first record (16-byte aligned), nrec <= 64 records are valid
  static __device__ __forceinline__ void store(const double* lds, double* __restrict__ g, int nrec, int lane) {
    wave_sync();
#pragma unroll
    for (int u = 0; u < REC / 2; u++) {
      const int e = 2 * (u * 64 + lane), r = e / REC, o = e - r * REC;
      if (r < nrec) {
        double2 v;
        v.x = lds[r * PITCH + o]; v.y = lds[r * PITCH + o + 1];
        *reinterpret_cast<double2*>(g + e) = v;
      }
    }
    wave_sync();   // the image may be overwritten afterwards
  }
  // global -> LDS image
  static __device__ __forceinline__ void load(double* lds, const double* __restrict__ g, int nrec, int lane) {
    double2 v[REC / 2];
#pragma unroll
    for (int u = 0; u < REC / 2; u++) {
      const int e = 2 * (u * 64 + lane), r = e / REC;
      v[u] = (r < nrec) ? *reinterpret_cast<const double2*>(g + e) : double2{0.0, 0.0};
    }
    wave_sync();   // earlier readers of the image are done
#pragma unroll
    for (int u = 0; u < REC / 2; u++) {
      const int e = 2 * (u * 64 + lane), r = e / REC, o = e - r * REC;
      lds[r * PITCH + o] = v[u].x; lds[r * PITCH + o + 1] = v[u].y;
    }
    wave_sync();
  }
};

}  // namespace gt
