// fused.h -- a GeneralSFM factor's whitened record recomputed where it is needed instead of read back from memory.
//
// The reference never materialises the Jacobians of its binary factors as such: BinaryJacobianFactor::updateHessian goes from the two
// blocks straight into the Hessian blocks (linear/BinaryJacobianFactor.h:51-83).  Rounds 1-4 wrote every factor's record
// [Jc 2x9 | Jp 2x3 | b 2] (208 B) once per linearisation and read it back in six kernels (1.2 GB per lambda try for 0.14 GB of
// records on the L1723 shape).  Since round 5 the record of a GeneralSFM factor only ever exists in the LDS image of the wavefront
// that needs it: a lane gathers its camera (17 doubles out of the L2-resident camera table), its point and its measurement -- 52 bytes
// of HBM traffic per factor instead of 208 -- and evaluates factors.h::sfm_linearize in place.  The record is a pure function of
// (values, measurement, noise row); k_lin_sfm's output is compared bit for bit with the stored-record build (tests/test_gpu_fused_linearization.py),
// and the step of the fused build with that build's.  (Between the six kernels that inline sfm_linearize the compiler is free to contract
// FMAs differently: they agree to rounding, not necessarily to the bit -- nothing relies on more.)
//
// Not for graphs with smart factors: the records of their measurements depend on the triangulation status of the factor and are
// overwritten by a second kernel for points at infinity (factors.hip) -- those graphs keep the stored records.
#pragma once
#include <hip/hip_runtime.h>

#include "factors.h"

namespace gt {

struct NoiseTab {
  const int32_t* kind;
  const int64_t* off;
  const double* data;
  const int32_t* rkind;
  const double* rk;
  __device__ __forceinline__ NoiseRef ref(int i) const { return NoiseRef{kind[i], data + off[i], rkind[i], rk[i]}; }
};

// what the recomputation reads: the GeneralSFM factor table, the packed values, the noise table
struct SfmTabs {
  const int32_t *cam_at, *pt_at, *nz;   // per factor: offsets of its camera and its point in the packed values, noise row
  const double* z;
  const double* values;
  NoiseTab nt;
};

// One entry of the CAMERA-SORTED contribution lists, packed once per graph (assemble.hip::k_cam_pack): what sfm_record reads through the
// factor index -- five 4 / 8-byte gathers out of five tables when the list is not in factor order (325 MB of 64-byte sectors per pass on
// the L1723 shape, whose factors are landmark-major) -- lies in list order here: two 16-byte loads per lane, coalesced.
struct alignas(16) CamPack { int32_t cam_at, pt_at, nz, pad; double z0, z1; };   // cam_at < 0: not a GeneralSFM entry

__device__ __forceinline__ void sfm_record_at(const SfmTabs& t, int cam_at, int pt_at, int nz, double z0, double z1, double* rec) {
  double c[17], p[3], zz[2] = {z0, z1};
  const double* cp = t.values + cam_at;
  const double* pp = t.values + pt_at;
#pragma unroll
  for (int k = 0; k < 17; k++) c[k] = cp[k];
#pragma unroll
  for (int k = 0; k < 3; k++) p[k] = pp[k];
  sfm_linearize(c, p, zz, t.nt.ref(nz), rec);
}
// record of GeneralSFM factor i -> rec[0 .. kSfmRec) (a row of the calling wavefront's LDS image)
__device__ __forceinline__ void sfm_record(const SfmTabs& t, int64_t i, double* rec) {
  sfm_record_at(t, t.cam_at[i], t.pt_at[i], t.nz[i], t.z[2 * i], t.z[2 * i + 1], rec);
}

}  // namespace gt

struct gtg_context;
namespace gt { SfmTabs sfm_tabs(gtg_context& c); }   // factors.hip: the tables of the handle's GeneralSFM factors at its CURRENT values
