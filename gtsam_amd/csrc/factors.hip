// factors.hip -- per-factor kernels: linearize (K1/K2), nonlinear error (K9), linear error, retract (K10).
//
// One factor per lane (64 factors per wavefront): the factor->variable index arrays (int32) and the
// measurements are read coalesced; the variable blocks are gathered from the packed Values, which for
// the BAL configs is a 234 KB camera table (L2-resident) plus one 24-byte point per factor; each lane
// writes its whitened record [A1 | A2 | b] contiguously (208 B for an SFM factor).
// (Since round 5 the records of GeneralSFM factors are NOT stored when the graph has no smart factors: every kernel that needs one
// recomputes it in its wavefront's LDS image, fused.h; k_lin_sfm then only runs for gtg_get_jacobians.)
// Replaces NonlinearFactorGraph::linearize (nonlinear/NonlinearFactorGraph.cpp:239-278),
// NonlinearFactorGraph::error (:170-179), GaussianFactorGraph::error (linear/GaussianFactorGraph.cpp:71-78)
// and Values::retract (nonlinear/Values.cpp:52-63).
#include "factors.h"
#include "fused.h"
#include "kernels.h"
#include "recio.h"

namespace gt {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 2048;

static inline int grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// deterministic block reduction (fixed tree) of one double per thread; result valid in thread 0
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double sm[kBlock / 64];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < kBlock / 64; w++) r += sm[w];
  return r;
}

// ---- linearize ------------------------------------------------------------------------------------
// chunk c of 64 consecutive factors belongs to wavefront c of the grid (grid-stride over chunks)
__global__ __launch_bounds__(kBlock) void k_lin_sfm(int64_t n, const int32_t* __restrict__ cam,
    const int32_t* __restrict__ pt, const double* __restrict__ z, const int32_t* __restrict__ nz,
    const double* __restrict__ values, const int64_t* __restrict__ val_off, NoiseTab nt,
    double* __restrict__ J, const int32_t* __restrict__ smart_of, const int32_t* __restrict__ smart_status) {
  typedef RecIO<kSfmRec> IO;
  __shared__ double img[kBlock / 64][IO::LDS_DOUBLES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* my = img[wave];
  const int64_t nchunks = (n + 63) / 64, stride = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t ch = blockIdx.x * (int64_t)(kBlock / 64) + wave; ch < nchunks; ch += stride) {
    const int64_t i = ch * 64 + lane;
    if (i < n) {
      double c[17], p[3], zz[2];
      const double* cp = values + val_off[cam[i]];
      const double* pp = values + val_off[pt[i]];
      for (int k = 0; k < 17; k++) c[k] = cp[k];
      for (int k = 0; k < 3; k++) p[k] = pp[k];
      zz[0] = z[2 * i]; zz[1] = z[2 * i + 1];
      // the record is built in place in the wavefront's LDS image: as a local array it lived in scratch memory (the whitening
      // loops index it at run time) and every factor cost 208 B of scratch writes + reads on top of its 208 B record --
      // measured 512 B written per factor (rocprofv3 WRITE_SIZE), 2.46 x the algorithmic bytes
      sfm_linearize(c, p, zz, nt.ref(nz[i]), my + lane * IO::PITCH);
      // a measurement of a smart factor whose landmark did not triangulate contributes nothing (ZERO_ON_DEGENERACY,
      // SmartProjectionFactor.h:204-211)
      if (smart_of && smart_of[i] >= 0 && smart_status[smart_of[i]] != kTriValid)
        for (int k = 0; k < kSfmRec; k++) my[lane * IO::PITCH + k] = 0.0;
    }
    const int64_t left = n - ch * 64;
    IO::store(my, J + (int64_t)kSfmRec * ch * 64, left < 64 ? (int)left : 64, lane);
  }
}

__global__ __launch_bounds__(kBlock) void k_lin_proj(int64_t n, const int32_t* __restrict__ pose,
    const int32_t* __restrict__ pt, const double* __restrict__ z, const int32_t* __restrict__ nz,
    const int32_t* __restrict__ calib_idx, const int32_t* __restrict__ sensor_idx,
    const double* __restrict__ calib, const double* __restrict__ sensor,
    const double* __restrict__ values, const int64_t* __restrict__ val_off, NoiseTab nt,
    double* __restrict__ J) {
  typedef RecIO<kProjRec> IO;
  __shared__ double img[kBlock / 64][IO::LDS_DOUBLES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* my = img[wave];
  const int64_t nchunks = (n + 63) / 64, stride = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t ch = blockIdx.x * (int64_t)(kBlock / 64) + wave; ch < nchunks; ch += stride) {
    const int64_t i = ch * 64 + lane;
    if (i < n) {
      double T[12], p[3], zz[2], K[kCalibStride], S[12];
      const double* tp = values + val_off[pose[i]];
      const double* pp = values + val_off[pt[i]];
      for (int k = 0; k < 12; k++) T[k] = tp[k];
      for (int k = 0; k < 3; k++) p[k] = pp[k];
      for (int k = 0; k < kCalibStride; k++) K[k] = calib[kCalibStride * calib_idx[i] + k];
      const int si = sensor_idx[i];
      if (si >= 0) for (int k = 0; k < 12; k++) S[k] = sensor[12 * si + k];
      zz[0] = z[2 * i]; zz[1] = z[2 * i + 1];
      proj_linearize(T, K, si >= 0 ? S : nullptr, p, zz, nt.ref(nz[i]), my + lane * IO::PITCH);   // in place in the LDS image (see k_lin_sfm)
    }
    const int64_t left = n - ch * 64;
    IO::store(my, J + (int64_t)kProjRec * ch * 64, left < 64 ? (int)left : 64, lane);
  }
}

// BetweenFactor<Pose3> or BetweenFactor<Pose2>: decided by the type of the factor's variables (both of one type)
__global__ __launch_bounds__(kBlock) void k_lin_between(int64_t n, const int32_t* __restrict__ v1,
    const int32_t* __restrict__ v2, const double* __restrict__ z, const int32_t* __restrict__ nz,
    const int32_t* __restrict__ var_type, const double* __restrict__ values, const int64_t* __restrict__ val_off,
    NoiseTab nt, double* __restrict__ J) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double* a = values + val_off[v1[i]];
    const double* b = values + val_off[v2[i]];
    const int ni = nz[i];
    if (var_type[v1[i]] == 3) {
      double p1[3], p2[3], zz[3];
      for (int k = 0; k < 3; k++) { p1[k] = a[k]; p2[k] = b[k]; zz[k] = z[12 * i + k]; }
      between2_linearize(p1, p2, zz, nt.ref(ni), J + (int64_t)kBetweenRec * i);
    } else {
      double T1[12], T2[12], Z[12];
      for (int k = 0; k < 12; k++) { T1[k] = a[k]; T2[k] = b[k]; Z[k] = z[12 * i + k]; }
      between_linearize(T1, T2, Z, nt.ref(ni), J + (int64_t)kBetweenRec * i);
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_lin_prior(int64_t n, const int32_t* __restrict__ var,
    const int64_t* __restrict__ poff, const double* __restrict__ pdata, const int32_t* __restrict__ nz,
    const int32_t* __restrict__ var_type, const double* __restrict__ values,
    const int64_t* __restrict__ val_off, NoiseTab nt, double* __restrict__ J) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int v = var[i];
    const int ni = nz[i];
    prior_linearize(var_type[v], values + val_off[v], pdata + poff[i], nt.ref(ni), J + (int64_t)kPriorRec * i);
  }
}

// ---- nonlinear error --------------------------------------------------------------------------------
struct ErrArgs {
  int64_t n_sfm, n_proj, n_between, n_prior;
  const int32_t *sfm_cam, *sfm_pt, *sfm_nz; const double* sfm_z;
  const int32_t *proj_pose, *proj_pt, *proj_nz, *proj_calib, *proj_sensor; const double *proj_z, *calib, *sensor;
  const int32_t *bt_v1, *bt_v2, *bt_nz; const double* bt_z;
  const int32_t *pr_var, *pr_nz; const int64_t* pr_off; const double* pr_data;
  const int32_t* var_type; const int64_t* val_off;
  const int32_t *smart_of, *smart_status;   // smart factors: the factor of an sfm observation (or -1) and its triangulation status
};

// Block b handles a fixed slice, so the summation order is fixed.  TYPES: bit 0 -- the GeneralSFM factors, bit 1 -- the other three
// types.  (One kernel for all four types -- rounds 1-4 -- needed 256 registers + scratch for the union of their evaluators: one
// wavefront per SIMD, 32 us for the 0.68 M factors of the L1723 shape, a gather-and-evaluate kernel that lives on occupancy.)
template <int TYPES>
__global__ __launch_bounds__(kBlock) void k_error(ErrArgs a, const double* __restrict__ values, NoiseTab nt,
                                                  double* __restrict__ partials) {
  double acc = 0.0;
  const int64_t tid = blockIdx.x * (int64_t)kBlock + threadIdx.x, stride = (int64_t)gridDim.x * kBlock;
  if constexpr ((TYPES & 1) != 0)
  for (int64_t i = tid; i < a.n_sfm; i += stride) {
    double c[17], p[3], zz[2];
    const double* cp = values + a.val_off[a.sfm_cam[i]];
    const double* pp = values + a.val_off[a.sfm_pt[i]];
    for (int k = 0; k < 17; k++) c[k] = cp[k];
    for (int k = 0; k < 3; k++) p[k] = pp[k];
    zz[0] = a.sfm_z[2 * i]; zz[1] = a.sfm_z[2 * i + 1];
    const int ni = a.sfm_nz[i];
    // (a smart factor whose landmark did not triangulate has error 0, SmartProjectionFactor.h:407-427)
    if (!(a.smart_of && a.smart_of[i] >= 0 && a.smart_status[a.smart_of[i]] != kTriValid)) acc += sfm_error(c, p, zz, nt.ref(ni));
  }
  if constexpr ((TYPES & 2) != 0) {
  for (int64_t i = tid; i < a.n_proj; i += stride) {
    double T[12], p[3], zz[2], K[kCalibStride], S[12];
    const double* tp = values + a.val_off[a.proj_pose[i]];
    const double* pp = values + a.val_off[a.proj_pt[i]];
    for (int k = 0; k < 12; k++) T[k] = tp[k];
    for (int k = 0; k < 3; k++) p[k] = pp[k];
    for (int k = 0; k < kCalibStride; k++) K[k] = a.calib[kCalibStride * a.proj_calib[i] + k];
    const int si = a.proj_sensor[i];
    if (si >= 0) for (int k = 0; k < 12; k++) S[k] = a.sensor[12 * si + k];
    zz[0] = a.proj_z[2 * i]; zz[1] = a.proj_z[2 * i + 1];
    const int ni = a.proj_nz[i];
    acc += proj_error(T, K, si >= 0 ? S : nullptr, p, zz, nt.ref(ni));
  }
  for (int64_t i = tid; i < a.n_between; i += stride) {
    const double* x = values + a.val_off[a.bt_v1[i]];
    const double* y = values + a.val_off[a.bt_v2[i]];
    if (a.var_type[a.bt_v1[i]] == 3) {
      double p1[3], p2[3], zz[3];
      for (int k = 0; k < 3; k++) { p1[k] = x[k]; p2[k] = y[k]; zz[k] = a.bt_z[12 * i + k]; }
      acc += between2_error(p1, p2, zz, nt.ref(a.bt_nz[i]));
      continue;
    }
    double T1[12], T2[12], Z[12];
    for (int k = 0; k < 12; k++) { T1[k] = x[k]; T2[k] = y[k]; Z[k] = a.bt_z[12 * i + k]; }
    const int ni = a.bt_nz[i];
    acc += between_error(T1, T2, Z, nt.ref(ni));
  }
  for (int64_t i = tid; i < a.n_prior; i += stride) {
    const int v = a.pr_var[i];
    const int ni = a.pr_nz[i];
    acc += prior_error(a.var_type[v], values + a.val_off[v], a.pr_data + a.pr_off[i], nt.ref(ni));
  }
  }
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// sums `n` partials (n <= kMaxBlocks) in a fixed order into out[slot]; nslots interleaved partial arrays
__global__ __launch_bounds__(kBlock) void k_final_sum(const double* __restrict__ partials, int n, int nslots,
                                                      double* __restrict__ out, int slot0) {
  for (int s = 0; s < nslots; s++) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) acc += partials[(int64_t)s * kMaxBlocks + i];
    const double r = block_sum(acc);
    if (threadIdx.x == 0) out[slot0 + s] = r;
    __syncthreads();
  }
}

// ---- linear error: 0.5*||b||^2 and 0.5*||A delta - b||^2 on the undamped system --------------------
struct LinErrArgs {
  int64_t n_sfm, n_proj, n_between, n_prior;
  const int32_t *sfm_cam, *sfm_pt, *proj_pose, *proj_pt, *bt_v1, *bt_v2, *pr_var;
  const double *sfm_J, *proj_J, *bt_J, *pr_J;
  const int32_t* var_type; const int64_t* dim_off;
};

// FUSED: the records of the GeneralSFM factors are recomputed (fused.h) instead of read: chunk c of 64 consecutive factors belongs to
// wavefront c of the grid, as in k_lin_sfm
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void k_linear_error(LinErrArgs a, SfmTabs t, const double* __restrict__ delta,
                                                         double* __restrict__ partials) {
  typedef RecIO<kSfmRec> IO;
  __shared__ double img[FUSED ? kBlock / 64 : 1][FUSED ? IO::LDS_DOUBLES : 1];
  double e0 = 0.0, e1 = 0.0;
  const int64_t tid = blockIdx.x * (int64_t)kBlock + threadIdx.x, stride = (int64_t)gridDim.x * kBlock;
  if constexpr (FUSED) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* J = img[wave] + lane * IO::PITCH;
    const int64_t nchunks = (a.n_sfm + 63) / 64, cstride = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t ch = blockIdx.x * (int64_t)(kBlock / 64) + wave; ch < nchunks; ch += cstride) {
      const int64_t i = ch * 64 + lane;
      if (i >= a.n_sfm) continue;
      sfm_record(t, i, J);
      const double* dc = delta + a.dim_off[a.sfm_cam[i]];
      const double* dp = delta + a.dim_off[a.sfm_pt[i]];
      for (int r = 0; r < 2; r++) {
        const double b = J[24 + r];
        double v = -b;
        for (int k = 0; k < 9; k++) v += J[9 * r + k] * dc[k];
        for (int k = 0; k < 3; k++) v += J[18 + 3 * r + k] * dp[k];
        e0 += b * b; e1 += v * v;
      }
    }
  } else {
  for (int64_t i = tid; i < a.n_sfm; i += stride) {
    const double* J = a.sfm_J + (int64_t)kSfmRec * i;
    const double* dc = delta + a.dim_off[a.sfm_cam[i]];
    const double* dp = delta + a.dim_off[a.sfm_pt[i]];
    for (int r = 0; r < 2; r++) {
      const double b = J[24 + r];
      double v = -b;
      for (int k = 0; k < 9; k++) v += J[9 * r + k] * dc[k];
      for (int k = 0; k < 3; k++) v += J[18 + 3 * r + k] * dp[k];
      e0 += b * b; e1 += v * v;
    }
  }
  }
  for (int64_t i = tid; i < a.n_proj; i += stride) {
    const double* J = a.proj_J + (int64_t)kProjRec * i;
    const double* dc = delta + a.dim_off[a.proj_pose[i]];
    const double* dp = delta + a.dim_off[a.proj_pt[i]];
    for (int r = 0; r < 2; r++) {
      const double b = J[18 + r];
      double v = -b;
      for (int k = 0; k < 6; k++) v += J[6 * r + k] * dc[k];
      for (int k = 0; k < 3; k++) v += J[12 + 3 * r + k] * dp[k];
      e0 += b * b; e1 += v * v;
    }
  }
  for (int64_t i = tid; i < a.n_between; i += stride) {
    const double* J = a.bt_J + (int64_t)kBetweenRec * i;
    const double* d1 = delta + a.dim_off[a.bt_v1[i]];
    const double* d2 = delta + a.dim_off[a.bt_v2[i]];
    const int d = a.var_type[a.bt_v1[i]] == 3 ? 3 : 6;   // Pose2 : Pose3
    for (int r = 0; r < d; r++) {
      const double b = J[72 + r];
      double v = -b;
      for (int k = 0; k < d; k++) v += J[d * r + k] * d1[k] + J[36 + d * r + k] * d2[k];
      e0 += b * b; e1 += v * v;
    }
  }
  for (int64_t i = tid; i < a.n_prior; i += stride) {
    const double* J = a.pr_J + (int64_t)kPriorRec * i;
    const int v_ = a.pr_var[i];
    const int t = a.var_type[v_];
    const int d = t == 0 ? 6 : t == 1 ? 9 : 3;
    const double* dv = delta + a.dim_off[v_];
    for (int r = 0; r < d; r++) {
      const double b = J[81 + r];
      double v = -b;
      for (int k = 0; k < d; k++) v += J[d * r + k] * dv[k];
      e0 += b * b; e1 += v * v;
    }
  }
  const double s0 = block_sum(0.5 * e0);
  const double s1 = block_sum(0.5 * e1);
  if (threadIdx.x == 0) { partials[blockIdx.x] = s0; partials[kMaxBlocks + blockIdx.x] = s1; }
}

// ---- retract ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_retract(int32_t n_vars, const int32_t* __restrict__ var_type,
    const int64_t* __restrict__ val_off, const int64_t* __restrict__ dim_off,
    const double* __restrict__ values, const double* __restrict__ delta, double* __restrict__ trial) {
  for (int64_t v = blockIdx.x * (int64_t)kBlock + threadIdx.x; v < n_vars; v += (int64_t)gridDim.x * kBlock)
    value_retract(var_type[v], values + val_off[v], delta + dim_off[v], trial + val_off[v]);
}

__global__ __launch_bounds__(kBlock) void k_sumsq(int64_t n, const double* __restrict__ x, double* __restrict__ partials) {
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) acc += x[i] * x[i];
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// ---- launchers --------------------------------------------------------------------------------------
static NoiseTab noise_tab(gtg_context& c) { return NoiseTab{c.noise_kind.p, c.noise_off.p, c.noise_data.p, c.noise_rkind.p, c.noise_rk.p}; }
SfmTabs sfm_tabs(gtg_context& c) { return SfmTabs{c.f.sfm_cam_at.p, c.f.sfm_point_at.p, c.f.sfm_noise.p, c.f.sfm_z.p, c.values.p, noise_tab(c)}; }

// ---- smart factors: the landmark of every factor from the cameras in `values` ------------------------------------------
// One factor per lane.  What SmartProjectionFactor::triangulateSafe does (SmartProjectionFactor.h:127-183): re-triangulate only
// when a camera pose moved by more than retriangulationThreshold in some entry since the cached triangulation (Pose3::equals),
// else reuse the cached result -- the cache is per factor and shared by linearize() and error(), as in the reference, so the
// sequence of calls is mirrored by the host (gate: the reference does not evaluate the error of a trial step whose linear
// cost change is negative, LevenbergMarquardtOptimizer.cpp:180-191).  The point goes into the hidden variable's value slot.
// the code of the case the reference throws in: the LARGER one wins when lanes meet both (positive doubles order like their bit patterns)
__device__ __forceinline__ void raise_unsupported(double* scalars, double code) {
  atomicMax(reinterpret_cast<unsigned long long*>(scalars + SC_UNSUPPORTED), (unsigned long long)__double_as_longlong(code));
}
struct SmartArgs {
  int64_t n, obs0;
  const int64_t* ptr; const int32_t *sfm_cam, *sfm_point; const double* sfm_z; const int64_t* val_off; const double* params;
  int32_t *status, *cache_state; double *cache_pose, *cache_point;
};
__global__ __launch_bounds__(kBlock) void k_smart_triangulate(SmartArgs a, double* __restrict__ values, const double* __restrict__ gate,
                                                              double* __restrict__ scalars, int for_linearize) {
  // (a try whose factorisation timed out is repeated, api.hip: its garbage trial point must not touch the triangulation cache)
  if (gate && (gate[SC_TIMEOUT] != 0.0 || !(gate[SC_LIN0] - gate[SC_LIN1] >= 0))) return;
  for (int64_t sf = blockIdx.x * (int64_t)kBlock + threadIdx.x; sf < a.n; sf += (int64_t)gridDim.x * kBlock) {
    const int64_t k0 = a.ptr[sf], o0 = a.obs0 + k0;
    const int m = (int)(a.ptr[sf + 1] - k0);
    if (m == 0) continue;                                // a factor of another shard (every factor has a measurement: gtg_upload_problem)
    const double* prm = a.params + 8 * sf;
    const double thr = prm[3];
    int st;
    double pt[3] = {0.0, 0.0, 0.0};
    if (m < 2) {
      st = kTriDegenerate;
    } else {
      bool again = a.cache_state[sf] < 0;
      for (int k = 0; k < m && !again; k++) {
        const double* pose = values + a.val_off[a.sfm_cam[o0 + k]];
        const double* old = a.cache_pose + 12 * (k0 + k);
        for (int e = 0; e < 12; e++) again = again || fabs(pose[e] - old[e]) > thr;
      }
      if (again) {
        for (int k = 0; k < m; k++) {
          const double* pose = values + a.val_off[a.sfm_cam[o0 + k]];
          for (int e = 0; e < 12; e++) a.cache_pose[12 * (k0 + k) + e] = pose[e];
        }
        st = smart_triangulate(m, a.sfm_cam + o0, a.val_off, values, a.sfm_z + 2 * o0, prm[0], prm[1], prm[2], pt, prm[6] != 0.0);
        a.cache_state[sf] = st;
        for (int e = 0; e < 3; e++) a.cache_point[3 * sf + e] = pt[e];
      } else {
        st = a.cache_state[sf];
        for (int e = 0; e < 3; e++) pt[e] = a.cache_point[3 * sf + e];
      }
    }
    // A failed track (TriangulationResult without a point: every status but VALID) is, depending on the factor's parameters,
    //   linearize: HESSIAN mode -- nothing under ZERO_ON_DEGENERACY (SmartProjectionFactor.h:212-219), else a POINT AT INFINITY, the
    //              direction of the first measurement from the first camera (computeJacobiansWithTriangulatedPoint, :356-371);
    //              JACOBIAN_Q / JACOBIAN_SVD -- an empty factor whatever the degeneracy mode (:245-272);
    //   error():   the point at infinity under HANDLE_INFINITY, 0.0 otherwise (totalReprojectionError, :419-429).
    // The direction is taken from the cameras of THIS call (it is not part of the cached triangulation).
    const int mode = (int)prm[4], lin_mode = (int)prm[5];
    bool at_infinity = st != kTriValid && st != kTriNoConvergence && st != kTriCheiralityThrown && (for_linearize ? (mode != 1 && lin_mode == 0) : mode == 2);
    if (at_infinity && !sfm_backproject_at_infinity(values + a.val_off[a.sfm_cam[o0]], a.sfm_z + 2 * o0, pt)) { st = kTriNoConvergence; at_infinity = false; }
    a.status[sf] = st | (at_infinity ? kTriAtInfinity : 0);
    double* slot = values + a.val_off[a.sfm_point[o0]];
    for (int e = 0; e < 3; e++) slot[e] = (st == kTriValid || at_infinity) ? pt[e] : 0.0;
    if (st == kTriNoConvergence) raise_unsupported(scalars, 1.0);      // Cal3Bundler::calibrate throws in the reference
    if (st == kTriCheiralityThrown) raise_unsupported(scalars, kUnsupportedCheirality);   // enableEPI: geom.h::triangulate_refine
  }
}

// The measurements of the tracks flagged kTriAtInfinity, one per lane: k_lin_sfm has zeroed their records, this kernel writes the
// rotation-only records of SmartFactorBase::computeJacobians<Unit3> in their place (landmark block 2 x 2 + a zero column: the hidden
// landmark keeps its 3-wide slot, k_point_factor puts a 1 on the uncoupled third diagonal entry).  A direction behind one of the
// cameras is a CheiralityException that nothing in the reference catches: reported (SC_UNSUPPORTED = kUnsupportedCheirality).
__global__ __launch_bounds__(kBlock) void k_lin_smart_at_infinity(int64_t n_meas, int64_t obs0, const int32_t* __restrict__ cam,
    const int32_t* __restrict__ pt, const double* __restrict__ z, const int32_t* __restrict__ nz, const double* __restrict__ values,
    const int64_t* __restrict__ val_off, NoiseTab nt, double* __restrict__ J, const int32_t* __restrict__ smart_of,
    const int32_t* __restrict__ status, double* __restrict__ scalars) {
  for (int64_t k = blockIdx.x * (int64_t)kBlock + threadIdx.x; k < n_meas; k += (int64_t)gridDim.x * kBlock) {
    const int64_t o = obs0 + k;
    if (!(status[smart_of[o]] & kTriAtInfinity)) continue;
    double c[17], d[3], zz[2], rec[kSfmRec];
    const double* cp = values + val_off[cam[o]];
    const double* dp = values + val_off[pt[o]];
    for (int e = 0; e < 17; e++) c[e] = cp[e];
    for (int e = 0; e < 3; e++) d[e] = dp[e];
    zz[0] = z[2 * o]; zz[1] = z[2 * o + 1];
    if (!sfm_linearize_at_infinity(c, d, zz, nt.ref(nz[o]), rec)) raise_unsupported(scalars, kUnsupportedCheirality);
    for (int e = 0; e < kSfmRec; e++) J[(int64_t)kSfmRec * o + e] = rec[e];
  }
}
// ... and their share of the error (HANDLE_INFINITY): one workgroup, fixed lane assignment, added to the sum k_error left in `slot`.
__global__ __launch_bounds__(kBlock) void k_error_smart_at_infinity(int64_t n_meas, int64_t obs0, const int32_t* __restrict__ cam,
    const int32_t* __restrict__ pt, const double* __restrict__ z, const int32_t* __restrict__ nz, const double* __restrict__ values,
    const int64_t* __restrict__ val_off, NoiseTab nt, const int32_t* __restrict__ smart_of, const int32_t* __restrict__ status,
    const double* __restrict__ gate, double* __restrict__ scalars, int slot) {
  if (gate && (gate[SC_TIMEOUT] != 0.0 || !(gate[SC_LIN0] - gate[SC_LIN1] >= 0))) return;      // (the triangulation of this trial point was skipped: see k_smart_triangulate)
  double acc = 0.0;
  for (int64_t k = threadIdx.x; k < n_meas; k += kBlock) {
    const int64_t o = obs0 + k;
    if (!(status[smart_of[o]] & kTriAtInfinity)) continue;
    double c[17], d[3], zz[2], e;
    const double* cp = values + val_off[cam[o]];
    const double* dp = values + val_off[pt[o]];
    for (int i = 0; i < 17; i++) c[i] = cp[i];
    for (int i = 0; i < 3; i++) d[i] = dp[i];
    zz[0] = z[2 * o]; zz[1] = z[2 * o + 1];
    if (!sfm_error_at_infinity(c, d, zz, nt.ref(nz[o]), &e)) raise_unsupported(scalars, kUnsupportedCheirality);
    acc += e;
  }
  const double s = block_sum(acc);
  if (threadIdx.x == 0) scalars[slot] += s;
}

void launch_smart_triangulate(gtg_context& c, double* values, const double* gate, bool for_linearize) {
  if (!c.n_smart) return;
  SmartArgs a{c.n_smart, c.smart_obs0, c.smart_ptr.p, c.f.sfm_cam.p, c.f.sfm_point.p, c.f.sfm_z.p, c.val_off.p, c.smart_params.p,
              for_linearize ? c.smart_lin_status.p : c.smart_status.p, c.smart_cache_state.p, c.smart_cache_pose.p, c.smart_cache_point.p};
  hipLaunchKernelGGL(k_smart_triangulate, dim3(grid_for(c.n_smart)), dim3(kBlock), 0, c.stream, a, values, gate, c.scalars.p,
                     for_linearize ? 1 : 0);
  check_hip(hipGetLastError(), "smart_triangulate");
}

void launch_linearize(gtg_context& c) {
  auto& f = c.f;
  NoiseTab nt = noise_tab(c);
  if (f.n_sfm && !c.fused_sfm)     // (fused: nobody reads stored records of these factors, fused.h)
    hipLaunchKernelGGL(k_lin_sfm, dim3(grid_for(f.n_sfm)), dim3(kBlock), 0, c.stream, f.n_sfm, f.sfm_cam.p,
                       f.sfm_point.p, f.sfm_z.p, f.sfm_noise.p, c.values.p, c.val_off.p, nt, f.sfm_J.p,
                       c.n_smart ? c.sfm_smart.p : nullptr, c.smart_lin_status.p);
  if (c.n_smart)
    hipLaunchKernelGGL(k_lin_smart_at_infinity, dim3(grid_for(f.n_sfm - c.smart_obs0)), dim3(kBlock), 0, c.stream, f.n_sfm - c.smart_obs0, c.smart_obs0,
                       f.sfm_cam.p, f.sfm_point.p, f.sfm_z.p, f.sfm_noise.p, c.values.p, c.val_off.p, nt, f.sfm_J.p, c.sfm_smart.p,
                       c.smart_lin_status.p, c.scalars.p);
  if (f.n_proj)
    hipLaunchKernelGGL(k_lin_proj, dim3(grid_for(f.n_proj)), dim3(kBlock), 0, c.stream, f.n_proj, f.proj_pose.p,
                       f.proj_point.p, f.proj_z.p, f.proj_noise.p, f.proj_calib.p, f.proj_sensor.p, f.calib.p,
                       f.sensor.p, c.values.p, c.val_off.p, nt, f.proj_J.p);
  if (f.n_between)
    hipLaunchKernelGGL(k_lin_between, dim3(grid_for(f.n_between)), dim3(kBlock), 0, c.stream, f.n_between,
                       f.between_v1.p, f.between_v2.p, f.between_z.p, f.between_noise.p, c.var_type.p, c.values.p,
                       c.val_off.p, nt, f.between_J.p);
  if (f.n_prior)
    hipLaunchKernelGGL(k_lin_prior, dim3(grid_for(f.n_prior)), dim3(kBlock), 0, c.stream, f.n_prior, f.prior_var.p,
                       f.prior_off.p, f.prior_data.p, f.prior_noise.p, c.var_type.p, c.values.p, c.val_off.p, nt,
                       f.prior_J.p);
  check_hip(hipGetLastError(), "linearize");
}

// once per graph: the packed-value offsets of every GeneralSFM factor's camera and point (fused.h::SfmTabs)
__global__ __launch_bounds__(kBlock) void k_sfm_value_offsets(int64_t n, const int32_t* __restrict__ cam, const int32_t* __restrict__ pt,
                                                              const int64_t* __restrict__ val_off, int32_t* __restrict__ cam_at,
                                                              int32_t* __restrict__ pt_at) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {   // (grid_for caps the grid)
    cam_at[i] = (int32_t)val_off[cam[i]];
    pt_at[i] = (int32_t)val_off[pt[i]];
  }
}
void launch_sfm_value_offsets(gtg_context& c) {
  auto& f = c.f;
  if (!f.n_sfm) return;
  hipLaunchKernelGGL(k_sfm_value_offsets, dim3(grid_for(f.n_sfm)), dim3(kBlock), 0, c.stream, f.n_sfm, f.sfm_cam.p, f.sfm_point.p, c.val_off.p,
                     f.sfm_cam_at.p, f.sfm_point_at.p);
  check_hip(hipGetLastError(), "sfm_value_offsets");
}

// debug (gtg_get_jacobians on a graph whose GeneralSFM records are not stored): the records as k_lin_sfm writes them, into `dst`
void launch_sfm_records(gtg_context& c, double* dst) {
  auto& f = c.f;
  if (!f.n_sfm) return;
  hipLaunchKernelGGL(k_lin_sfm, dim3(grid_for(f.n_sfm)), dim3(kBlock), 0, c.stream, f.n_sfm, f.sfm_cam.p, f.sfm_point.p, f.sfm_z.p,
                     f.sfm_noise.p, c.values.p, c.val_off.p, noise_tab(c), dst, nullptr, nullptr);
  check_hip(hipGetLastError(), "sfm_records");
}

void launch_error(gtg_context& c, const double* values, int slot, const double* gate) {
  auto& f = c.f;
  ErrArgs a{f.n_sfm, f.n_proj, f.n_between, f.n_prior,
            f.sfm_cam.p, f.sfm_point.p, f.sfm_noise.p, f.sfm_z.p,
            f.proj_pose.p, f.proj_point.p, f.proj_noise.p, f.proj_calib.p, f.proj_sensor.p, f.proj_z.p, f.calib.p, f.sensor.p,
            f.between_v1.p, f.between_v2.p, f.between_noise.p, f.between_z.p,
            f.prior_var.p, f.prior_noise.p, f.prior_off.p, f.prior_data.p,
            c.var_type.p, c.val_off.p, c.n_smart ? c.sfm_smart.p : nullptr, c.smart_status.p};
  // the GeneralSFM factors and the other types in a kernel each, their block partials one behind the other
  const int64_t nrest = std::max(f.n_proj, std::max(f.n_between, f.n_prior));
  const int g1 = f.n_sfm ? std::min(grid_for(f.n_sfm), kMaxBlocks / 2) : 0, g2 = (nrest || !g1) ? std::min(grid_for(nrest), kMaxBlocks / 2) : 0;
  if (g1) hipLaunchKernelGGL(k_error<1>, dim3(g1), dim3(kBlock), 0, c.stream, a, values, noise_tab(c), c.partials.p);
  if (g2) hipLaunchKernelGGL(k_error<2>, dim3(g2), dim3(kBlock), 0, c.stream, a, values, noise_tab(c), c.partials.p + g1);
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(kBlock), 0, c.stream, c.partials.p, g1 + g2, 1, c.scalars.p, slot);
  if (c.n_smart)
    hipLaunchKernelGGL(k_error_smart_at_infinity, dim3(1), dim3(kBlock), 0, c.stream, f.n_sfm - c.smart_obs0, c.smart_obs0, f.sfm_cam.p, f.sfm_point.p,
                       f.sfm_z.p, f.sfm_noise.p, values, c.val_off.p, noise_tab(c), c.sfm_smart.p, c.smart_status.p,
                       gate, c.scalars.p, slot);
  check_hip(hipGetLastError(), "error");
}

void launch_linear_error(gtg_context& c) {
  auto& f = c.f;
  LinErrArgs a{f.n_sfm, f.n_proj, f.n_between, f.n_prior,
               f.sfm_cam.p, f.sfm_point.p, f.proj_pose.p, f.proj_point.p, f.between_v1.p, f.between_v2.p, f.prior_var.p,
               f.sfm_J.p, f.proj_J.p, f.between_J.p, f.prior_J.p, c.var_type.p, c.dim_off.p};
  const int64_t nmax = std::max(std::max(f.n_sfm, f.n_proj), std::max(f.n_between, f.n_prior));
  const int g = grid_for(nmax);
  if (c.fused_sfm) hipLaunchKernelGGL(k_linear_error<true>, dim3(g), dim3(kBlock), 0, c.stream, a, sfm_tabs(c), c.delta.p, c.partials.p);
  else hipLaunchKernelGGL(k_linear_error<false>, dim3(g), dim3(kBlock), 0, c.stream, a, sfm_tabs(c), c.delta.p, c.partials.p);
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(kBlock), 0, c.stream, c.partials.p, g, 2, c.scalars.p, (int)SC_LIN0);
  check_hip(hipGetLastError(), "linear_error");
}

void launch_retract(gtg_context& c) {
  hipLaunchKernelGGL(k_retract, dim3(grid_for(c.n_vars)), dim3(kBlock), 0, c.stream, c.n_vars, c.var_type.p,
                     c.val_off.p, c.dim_off.p, c.values.p, c.delta.p, c.trial.p);
  const int g = grid_for(c.user_dim_size);   // |delta| over the caller's variables (not the hidden landmarks of smart factors)
  hipLaunchKernelGGL(k_sumsq, dim3(g), dim3(kBlock), 0, c.stream, c.user_dim_size, c.delta.p, c.partials.p);
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(kBlock), 0, c.stream, c.partials.p, g, 1, c.scalars.p, (int)SC_DELTA_SQ);
  check_hip(hipGetLastError(), "retract");
}

}  // namespace gt

// ---- gtg_prewarm: this unit's kernels (kernels.h)
namespace {
void prewarm_factors(int) {
  gt::prewarm_kernels({(const void*)gt::k_lin_sfm, (const void*)gt::k_lin_proj, (const void*)gt::k_lin_between, (const void*)gt::k_lin_prior,
                       (const void*)gt::k_error<1>, (const void*)gt::k_error<2>, (const void*)gt::k_final_sum, (const void*)gt::k_linear_error<true>,
                       (const void*)gt::k_linear_error<false>, (const void*)gt::k_retract, (const void*)gt::k_sumsq, (const void*)gt::k_smart_triangulate,
                       (const void*)gt::k_lin_smart_at_infinity, (const void*)gt::k_error_smart_at_infinity, (const void*)gt::k_sfm_value_offsets});
}
gt::PrewarmUnit prewarm_factors_registered(prewarm_factors);
}  // namespace
