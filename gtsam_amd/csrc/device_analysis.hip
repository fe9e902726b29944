// device_analysis.hip -- the Schur term lists of the symbolic analysis, built on the device.
//
// What the host does in analysis.hip for every landmark -- every pair of its observations is one term E_a E_b^T of the block
// (row = the later position, column = the earlier one), terms grouped by block, the terms of a block in landmark order -- as
// data-parallel passes (the formulation is pinned bit for bit against the host lists by tests/test_device_analysis_spec.py
// and, on the GPU, by tests/test_gpu_device_analysis.py):
//   k_da_nominal  one landmark per lane: k (k + 1) / 2 terms;  prim::exclusive_scan -> term offsets (emission order = landmark order)
//   k_da_dups     one NOMINAL PAIR per lane: a pair of two observations by the same camera is one more term of its landmark
//                 (none on the usual graphs; otherwise the offsets are summed again)
//   k_da_emit     one TERM per lane (round 5; rounds 3 - 4: one landmark per lane): the landmark by binary search in the offsets,
//                 the pair in closed form; key = row position * n + column position, oriented (oa, ob); landmarks with a double
//                 sighting: k_da_emit_dups, the per-landmark loop with its mirrored duplicates
//   prim::sort_pairs (primitives.hip)  stable, so the terms of a block keep the landmark order (= the summation order of
//                                      k_schur_pairs: the device's results do not depend on which side built the lists)
//   gather (oa, ob) into the final lists, prim::runs -> the unique blocks and pair_ptr (the first term of every block)
// The 6 M terms of the L1723 shape take 1.3 ms (2.8 with one landmark per lane, profiles/r02_device_analysis_proto.log) against 12 ms on 32 host threads, and
// 49 MB of term lists never cross PCIe; the host gets back the 0.2 M block keys and offsets it needs for the ordering and the
// tile schedule.  After the ordering the blocks whose orientation flips are swapped in place, indices and terms (k_da_orient).
// The incidence lists in front of that pass -- observation -> (reduced variable, landmark, position), landmark -> observations,
// reduced variable -> its factors -- are built here as well (device_incidence_lists: one kernel per observation with the role
// check of the factor keys, two stable radix sorts, CSR offsets by binary search in the sorted keys; 4.6 ms of host passes over the
// 0.68 M observations of the L1723 shape before).  Stable sorts keep the factor order inside every list, which is the order the
// host's counting sorts produce: the lists -- and with them every sum the device forms -- are the same whichever side built them.
// The host version stays: it serves the sharded upload (a shard needs the blocks of the WHOLE graph but only its own terms)
// and the dry-run runtime of the CPU tests, which cannot run kernels.
#include <cstring>
#include <stdexcept>

#include "kernels.h"
#include "primitives.h"

namespace gt {

namespace {

// The Schur terms of a landmark seen by k cameras: one per pair (a, b), b <= a, of its observations, in the order a = 0 .. k - 1,
// b = 0 .. a -- term i = a (a + 1) / 2 + b -- plus one more right behind a pair of two DIFFERENT observations by the SAME camera (rare:
// a camera that sees a landmark twice).  Rounds 3 - 4 walked that double loop with one lane per landmark, twice (count, then emit): the
// lanes of a wavefront waited for its longest track and every store was a scattered 4 / 8 bytes (0.78 + 1.85 ms on the L1723 shape).
// Now one lane per TERM: the landmark by binary search in the offsets, (a, b) in closed form, coalesced stores.  Landmarks with such a
// double sighting are counted by the same walk over the nominal pairs and emitted by the old per-landmark loop (k_da_emit_dups).
__device__ __forceinline__ int64_t tri(int64_t a) { return a * (a + 1) / 2; }
__device__ __forceinline__ int da_find(const int64_t* __restrict__ off, int n, int64_t t) {   // the l with off[l] <= t < off[l + 1]
  int lo = 0, hi = n;            // off[0] = 0 <= t < off[n]
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= t) lo = mid; else hi = mid; }
  return lo;
}
__device__ __forceinline__ void da_pair(int64_t i, int& a, int& b) {
  int64_t x = (int64_t)((sqrt(8.0 * (double)i + 1.0) - 1.0) * 0.5);
  while (tri(x + 1) <= i) x++;
  while (tri(x) > i) x--;
  a = (int)x; b = (int)(i - tri(x));
}
__global__ __launch_bounds__(256) void k_da_nominal(int n_lm, const int64_t* __restrict__ ptr, int64_t* __restrict__ cnt, int32_t* __restrict__ dup) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l > n_lm) return;
  if (l == n_lm) { cnt[l] = 0; dup[l] = 0; return; }     // (dup[n_lm]: the number of landmarks with a double sighting)
  cnt[l] = tri(ptr[l + 1] - ptr[l]); dup[l] = 0;
}
// one lane per nominal pair: a pair of two observations by the same camera adds a term to its landmark
__global__ __launch_bounds__(256) void k_da_dups(int64_t total0, int n_lm, const int64_t* __restrict__ off0, const int64_t* __restrict__ ptr,
                                                 const int32_t* __restrict__ lm_obs, const int32_t* __restrict__ obs_pos, int32_t* __restrict__ dup) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total0) return;
  const int l = da_find(off0, n_lm, t);
  int a, b; da_pair(t - off0[l], a, b);
  if (b == a) return;
  const int64_t b0 = ptr[l];
  if (obs_pos[lm_obs[b0 + a]] == obs_pos[lm_obs[b0 + b]]) { if (atomicAdd(dup + l, 1) == 0) atomicAdd(dup + n_lm, 1); }
}
__global__ __launch_bounds__(256) void k_da_add_dups(int n_lm, int64_t* __restrict__ cnt, const int32_t* __restrict__ dup) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l < n_lm) cnt[l] += dup[l];
}

// one lane per term (landmarks without a double sighting; the others: k_da_emit_dups)
__global__ __launch_bounds__(256) void k_da_emit(int64_t total, int n_lm, int nrv, const int64_t* __restrict__ ptr, const int32_t* __restrict__ lm_obs,
                                                 const int32_t* __restrict__ obs_pos, const int64_t* __restrict__ off, const int32_t* __restrict__ dup,
                                                 uint64_t* __restrict__ key, uint32_t* __restrict__ idx, int32_t* __restrict__ oa_out,
                                                 int32_t* __restrict__ ob_out) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= total) return;
  const int l = da_find(off, n_lm, w);
  if (dup[l]) return;
  int a, b; da_pair(w - off[l], a, b);
  const int64_t b0 = ptr[l];
  int32_t oa = lm_obs[b0 + a], ob = lm_obs[b0 + b];
  int pa = obs_pos[oa], pb = obs_pos[ob];
  if (pa < pb) { const int32_t t = oa; oa = ob; ob = t; const int u = pa; pa = pb; pb = u; }
  key[w] = (uint64_t)pa * (uint64_t)nrv + (uint64_t)pb; idx[w] = (uint32_t)w; oa_out[w] = oa; ob_out[w] = ob;
}
__global__ __launch_bounds__(256) void k_da_emit_dups(int n_lm, int nrv, const int64_t* __restrict__ ptr, const int32_t* __restrict__ lm_obs,
                                                      const int32_t* __restrict__ obs_pos, const int64_t* __restrict__ off, const int32_t* __restrict__ dup,
                                                      uint64_t* __restrict__ key, uint32_t* __restrict__ idx, int32_t* __restrict__ oa_out,
                                                      int32_t* __restrict__ ob_out) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_lm || !dup[l]) return;
  const int64_t b0 = ptr[l], k = ptr[l + 1] - b0;
  int64_t w = off[l];
  for (int64_t a = 0; a < k; a++) {
    const int32_t xa0 = lm_obs[b0 + a];
    const int pa0 = obs_pos[xa0];
    for (int64_t b = 0; b <= a; b++) {
      int32_t oa = xa0, ob = lm_obs[b0 + b];
      int pa = pa0, pb = obs_pos[ob];
      if (pa < pb) { const int32_t t = oa; oa = ob; ob = t; const int u = pa; pa = pb; pb = u; }
      const uint64_t kk = (uint64_t)pa * (uint64_t)nrv + (uint64_t)pb;
      key[w] = kk; idx[w] = (uint32_t)w; oa_out[w] = oa; ob_out[w] = ob; w++;
      if (pa == pb && oa != ob) { key[w] = kk; idx[w] = (uint32_t)w; oa_out[w] = ob; ob_out[w] = oa; w++; }   // same camera twice
    }
  }
}

__global__ __launch_bounds__(256) void k_da_gather(int64_t n, const uint32_t* __restrict__ idx, const int32_t* __restrict__ a,
                                                   const int32_t* __restrict__ b, int32_t* __restrict__ ao, int32_t* __restrict__ bo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { ao[i] = a[idx[i]]; bo[i] = b[idx[i]]; }
}

// one wavefront per flagged block: swap the two sides of its terms
// the block list as pairs of reduced indices: key = row position * n + column position (positions at the time of the emission)
__global__ __launch_bounds__(256) void k_da_split(int64_t n, int nrv, const uint64_t* __restrict__ key, const int32_t* __restrict__ pos_to_red,
                                                  int32_t* __restrict__ row, int32_t* __restrict__ col) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = key[i];
  row[i] = pos_to_red[k / (uint64_t)nrv]; col[i] = pos_to_red[k % (uint64_t)nrv];
}
// After the ordering, one wavefront per block: a block whose row variable is now placed EARLIER than its column variable changes
// orientation -- its two indices and the (oa, ob) of every one of its terms are swapped (the host used to walk the 0.2 M blocks, send the
// list of the flipped ones and k_da_flip swapped their terms)
__global__ __launch_bounds__(256) void k_da_orient(int64_t n, const int32_t* __restrict__ pos, int32_t* __restrict__ row, int32_t* __restrict__ col,
                                                   const int64_t* __restrict__ pptr, int32_t* __restrict__ oa, int32_t* __restrict__ ob) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const int32_t r = row[i], q = col[i];
  if (pos[r] >= pos[q]) return;
  for (int64_t t = pptr[i] + lane; t < pptr[i + 1]; t += 64) { const int32_t a = oa[t]; oa[t] = ob[t]; ob[t] = a; }
  if (lane == 0) { row[i] = q; col[i] = r; }
}
// one observation per lane: its reduced variable, landmark and position; the keys' roles (GeneralSFMFactor: SFM_CAMERA + POINT3,
// GenericProjectionFactor: POSE3 + POINT3) checked on the way -> bad[0] = 1 / 2
__global__ __launch_bounds__(256) void k_da_obs(int64_t n_sfm, int64_t n_proj, const int32_t* __restrict__ sfm_cam, const int32_t* __restrict__ sfm_point,
    const int32_t* __restrict__ proj_pose, const int32_t* __restrict__ proj_point, const int32_t* __restrict__ var_type,
    const int32_t* __restrict__ red_index, const int32_t* __restrict__ lm_index, const int32_t* __restrict__ red_pos,
    int32_t* __restrict__ obs_red, int32_t* __restrict__ obs_lm, int32_t* __restrict__ obs_pos, uint32_t* __restrict__ key,
    uint32_t* __restrict__ val, int32_t* __restrict__ bad) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_sfm + n_proj) return;
  const bool sfm = o < n_sfm;
  const int cam = sfm ? sfm_cam[o] : proj_pose[o - n_sfm], pt = sfm ? sfm_point[o] : proj_point[o - n_sfm];
  if (var_type[cam] != (sfm ? GTG_VAR_SFM_CAMERA : GTG_VAR_POSE3) || var_type[pt] != GTG_VAR_POINT3) { bad[0] = sfm ? 1 : 2; key[o] = 0; val[o] = (uint32_t)o; return; }
  const int r = red_index[cam], l = lm_index[pt];
  obs_red[o] = r; obs_lm[o] = l; obs_pos[o] = red_pos[r];
  key[o] = (uint32_t)l; val[o] = (uint32_t)o;
}
// the factors of every reduced variable, in the order the host lists them: all GeneralSFM observations, all projection observations,
// the between factors (factor i: its first key, then its second), the priors.  One entry per (factor, key); key = reduced index,
// or `none` for a key that is not a reduced variable (a prior on a landmark) -- those sort behind everything and are dropped.
__global__ __launch_bounds__(256) void k_da_inc_keys(int64_t n_sfm, int64_t n_proj, int64_t n_btw, int64_t n_pri, const int32_t* __restrict__ sfm_cam,
    const int32_t* __restrict__ proj_pose, const int32_t* __restrict__ bt1, const int32_t* __restrict__ bt2, const int32_t* __restrict__ pri_var,
    const int32_t* __restrict__ red_index, uint32_t none, uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b0 = n_sfm + n_proj, b1 = b0 + 2 * n_btw;
  if (e >= b1 + n_pri) return;
  int v;
  if (e < n_sfm) v = sfm_cam[e];
  else if (e < b0) v = proj_pose[e - n_sfm];
  else if (e < b1) { const int64_t j = e - b0; v = (j & 1) ? bt2[j >> 1] : bt1[j >> 1]; }
  else v = pri_var[e - b1];
  const int r = red_index[v];
  key[e] = r >= 0 ? (uint32_t)r : none;
  val[e] = (uint32_t)e;
}
__global__ __launch_bounds__(256) void k_da_inc_decode(int64_t n, int64_t n_sfm, int64_t n_proj, int64_t n_btw, const uint32_t* __restrict__ seq,
                                                       int32_t* __restrict__ kind, int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t e = seq[i], b0 = n_sfm + n_proj, b1 = b0 + 2 * n_btw;
  if (e < n_sfm) { kind[i] = 0; idx[i] = (int32_t)e; }
  else if (e < b0) { kind[i] = 1; idx[i] = (int32_t)(e - n_sfm); }
  else if (e < b1) { const int64_t j = e - b0; kind[i] = 2 + (int)(j & 1); idx[i] = (int32_t)(j >> 1); }
  else { kind[i] = 4; idx[i] = (int32_t)(e - b1); }
}
// CSR offsets of a sorted key array: ptr[g] = first position whose key is >= g, g = 0 .. n_groups
__global__ __launch_bounds__(256) void k_da_offsets(int64_t n, const uint32_t* __restrict__ sorted, int n_groups, int64_t* __restrict__ ptr) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > n_groups) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted[mid] < (uint32_t)g) lo = mid + 1; else hi = mid; }
  ptr[g] = lo;
}
__global__ __launch_bounds__(256) void k_da_u32_to_i32(int64_t n, const uint32_t* __restrict__ a, int32_t* __restrict__ b) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = (int32_t)a[i];
}

inline void hc(hipError_t e, const char* what) { check_hip(e, what); }

}  // namespace

// Fills c.pair_oa / c.pair_ob / c.pair_ptr (device, final buffers) from the landmark -> observation lists (already uploaded to
// c.lm_obs_ptr / c.lm_obs) and the positions of the observations' cameras; returns the unique block keys (row position * nrv +
// column position, ascending) and the term offsets of the blocks to the host.
void device_schur_terms(gtg_context& c, DevBuf<int32_t>& d_pos, int nrv, const std::vector<int32_t>& pos_to_red, std::vector<int32_t>& block_row,
                        std::vector<int32_t>& block_col, std::vector<int64_t>& block_ptr) {
  hipStream_t s = c.stream;
  const int n_lm = c.n_lm;
  DevBuf<int64_t> d_cnt, d_off; d_cnt.alloc((size_t)n_lm + 1); d_off.alloc((size_t)n_lm + 1);
  DevBuf<unsigned char> scan_tmp; scan_tmp.alloc(prim::scan_scratch_bytes((size_t)n_lm + 1));
  DevBuf<int32_t> d_dup; d_dup.alloc((size_t)n_lm + 1);
  hipLaunchKernelGGL(k_da_nominal, dim3((unsigned)((n_lm + 1 + 255) / 256)), dim3(256), 0, s, n_lm, c.lm_obs_ptr.p, d_cnt.p, d_dup.p);
  prim::exclusive_scan(d_cnt.p, d_off.p, (size_t)n_lm + 1, scan_tmp.p, s);
  int64_t total = 0;
  hc(hipMemcpyAsync(&total, d_off.p + n_lm, sizeof(int64_t), hipMemcpyDeviceToHost, s), "D2H");
  hc(hipStreamSynchronize(s), "sync");
  if (total < 0 || total >= ((int64_t)1 << 31)) throw std::runtime_error("device analysis: term count out of range");
  int32_t n_dup_lm = 0;
  if (total > 0) {   // landmarks that a camera sees twice have one more term per such pair: count them, and offset again if there are any
    hipLaunchKernelGGL(k_da_dups, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, n_lm, d_off.p, c.lm_obs_ptr.p, c.lm_obs.p, d_pos.p, d_dup.p);
    hc(hipMemcpyAsync(&n_dup_lm, d_dup.p + n_lm, sizeof(int32_t), hipMemcpyDeviceToHost, s), "D2H");
    hc(hipStreamSynchronize(s), "sync");
    if (n_dup_lm > 0) {
      hipLaunchKernelGGL(k_da_add_dups, dim3((unsigned)((n_lm + 255) / 256)), dim3(256), 0, s, n_lm, d_cnt.p, d_dup.p);
      prim::exclusive_scan(d_cnt.p, d_off.p, (size_t)n_lm + 1, scan_tmp.p, s);
      hc(hipMemcpyAsync(&total, d_off.p + n_lm, sizeof(int64_t), hipMemcpyDeviceToHost, s), "D2H");
      hc(hipStreamSynchronize(s), "sync");
      if (total < 0 || total >= ((int64_t)1 << 31)) throw std::runtime_error("device analysis: term count out of range");
    }
  }
  c.n_pair_terms = total;
  c.pair_oa.alloc((size_t)std::max<int64_t>(total, 1)); c.pair_ob.alloc((size_t)std::max<int64_t>(total, 1));
  block_row.clear(); block_col.clear(); block_ptr.assign(1, 0);
  if (total == 0) { d_pos.free(); d_cnt.free(); d_off.free(); d_dup.free(); scan_tmp.free(); c.pair_ptr.upload(block_ptr.data(), 1, s); c.pair_row.alloc(1); c.pair_col.alloc(1); return; }
  // one scratch allocation for everything that does not outlive the call (a dozen separate hipMalloc / hipFree of tens of
  // megabytes cost more than the kernels)
  const size_t N = (size_t)total;
  auto al = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
  int bits = 1;
  while (((uint64_t)1 << bits) < (uint64_t)nrv * (uint64_t)nrv) bits++;
  // (a stable LSD radix sort over the `bits` significant key bits -- stability is what keeps the landmark order inside a block, i.e. the
  // summation order of the Schur complement --, then the runs of the sorted keys: primitives.hip)
  const size_t need_tmp = std::max(prim::sort_scratch_bytes(N), prim::runs_scratch_bytes(N));
  const size_t bytes = 3 * al(8 * N) + 4 * al(4 * N) + al(8 * (N + 1)) + al(16) + al(need_tmp) + al(4 * (size_t)nrv);
  DevBuf<unsigned char> pool_buf; pool_buf.alloc(bytes);     // (through DevBuf: a block of this size is kept for the next handle, api.hip)
  char* pool = reinterpret_cast<char*>(pool_buf.p);
  size_t at = 0;
  auto take = [&](size_t nbytes) { char* q = pool + at; at += al(nbytes); return q; };
  uint64_t* key = reinterpret_cast<uint64_t*>(take(8 * N)); uint64_t* key2 = reinterpret_cast<uint64_t*>(take(8 * N));
  uint64_t* uniq = reinterpret_cast<uint64_t*>(take(8 * N));
  int32_t* t_oa = reinterpret_cast<int32_t*>(take(4 * N)); int32_t* t_ob = reinterpret_cast<int32_t*>(take(4 * N));
  uint32_t* idx = reinterpret_cast<uint32_t*>(take(4 * N)); uint32_t* idx2 = reinterpret_cast<uint32_t*>(take(4 * N));
  int64_t* pp = reinterpret_cast<int64_t*>(take(8 * (N + 1)));
  int32_t* d_nruns = reinterpret_cast<int32_t*>(take(16));
  void* prim_tmp = take(need_tmp);
  int32_t* d_p2r = reinterpret_cast<int32_t*>(take(4 * (size_t)nrv));   // position -> reduced index (k_da_split)
  hipLaunchKernelGGL(k_da_emit, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, n_lm, nrv, c.lm_obs_ptr.p, c.lm_obs.p, d_pos.p, d_off.p,
                     d_dup.p, key, idx, t_oa, t_ob);
  if (n_dup_lm > 0)
    hipLaunchKernelGGL(k_da_emit_dups, dim3((unsigned)((n_lm + 255) / 256)), dim3(256), 0, s, n_lm, nrv, c.lm_obs_ptr.p, c.lm_obs.p, d_pos.p, d_off.p,
                       d_dup.p, key, idx, t_oa, t_ob);
  prim::sort_pairs(key, key2, idx, idx2, N, bits, prim_tmp, s);
  hipLaunchKernelGGL(k_da_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, idx2, t_oa, t_ob, c.pair_oa.p, c.pair_ob.p);
  prim::runs(key2, N, uniq, pp, d_nruns, prim_tmp, s);     // uniq = the blocks, pp = the first term of every block (+ the total behind the last)
  int nruns = 0;
  hc(hipMemcpyAsync(&nruns, d_nruns, sizeof(int), hipMemcpyDeviceToHost, s), "D2H");
  hc(hipStreamSynchronize(s), "sync");
  // the blocks as (row, column) reduced indices: they STAY on the device (c.pair_row / c.pair_col: the ordering's edge list, the tile marks
  // and k_schur_pairs read them there; k_da_orient re-orients them after the ordering), the host gets a copy for its own passes
  block_row.resize((size_t)nruns); block_col.resize((size_t)nruns); block_ptr.resize((size_t)nruns + 1);
  c.pair_row.alloc(std::max<size_t>((size_t)nruns, 1)); c.pair_col.alloc(std::max<size_t>((size_t)nruns, 1));
  {
    hc(hipMemcpyAsync(d_p2r, pos_to_red.data(), sizeof(int32_t) * pos_to_red.size(), hipMemcpyHostToDevice, s), "H2D");
    if (nruns) hipLaunchKernelGGL(k_da_split, dim3((unsigned)((nruns + 255) / 256)), dim3(256), 0, s, (int64_t)nruns, nrv, uniq, d_p2r, c.pair_row.p, c.pair_col.p);
  }
  hc(hipMemcpyAsync(block_row.data(), c.pair_row.p, sizeof(int32_t) * (size_t)nruns, hipMemcpyDeviceToHost, s), "D2H");
  hc(hipMemcpyAsync(block_col.data(), c.pair_col.p, sizeof(int32_t) * (size_t)nruns, hipMemcpyDeviceToHost, s), "D2H");
  hc(hipMemcpyAsync(block_ptr.data(), pp, sizeof(int64_t) * ((size_t)nruns + 1), hipMemcpyDeviceToHost, s), "D2H");
  c.pair_ptr.alloc((size_t)nruns + 1);
  hc(hipMemcpyAsync(c.pair_ptr.p, pp, sizeof(int64_t) * ((size_t)nruns + 1), hipMemcpyDeviceToDevice, s), "D2D");
  hc(hipStreamSynchronize(s), "sync");
  d_pos.free(); d_cnt.free(); d_off.free(); d_dup.free(); scan_tmp.free();
  pool_buf.free();
}

// The incidence lists of the analysis, on the device (see the file header).  In: the factor tables (c.f.*), c.var_type; up: the
// variable -> landmark / reduced index maps and the positions of the reduced variables.  Out: c.obs_red, c.obs_lm, c.lm_obs_ptr,
// c.lm_obs, c.red_inc_ptr / kind / idx, and d_pos (observation -> position of its camera) for device_schur_terms.
// Throws std::invalid_argument for factor keys of the wrong variable type, with the host pass's messages.
void device_incidence_lists(gtg_context& c, const std::vector<int32_t>& red_pos, DevBuf<int32_t>& d_pos) {
  hipStream_t s = c.stream;
  auto& f = c.f;
  const int64_t n_sfm = f.n_sfm, n_proj = f.n_proj, n_btw = f.n_between, n_pri = f.n_prior, n_obs = n_sfm + n_proj;
  const int n_lm = c.n_lm, nrv = c.n_red_vars;
  const int64_t M = n_obs + 2 * n_btw + n_pri;
  if (n_obs >= ((int64_t)1 << 31) || M >= ((int64_t)1 << 31)) throw std::runtime_error("device analysis: too many factors");
  c.lm_index.upload(c.h_lm_index.data(), c.h_lm_index.size(), s); c.red_index.upload(c.h_red_index.data(), c.h_red_index.size(), s);
  DevBuf<int32_t> d_red_pos; d_red_pos.upload(red_pos.data(), red_pos.size(), s);
  c.obs_red.alloc((size_t)std::max<int64_t>(n_obs, 1)); c.obs_lm.alloc((size_t)std::max<int64_t>(n_obs, 1)); d_pos.alloc((size_t)std::max<int64_t>(n_obs, 1));
  c.lm_obs_ptr.alloc((size_t)n_lm + 1); c.lm_obs.alloc((size_t)std::max<int64_t>(n_obs, 1));
  c.red_inc_ptr.alloc((size_t)nrv + 1);
  auto al = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
  const size_t N = (size_t)std::max<int64_t>(M, 1);
  int bits_lm = 1, bits_red = 1;
  while (((uint64_t)1 << bits_lm) < (uint64_t)n_lm + 1) bits_lm++;
  while (((uint64_t)1 << bits_red) < (uint64_t)nrv + 2) bits_red++;
  const size_t need_tmp = prim::sort_scratch_bytes(N);
  DevBuf<unsigned char> pool_buf; pool_buf.alloc(4 * al(4 * N) + al(16) + al(need_tmp));
  char* pool = reinterpret_cast<char*>(pool_buf.p);
  size_t at = 0;
  auto take = [&](size_t nbytes) { char* q = pool + at; at += al(nbytes); return q; };
  uint32_t* key = reinterpret_cast<uint32_t*>(take(4 * N)); uint32_t* key2 = reinterpret_cast<uint32_t*>(take(4 * N));
  uint32_t* val = reinterpret_cast<uint32_t*>(take(4 * N)); uint32_t* val2 = reinterpret_cast<uint32_t*>(take(4 * N));
  int32_t* bad = reinterpret_cast<int32_t*>(take(16));
  void* tmp = take(need_tmp);
  hc(hipMemsetAsync(bad, 0, 16, s), "memset");
  auto grid = [](int64_t n) { return dim3((unsigned)((std::max<int64_t>(n, 1) + 255) / 256)); };
  if (n_obs) {
    hipLaunchKernelGGL(k_da_obs, grid(n_obs), dim3(256), 0, s, n_sfm, n_proj, f.sfm_cam.p, f.sfm_point.p, f.proj_pose.p, f.proj_point.p, c.var_type.p,
                       c.red_index.p, c.lm_index.p, d_red_pos.p, c.obs_red.p, c.obs_lm.p, d_pos.p, key, val, bad);
    prim::sort_pairs(key, key2, val, val2, (size_t)n_obs, bits_lm, tmp, s);
    hipLaunchKernelGGL(k_da_u32_to_i32, grid(n_obs), dim3(256), 0, s, n_obs, val2, c.lm_obs.p);
  }
  hipLaunchKernelGGL(k_da_offsets, grid(n_lm + 1), dim3(256), 0, s, n_obs, key2, n_lm, c.lm_obs_ptr.p);
  int32_t h_bad = 0;
  hc(hipMemcpyAsync(&h_bad, bad, sizeof(int32_t), hipMemcpyDeviceToHost, s), "D2H");
  hipLaunchKernelGGL(k_da_inc_keys, grid(M), dim3(256), 0, s, n_sfm, n_proj, n_btw, n_pri, f.sfm_cam.p, f.proj_pose.p, f.between_v1.p, f.between_v2.p,
                     f.prior_var.p, c.red_index.p, (uint32_t)nrv, key, val);
  prim::sort_pairs(key, key2, val, val2, (size_t)M, bits_red, tmp, s);
  hipLaunchKernelGGL(k_da_offsets, grid(nrv + 1), dim3(256), 0, s, M, key2, nrv, c.red_inc_ptr.p);
  int64_t n_inc = 0;
  hc(hipMemcpyAsync(&n_inc, c.red_inc_ptr.p + nrv, sizeof(int64_t), hipMemcpyDeviceToHost, s), "D2H");
  hc(hipStreamSynchronize(s), "sync");
  if (h_bad) {
    pool_buf.free(); d_red_pos.free();
    throw std::invalid_argument(h_bad == 1 ? "GeneralSFMFactor keys must be (SFM_CAMERA, POINT3)" : "GenericProjectionFactor keys must be (POSE3, POINT3)");
  }
  c.red_inc_kind.alloc((size_t)std::max<int64_t>(n_inc, 1)); c.red_inc_idx.alloc((size_t)std::max<int64_t>(n_inc, 1));
  hipLaunchKernelGGL(k_da_inc_decode, grid(n_inc), dim3(256), 0, s, n_inc, n_sfm, n_proj, n_btw, val2, c.red_inc_kind.p, c.red_inc_idx.p);
  hc(hipStreamSynchronize(s), "sync");
  pool_buf.free();
  d_red_pos.free();
}

// After the ordering: blocks whose row variable is now placed EARLIER than their column variable change orientation (c.pair_row / c.pair_col
// and the terms' (oa, ob), all in place on the device; red_pos = the new position of every reduced variable).
void device_orient_blocks(gtg_context& c, int64_t n_blocks, const std::vector<int32_t>& red_pos) {
  if (n_blocks == 0) return;
  DevBuf<int32_t> d_pos; d_pos.upload(red_pos.data(), red_pos.size(), c.stream);
  hipLaunchKernelGGL(k_da_orient, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, c.stream, n_blocks, d_pos.p, c.pair_row.p, c.pair_col.p, c.pair_ptr.p,
                     c.pair_oa.p, c.pair_ob.p);
  check_hip(hipStreamSynchronize(c.stream), "sync");
  d_pos.free();
}

// ---- the marks of the Schur blocks in the tile / strip structure of the reduced system (after the ordering) ---------------------------
// What analysis.hip's host loop does for every block of the block list (0.2 M on the L1723 shape, 3 ms): one lane per block sets the
// 128 x 128 tiles its d x d entries touch (T1, bytes), the 16-row strips (M16: column = strip, a bit set over the strips at and below it:
// the input of the strip-level symbolic factorisation) and adds the block's term to the commutative sum that identifies the block set
// (structure_hash: same mixing function as the host's, wrap-around 64-bit sum = order-free).  Blocks of a variable with itself (a
// camera's own Schur terms) only touch what the host marks for every variable's diagonal block anyway and are skipped.
namespace {
__global__ __launch_bounds__(256) void k_da_tile_marks(int64_t nb, const int32_t* __restrict__ row, const int32_t* __restrict__ col,
                                                       const RedLayout* __restrict__ lay, int nt, int w16, unsigned char* __restrict__ T1,
                                                       unsigned long long* __restrict__ M16, unsigned long long* __restrict__ hsum) {
  __shared__ unsigned long long wg_sum;
  if (threadIdx.x == 0) wg_sum = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb && row[i] != col[i]) {
    const RedLayout a = lay[row[i]], b = lay[col[i]];     // (row = the variable placed later: re-oriented by the host after the ordering)
    const int64_t a_end = a.off + a.dim - 1, b_end = b.off + b.dim - 1;
    for (int64_t ta = a.off / kTile; ta <= a_end / kTile; ta++)
      for (int64_t tb = b.off / kTile; tb <= b_end / kTile; tb++) T1[(ta > tb ? ta : tb) * nt + (ta > tb ? tb : ta)] = 1;
    for (int64_t sa = a.off / kSub; sa <= a_end / kSub; sa++)
      for (int64_t sb = b.off / kSub; sb <= b_end / kSub; sb++) {
        const int64_t hi = sa > sb ? sa : sb, lo = sa > sb ? sb : sa;
        atomicOr(&M16[lo * w16 + (hi >> 6)], 1ull << (hi & 63));
      }
    unsigned long long z = (unsigned long long)a.off * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)b.off + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full ^
                           (unsigned long long)(a.dim | (b.dim << 8));
    z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    atomicAdd(&wg_sum, z);
  }
  __syncthreads();
  if (threadIdx.x == 0 && wg_sum) atomicAdd(hsum, wg_sum);
}
}  // namespace

// In: c.pair_row / c.pair_col (device, re-oriented), the layout of the reduced variables.  Out: T1 (nt x nt) and M16 (n16 x w16) with the
// marks of the off-diagonal Schur blocks (the caller adds the diagonal blocks and the padding), *block_sum += the blocks' identity terms.
void device_tile_marks(gtg_context& c, int64_t n_blocks, const std::vector<RedLayout>& layout, int nt, int n16, int w16,
                       std::vector<uint8_t>& T1, std::vector<uint64_t>& M16, uint64_t* block_sum) {
  hipStream_t s = c.stream;
  const size_t b_t1 = ((size_t)nt * nt + 255) & ~(size_t)255, b_m16 = (size_t)n16 * w16 * 8, b_lay = (layout.size() * sizeof(RedLayout) + 255) & ~(size_t)255;
  DevBuf<unsigned char> ws; ws.alloc(b_t1 + b_m16 + 256 + b_lay);
  struct Release { DevBuf<unsigned char>& b; ~Release() { b.free(); } } release{ws};
  unsigned char* d_t1 = ws.p;
  unsigned long long* d_m16 = reinterpret_cast<unsigned long long*>(ws.p + b_t1);
  unsigned long long* d_sum = reinterpret_cast<unsigned long long*>(ws.p + b_t1 + b_m16);
  RedLayout* d_lay = reinterpret_cast<RedLayout*>(ws.p + b_t1 + b_m16 + 256);
  hc(hipMemsetAsync(ws.p, 0, b_t1 + b_m16 + 256, s), "memset");
  hc(hipMemcpyAsync(d_lay, layout.data(), layout.size() * sizeof(RedLayout), hipMemcpyHostToDevice, s), "H2D");
  hipLaunchKernelGGL(k_da_tile_marks, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, s, n_blocks, c.pair_row.p, c.pair_col.p, d_lay, nt, w16, d_t1, d_m16, d_sum);
  hc(hipGetLastError(), "tile marks");
  uint64_t sum = 0;
  T1.resize((size_t)nt * nt); M16.resize((size_t)n16 * w16);
  hc(hipMemcpyAsync(T1.data(), d_t1, (size_t)nt * nt, hipMemcpyDeviceToHost, s), "D2H");
  hc(hipMemcpyAsync(M16.data(), d_m16, b_m16, hipMemcpyDeviceToHost, s), "D2H");
  hc(hipMemcpyAsync(&sum, d_sum, sizeof(sum), hipMemcpyDeviceToHost, s), "D2H");
  hc(hipStreamSynchronize(s), "tile marks");
  *block_sum += sum;
}

// gtg_prewarm: this unit's kernels (kernels.h)
static void prewarm_device_analysis(int) {
  prewarm_kernels({(const void*)k_da_nominal, (const void*)k_da_dups, (const void*)k_da_add_dups, (const void*)k_da_emit, (const void*)k_da_emit_dups, (const void*)k_da_gather,
                   (const void*)k_da_split, (const void*)k_da_orient, (const void*)k_da_obs, (const void*)k_da_inc_keys, (const void*)k_da_inc_decode, (const void*)k_da_offsets, (const void*)k_da_u32_to_i32, (const void*)k_da_tile_marks});
}
static PrewarmUnit prewarm_device_analysis_registered(prewarm_device_analysis);

}  // namespace gt
