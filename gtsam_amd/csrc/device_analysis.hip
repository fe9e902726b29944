// device_analysis.hip -- the Schur term lists of the symbolic analysis, built on the device.
//
// What the host does in analysis.hip for every landmark -- every pair of its observations is one term E_a E_b^T of the block
// (row = the later position, column = the earlier one), terms grouped by block, the terms of a block in landmark order -- as
// four data-parallel passes (the formulation is pinned bit for bit against the host lists by tests/test_device_analysis_spec.py
// and, on the GPU, by tests/test_gpu_device_analysis.py):
//   k_da_count    one landmark per lane: number of terms = k (k + 1) / 2 + pairs of observations by the same camera
//   ExclusiveSum                       -> term offsets (emission order = landmark order)
//   k_da_emit     one landmark per lane: key = row position * n + column position, oriented (oa, ob), mirrored duplicates
//   DeviceRadixSort::SortPairs         stable, so the terms of a block keep the landmark order (= the summation order of
//                                      k_schur_pairs: the device's results do not depend on which side built the lists)
//   gather (oa, ob) into the final lists, DeviceRunLengthEncode::Encode -> the unique blocks, ExclusiveSum -> pair_ptr
// The 6 M terms of the L1723 shape take 2.8 ms (profiles/r02_device_analysis_proto.log) against 12 ms on 32 host threads, and
// 49 MB of term lists never cross PCIe; the host gets back the 0.2 M block keys and offsets it needs for the ordering and the
// tile schedule.  After the ordering the terms of the blocks whose orientation flips are swapped in place (k_da_flip).
// The host version stays: it serves the sharded upload (a shard needs the blocks of the WHOLE graph but only its own terms)
// and the dry-run runtime of the CPU tests, which cannot run kernels.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include <stdexcept>

#include "kernels.h"

namespace gt {

namespace {

__global__ __launch_bounds__(256) void k_da_count(int n_lm, const int64_t* __restrict__ ptr, const int32_t* __restrict__ lm_obs,
                                                  const int32_t* __restrict__ obs_pos, int64_t* __restrict__ cnt) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l > n_lm) return;
  if (l == n_lm) { cnt[l] = 0; return; }
  const int64_t b0 = ptr[l], k = ptr[l + 1] - b0;
  int64_t c = k * (k + 1) / 2;
  for (int64_t a = 1; a < k; a++) {
    const int pa = obs_pos[lm_obs[b0 + a]];
    for (int64_t b = 0; b < a; b++) c += obs_pos[lm_obs[b0 + b]] == pa;
  }
  cnt[l] = c;
}

__global__ __launch_bounds__(256) void k_da_emit(int n_lm, int nrv, const int64_t* __restrict__ ptr, const int32_t* __restrict__ lm_obs,
                                                 const int32_t* __restrict__ obs_pos, const int64_t* __restrict__ off,
                                                 uint64_t* __restrict__ key, uint32_t* __restrict__ idx, int32_t* __restrict__ oa_out,
                                                 int32_t* __restrict__ ob_out) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_lm) return;
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
__global__ __launch_bounds__(256) void k_da_flip(int64_t n_flip, const int64_t* __restrict__ which, const int64_t* __restrict__ pptr,
                                                 int32_t* __restrict__ oa, int32_t* __restrict__ ob) {
  const int64_t w = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6);
  if (w >= n_flip) return;
  const int64_t p = which[w];
  for (int64_t t = pptr[p] + (threadIdx.x & 63); t < pptr[p + 1]; t += 64) { const int32_t x = oa[t]; oa[t] = ob[t]; ob[t] = x; }
}

inline void hc(hipError_t e, const char* what) { check_hip(e, what); }

}  // namespace

// Fills c.pair_oa / c.pair_ob / c.pair_ptr (device, final buffers) from the landmark -> observation lists (already uploaded to
// c.lm_obs_ptr / c.lm_obs) and the positions of the observations' cameras; returns the unique block keys (row position * nrv +
// column position, ascending) and the term offsets of the blocks to the host.
void device_schur_terms(gtg_context& c, const std::vector<int32_t>& obs_pos, int nrv, std::vector<uint64_t>& block_keys,
                        std::vector<int64_t>& block_ptr) {
  hipStream_t s = c.stream;
  const int n_lm = c.n_lm;
  DevBuf<int32_t> d_pos; d_pos.upload(obs_pos.data(), obs_pos.size(), s);
  DevBuf<int64_t> d_cnt, d_off; d_cnt.alloc((size_t)n_lm + 1); d_off.alloc((size_t)n_lm + 1);
  size_t tmp_bytes = 0, need = 0; void* tmp = nullptr;
  auto ensure = [&](size_t n) { if (n > tmp_bytes) { if (tmp) (void)hipFree(tmp); hc(hipMalloc(&tmp, n), "hipMalloc"); tmp_bytes = n; } };
  hipLaunchKernelGGL(k_da_count, dim3((unsigned)((n_lm + 1 + 255) / 256)), dim3(256), 0, s, n_lm, c.lm_obs_ptr.p, c.lm_obs.p, d_pos.p, d_cnt.p);
  hc(rocprim::exclusive_scan(nullptr, need, d_cnt.p, d_off.p, (int64_t)0, (size_t)n_lm + 1, rocprim::plus<int64_t>(), s), "scan"); ensure(need);
  hc(rocprim::exclusive_scan(tmp, need, d_cnt.p, d_off.p, (int64_t)0, (size_t)n_lm + 1, rocprim::plus<int64_t>(), s), "scan");
  int64_t total = 0;
  hc(hipMemcpyAsync(&total, d_off.p + n_lm, sizeof(int64_t), hipMemcpyDeviceToHost, s), "D2H");
  hc(hipStreamSynchronize(s), "sync");
  if (total < 0 || total >= ((int64_t)1 << 31)) throw std::runtime_error("device analysis: term count out of range");
  c.n_pair_terms = total;
  c.pair_oa.alloc((size_t)std::max<int64_t>(total, 1)); c.pair_ob.alloc((size_t)std::max<int64_t>(total, 1));
  block_keys.clear(); block_ptr.assign(1, 0);
  if (total == 0) { d_pos.free(); d_cnt.free(); d_off.free(); if (tmp) (void)hipFree(tmp); c.pair_ptr.upload(block_ptr.data(), 1, s); return; }
  // one scratch allocation for everything that does not outlive the call (a dozen separate hipMalloc / hipFree of tens of
  // megabytes cost more than the kernels)
  const size_t N = (size_t)total;
  auto al = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
  int bits = 1;
  while (((uint64_t)1 << bits) < (uint64_t)nrv * (uint64_t)nrv) bits++;
  size_t need_sort = 0, need_rle = 0, need_scan = 0;
  // (rocPRIM: a stable LSD radix sort over the `bits` significant key bits -- stability is what keeps the landmark order inside a
  // block, i.e. the summation order of the Schur complement --, run-length encode, exclusive scan)
  hc(rocprim::radix_sort_pairs(nullptr, need_sort, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)total, 0u, (unsigned)bits, s), "sort");
  hc(rocprim::run_length_encode(nullptr, need_rle, (uint64_t*)nullptr, (unsigned)total, (uint64_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, s), "rle");
  hc(rocprim::exclusive_scan(nullptr, need_scan, (int32_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)total + 1, rocprim::plus<int64_t>(), s), "scan");
  const size_t need_tmp = std::max(need_sort, std::max(need_rle, need_scan));
  const size_t bytes = 3 * al(8 * N) + 4 * al(4 * N) + al(4 * (N + 1)) + al(8 * (N + 1)) + al(16) + al(need_tmp);
  char* pool = nullptr;
  hc(hipMalloc(reinterpret_cast<void**>(&pool), bytes), "hipMalloc");
  size_t at = 0;
  auto take = [&](size_t nbytes) { char* q = pool + at; at += al(nbytes); return q; };
  uint64_t* key = reinterpret_cast<uint64_t*>(take(8 * N)); uint64_t* key2 = reinterpret_cast<uint64_t*>(take(8 * N));
  uint64_t* uniq = reinterpret_cast<uint64_t*>(take(8 * N));
  int32_t* t_oa = reinterpret_cast<int32_t*>(take(4 * N)); int32_t* t_ob = reinterpret_cast<int32_t*>(take(4 * N));
  uint32_t* idx = reinterpret_cast<uint32_t*>(take(4 * N)); uint32_t* idx2 = reinterpret_cast<uint32_t*>(take(4 * N));
  int32_t* runs = reinterpret_cast<int32_t*>(take(4 * (N + 1))); int64_t* pp = reinterpret_cast<int64_t*>(take(8 * (N + 1)));
  int32_t* d_nruns = reinterpret_cast<int32_t*>(take(16));
  void* cub_tmp = take(need_tmp);
  hipLaunchKernelGGL(k_da_emit, dim3((unsigned)((n_lm + 255) / 256)), dim3(256), 0, s, n_lm, nrv, c.lm_obs_ptr.p, c.lm_obs.p, d_pos.p, d_off.p,
                     key, idx, t_oa, t_ob);
  need = need_sort; hc(rocprim::radix_sort_pairs(cub_tmp, need, key, key2, idx, idx2, (size_t)total, 0u, (unsigned)bits, s), "sort");
  hipLaunchKernelGGL(k_da_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, idx2, t_oa, t_ob, c.pair_oa.p, c.pair_ob.p);
  need = need_rle; hc(rocprim::run_length_encode(cub_tmp, need, key2, (unsigned)total, uniq, runs, d_nruns, s), "rle");
  int nruns = 0;
  hc(hipMemcpyAsync(&nruns, d_nruns, sizeof(int), hipMemcpyDeviceToHost, s), "D2H");
  hc(hipStreamSynchronize(s), "sync");
  hc(hipMemsetAsync(runs + nruns, 0, sizeof(int32_t), s), "memset");
  need = need_scan; hc(rocprim::exclusive_scan(cub_tmp, need, runs, pp, (int64_t)0, (size_t)nruns + 1, rocprim::plus<int64_t>(), s), "scan");   // int32 counts -> int64 offsets
  block_keys.resize((size_t)nruns); block_ptr.resize((size_t)nruns + 1);
  hc(hipMemcpyAsync(block_keys.data(), uniq, sizeof(uint64_t) * (size_t)nruns, hipMemcpyDeviceToHost, s), "D2H");
  hc(hipMemcpyAsync(block_ptr.data(), pp, sizeof(int64_t) * ((size_t)nruns + 1), hipMemcpyDeviceToHost, s), "D2H");
  c.pair_ptr.alloc((size_t)nruns + 1);
  hc(hipMemcpyAsync(c.pair_ptr.p, pp, sizeof(int64_t) * ((size_t)nruns + 1), hipMemcpyDeviceToDevice, s), "D2D");
  hc(hipStreamSynchronize(s), "sync");
  d_pos.free(); d_cnt.free(); d_off.free();
  if (tmp) (void)hipFree(tmp);
  (void)hipFree(pool);
}

// After the ordering: blocks whose row variable is now placed EARLIER than their column variable change orientation.
void device_flip_terms(gtg_context& c, const std::vector<int64_t>& flipped) {
  if (flipped.empty()) return;
  DevBuf<int64_t> d_which; d_which.upload(flipped.data(), flipped.size(), c.stream);
  hipLaunchKernelGGL(k_da_flip, dim3((unsigned)((flipped.size() + 3) / 4)), dim3(256), 0, c.stream, (int64_t)flipped.size(), d_which.p, c.pair_ptr.p,
                     c.pair_oa.p, c.pair_ob.p);
  check_hip(hipStreamSynchronize(c.stream), "sync");
  d_which.free();
}

}  // namespace gt
