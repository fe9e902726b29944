// schur_groups.hip -- the Schur complement's sums, grouped (GTG_SCHUR=groups; NOT the default: built at the end of round 4, pinned on the
// CPU against its numpy statement, not yet run on hardware -- see DESIGN.md section 8).
//
// k_schur_pairs (assemble.hip) takes one wavefront per block pair (a, b) of the reduced system and fetches two 256-byte E slots per term
// from memory: 12.1 M scattered slot reads for the 6.07 M terms of the L1723 shape, 1.9 GB from beyond L2 for 0.17 GB of slots.  Here the
// cameras are cut into GROUPS of kSchurGroup = 8 consecutive positions of the elimination order.  A workgroup owns a PAIR of groups
// (ga >= gb) and walks its CELLS -- one per landmark seen from both groups: the landmark's observations in ga (A entries) and in gb (B
// entries) -- in landmark order, a CHUNK of cells at a time: every slot of the chunk is staged into LDS ONCE (16-byte loads, four slots
// per memory instruction) and read from there by all wavefronts.  4.6 M staged slots instead of 12.1 M reads (2.6 x fewer; counted in
// tests/test_schur_groups_spec.py and tools/schur_traffic_model.py), 3.6 x fewer vector-memory instructions per slot.
//
// Wavefront w owns ROW camera 8 ga + w: for each of its A entries it multiplies with every B entry of the cell (diagonal group pair: the
// B entries at positions <= its own) on the FP64 matrix core -- one v_mfma_f64_16x16x4 per term, as in k_schur_pairs -- into one of
// eight accumulators, chosen by the column camera.  Every block (a, b) therefore receives all its terms from ONE wavefront, in landmark
// order: the same additions in the same order as k_schur_pairs (bit-identical S unless a camera sees a landmark twice: then the four
// terms of that landmark in the diagonal block come in another order).  No atomics, no cross-workgroup sums: deterministic.
//
// Lists: analysis.hip::build_schur_groups (context.h::SchurGroups).  Group pairs are taken in descending order of their cell count.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <stdexcept>

#include "kernels.h"

namespace gt {

namespace {

#include "schur_groups_kernel.h"
#include "schur_groups_lists_kernel.h"

}  // namespace

void launch_schur_groups(gtg_context& c, SMat S) {
  const SchurGroups& g = c.sg;
  if (!g.active || g.n_pairs == 0) return;
  if (g.pipelined)
    hipLaunchKernelGGL(k_schur_groups_pipe, dim3((unsigned)g.n_pairs), dim3(kThreads), 0, c.stream, (int)g.n_pairs, g.NG, c.n_red_vars, g.order.p,
                       g.pair_key.p, g.pair_ptr.p, g.cell_a0.p, g.cell_b0.p, g.cell_pq.p, g.obs.p, g.pos_red.p, c.red_dim.p,
                       c.red_off.p, c.E.p, S);
  else
    hipLaunchKernelGGL(k_schur_groups, dim3((unsigned)g.n_pairs), dim3(kThreads), 0, c.stream, (int)g.n_pairs, g.NG, c.n_red_vars, g.order.p,
                       g.pair_key.p, g.pair_ptr.p, g.cell_a0.p, g.cell_b0.p, g.cell_pq.p, g.obs.p, g.pos_red.p, c.red_dim.p,
                       c.red_off.p, c.E.p, S);
  check_hip(hipGetLastError(), "schur (grouped)");
}


// The lists of analysis.hip::build_schur_groups on the device (GTG_SCHUR_LISTS=device; the host version is the default of the
// experimental GTG_SCHUR switch until this one has been compared with it on hardware -- same arrays, bit for bit: the per-landmark
// kernels are checked against the numpy statement on host threads, the sort / run-length encode / scan are the calls of
// device_analysis.hip::device_schur_terms).  In: c.lm_obs_ptr / c.lm_obs / c.obs_red on the device, the final positions (host).
bool device_schur_groups(gtg_context& c) {
  SchurGroups& g = c.sg;
  hipStream_t s = c.stream;
  const int G = kSchurGroup, nrv = c.n_red_vars, n_lm = c.n_lm;
  const int64_t n_obs = c.n_obs;
  if (nrv == 0 || n_obs == 0 || n_lm == 0 || n_obs >= ((int64_t)1 << 28)) return false;
  const int NG = (nrv + G - 1) / G;
  if ((int64_t)NG * NG >= ((int64_t)1 << 31)) return false;
  auto hc = [](hipError_t e, const char* what) { check_hip(e, what); };
  std::vector<int32_t> pos_red((size_t)nrv);
  for (int r = 0; r < nrv; r++) pos_red[(size_t)c.h_red_pos[r]] = r;
  DevBuf<int32_t> d_red_pos; d_red_pos.upload(c.h_red_pos.data(), c.h_red_pos.size(), s);
  DevBuf<int32_t> gpos; gpos.alloc((size_t)n_obs);
  DevBuf<int64_t> d_cnt, d_off; d_cnt.alloc((size_t)n_lm + 1); d_off.alloc((size_t)n_lm + 1);
  g.obs.alloc((size_t)n_obs);
  const dim3 grid_lm((unsigned)((n_lm + 1 + 255) / 256));
  hipLaunchKernelGGL(k_sg_sort_count, grid_lm, dim3(256), 0, s, n_lm, c.lm_obs_ptr.p, c.lm_obs.p, c.obs_red.p, d_red_pos.p, g.obs.p, gpos.p, d_cnt.p);
  size_t need = 0; void* tmp = nullptr; size_t tmp_bytes = 0;
  auto ensure = [&](size_t n) { if (n > tmp_bytes) { if (tmp) (void)hipFree(tmp); hc(hipMalloc(&tmp, n), "hipMalloc"); tmp_bytes = n; } };
  hc(rocprim::exclusive_scan(nullptr, need, d_cnt.p, d_off.p, (int64_t)0, (size_t)n_lm + 1, rocprim::plus<int64_t>(), s), "scan"); ensure(need);
  hc(rocprim::exclusive_scan(tmp, need, d_cnt.p, d_off.p, (int64_t)0, (size_t)n_lm + 1, rocprim::plus<int64_t>(), s), "scan");
  int64_t total = 0;
  hc(hipMemcpyAsync(&total, d_off.p + n_lm, sizeof(int64_t), hipMemcpyDeviceToHost, s), "D2H");
  hc(hipStreamSynchronize(s), "sync");
  auto cleanup = [&] { d_red_pos.free(); gpos.free(); d_cnt.free(); d_off.free(); if (tmp) (void)hipFree(tmp); };
  if (total <= 0 || total >= ((int64_t)1 << 31)) { cleanup(); g.obs.free(); return false; }
  const size_t N = (size_t)total;
  int bits = 1;
  while (((uint64_t)1 << bits) < (uint64_t)NG * (uint64_t)NG) bits++;
  DevBuf<int32_t> t_a0, t_b0, t_pq, d_bad, runs, d_nruns;
  DevBuf<int64_t> pp;
  t_a0.alloc(N); t_b0.alloc(N); t_pq.alloc(N); d_bad.alloc(4); runs.alloc(N + 1); d_nruns.alloc(4); pp.alloc(N + 1);
  // (uint32 keys / indices live in int32 buffers: DevBuf is instantiated for the signed type only)
  DevBuf<int32_t> k1, k2, i1, i2, uq;
  k1.alloc(N); k2.alloc(N); i1.alloc(N); i2.alloc(N); uq.alloc(N);
  hc(hipMemsetAsync(d_bad.p, 0, 16, s), "memset");
  hipLaunchKernelGGL(k_sg_emit, dim3((unsigned)((n_lm + 255) / 256)), dim3(256), 0, s, n_lm, NG, c.lm_obs_ptr.p, gpos.p, d_off.p,
                     reinterpret_cast<uint32_t*>(k1.p), reinterpret_cast<uint32_t*>(i1.p), t_a0.p, t_b0.p, t_pq.p, d_bad.p);
  size_t need_sort = 0, need_rle = 0, need_scan = 0;
  hc(rocprim::radix_sort_pairs(nullptr, need_sort, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, N, 0u, (unsigned)bits, s), "sort");
  hc(rocprim::run_length_encode(nullptr, need_rle, (uint32_t*)nullptr, (unsigned)total, (uint32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, s), "rle");
  hc(rocprim::exclusive_scan(nullptr, need_scan, (int32_t*)nullptr, (int64_t*)nullptr, (int64_t)0, N + 1, rocprim::plus<int64_t>(), s), "scan");
  ensure(std::max(need_sort, std::max(need_rle, need_scan)));
  need = need_sort;
  hc(rocprim::radix_sort_pairs(tmp, need, reinterpret_cast<uint32_t*>(k1.p), reinterpret_cast<uint32_t*>(k2.p), reinterpret_cast<uint32_t*>(i1.p),
                               reinterpret_cast<uint32_t*>(i2.p), N, 0u, (unsigned)bits, s), "sort");   // stable: landmark order inside a group pair
  g.cell_a0.alloc(N); g.cell_b0.alloc(N); g.cell_pq.alloc(N);
  hipLaunchKernelGGL(k_sg_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, reinterpret_cast<uint32_t*>(i2.p), t_a0.p, t_b0.p, t_pq.p,
                     g.cell_a0.p, g.cell_b0.p, g.cell_pq.p);
  need = need_rle;
  hc(rocprim::run_length_encode(tmp, need, reinterpret_cast<uint32_t*>(k2.p), (unsigned)total, reinterpret_cast<uint32_t*>(uq.p), runs.p, d_nruns.p, s), "rle");
  int nruns = 0, h_bad = 0;
  hc(hipMemcpyAsync(&nruns, d_nruns.p, sizeof(int), hipMemcpyDeviceToHost, s), "D2H");
  hc(hipMemcpyAsync(&h_bad, d_bad.p, sizeof(int), hipMemcpyDeviceToHost, s), "D2H");
  hc(hipStreamSynchronize(s), "sync");
  bool ok = h_bad == 0 && nruns > 0;
  if (ok) {
    hc(hipMemsetAsync(runs.p + nruns, 0, sizeof(int32_t), s), "memset");
    need = need_scan;
    hc(rocprim::exclusive_scan(tmp, need, runs.p, pp.p, (int64_t)0, (size_t)nruns + 1, rocprim::plus<int64_t>(), s), "scan");
    std::vector<int64_t> h_ptr((size_t)nruns + 1);
    hc(hipMemcpyAsync(h_ptr.data(), pp.p, sizeof(int64_t) * h_ptr.size(), hipMemcpyDeviceToHost, s), "D2H");
    g.pair_key.alloc((size_t)nruns); g.pair_ptr.alloc((size_t)nruns + 1);
    hc(hipMemcpyAsync(g.pair_key.p, uq.p, sizeof(int32_t) * (size_t)nruns, hipMemcpyDeviceToDevice, s), "D2D");
    hc(hipMemcpyAsync(g.pair_ptr.p, pp.p, sizeof(int64_t) * ((size_t)nruns + 1), hipMemcpyDeviceToDevice, s), "D2D");
    hc(hipStreamSynchronize(s), "sync");
    std::vector<int32_t> order((size_t)nruns);
    for (int i = 0; i < nruns; i++) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return h_ptr[(size_t)x + 1] - h_ptr[(size_t)x] > h_ptr[(size_t)y + 1] - h_ptr[(size_t)y]; });
    g.order.upload(order.data(), order.size(), s);
    g.pos_red.upload(pos_red.data(), pos_red.size(), s);
    g.NG = NG; g.n_pairs = nruns; g.n_cells = total; g.active = true;
    hc(hipStreamSynchronize(s), "sync");
  } else {
    g.obs.free(); g.cell_a0.free(); g.cell_b0.free(); g.cell_pq.free();
  }
  for (DevBuf<int32_t>* b : {&t_a0, &t_b0, &t_pq, &d_bad, &runs, &d_nruns, &k1, &k2, &i1, &i2, &uq}) b->free();
  pp.free();
  cleanup();
  return ok;
}

}  // namespace gt
