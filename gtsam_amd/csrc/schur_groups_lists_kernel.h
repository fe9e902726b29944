// schur_groups_lists_kernel.h -- the two per-landmark kernels of the DEVICE version of analysis.hip::build_schur_groups (schur_groups.hip::
// device_schur_groups; GTG_SCHUR_LISTS=device), in a header of their own so that tools/kernel_emu can run the same text on the host
// (tests/test_schur_groups_emulated.py checks them against the numpy statement of the lists).  One landmark per lane, no collectives.
#pragma once

// Pass 1: the landmark's observations, stably sorted by the FINAL position of their camera (insertion sort: tracks are short; the same
// order as the host's std::stable_sort), into gobs -- packed as observation | (position mod kSchurGroup) << 28 --, the positions into
// gpos (scratch of pass 2), and the number of cells of the landmark: d (d + 1) / 2 for d distinct groups.
__global__ __launch_bounds__(256) void k_sg_sort_count(int n_lm, const int64_t* __restrict__ ptr, const int32_t* __restrict__ lm_obs,
                                                       const int32_t* __restrict__ obs_red, const int32_t* __restrict__ red_pos,
                                                       int32_t* __restrict__ gobs, int32_t* __restrict__ gpos, int64_t* __restrict__ cnt) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l > n_lm) return;
  if (l == n_lm) { cnt[l] = 0; return; }
  const int64_t b = ptr[l];
  const int k = (int)(ptr[l + 1] - b);
  for (int i = 0; i < k; i++) {
    const int32_t o = lm_obs[b + i];
    const int32_t pos = red_pos[obs_red[o]];
    int j = i;
    while (j > 0 && gpos[b + j - 1] > pos) { gpos[b + j] = gpos[b + j - 1]; gobs[b + j] = gobs[b + j - 1]; j--; }
    gpos[b + j] = pos; gobs[b + j] = o;
  }
  int d = 0, last = -1;
  for (int i = 0; i < k; i++) {
    const int pos = gpos[b + i], grp = pos / kSchurGroup;
    if (grp != last) { d++; last = grp; }
    gobs[b + i] |= (pos % kSchurGroup) << 28;
  }
  cnt[l] = (int64_t)d * (d + 1) / 2;
}

// Pass 2: the landmark's cells at the offsets of the exclusive scan of the counts: for every group run ia (ascending), for every run
// ib <= ia: key = ga NG + gb, the runs' starts in gobs and their lengths p | q << 16; idx = the cell's own number (the stable sort by
// key carries it along for the gather).  bad[0] is raised for a run that is too long for the kernel's slot buffer / 16-bit lengths.
__global__ __launch_bounds__(256) void k_sg_emit(int n_lm, int NG, const int64_t* __restrict__ ptr, const int32_t* __restrict__ gpos,
                                                 const int64_t* __restrict__ off, uint32_t* __restrict__ key, uint32_t* __restrict__ idx,
                                                 int32_t* __restrict__ a0, int32_t* __restrict__ b0, int32_t* __restrict__ pq,
                                                 int32_t* __restrict__ bad) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_lm) return;
  const int64_t b = ptr[l];
  const int k = (int)(ptr[l + 1] - b);
  int64_t w = off[l];
  int sa = 0;
  while (sa < k) {
    const int ga = gpos[b + sa] / kSchurGroup;
    int ea = sa + 1;
    while (ea < k && gpos[b + ea] / kSchurGroup == ga) ea++;
    int sb = 0;
    while (sb <= sa) {
      const int gb = gpos[b + sb] / kSchurGroup;
      int eb = sb + 1;
      while (eb < k && gpos[b + eb] / kSchurGroup == gb) eb++;
      const int p = ea - sa, q = eb - sb;
      if (p > 0xffff || q > 0xffff || (sa == sb ? p : p + q) > kSchurChunkSlots) bad[0] = 1;
      key[w] = (uint32_t)ga * (uint32_t)NG + (uint32_t)gb; idx[w] = (uint32_t)w;
      a0[w] = (int32_t)(b + sa); b0[w] = (int32_t)(b + sb); pq[w] = p | (q << 16);
      w++;
      sb = eb;
    }
    sa = ea;
  }
}

// the cell descriptors in sorted order
__global__ __launch_bounds__(256) void k_sg_gather(int64_t n, const uint32_t* __restrict__ idx, const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                   const int32_t* __restrict__ c, int32_t* __restrict__ ao, int32_t* __restrict__ bo, int32_t* __restrict__ co) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const uint32_t s = idx[i]; ao[i] = a[s]; bo[i] = b[s]; co[i] = c[s]; }
}
