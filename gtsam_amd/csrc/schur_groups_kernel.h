// schur_groups_kernel.h -- the kernel of schur_groups.hip, in a header of its own so that tools/kernel_emu can compile the SAME text for the
// host (512 host threads per workgroup, wave collectives by rendezvous: tests/test_schur_groups_emulated.py).  Needs, from whoever
// includes it: kSchurGroup, kSchurChunkSlots, kEStride, double2, SMat::at_stored, the HIP thread indices, __syncthreads, __shfl_up,
// __ballot, __popcll and the two amdgcn builtins used below.  Included inside namespace gt { namespace { ... } }.
#pragma once

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kG = kSchurGroup;
constexpr int kNS = kSchurChunkSlots;
constexpr int kThreads = 64 * kG;       // 512
constexpr int kChunkCells = 64;         // one cell per lane of the wavefront that cuts the chunk

static_assert(kG == 8, "the accumulator switch below is written for eight column cameras");

#define GT_SG_CASE(n) case n: acc##n = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc##n, 0, 0, 0); break;

__global__ __launch_bounds__(kThreads) void k_schur_groups(int n_pairs, int NG, int nrv, const int32_t* __restrict__ order,
    const int32_t* __restrict__ pair_key, const int64_t* __restrict__ pair_ptr, const int32_t* __restrict__ cell_a0,
    const int32_t* __restrict__ cell_b0, const int32_t* __restrict__ cell_pq, const int32_t* __restrict__ gobs,
    const int32_t* __restrict__ obs_pos, const int32_t* __restrict__ pos_red, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, const double* __restrict__ E, SMat S) {
  __shared__ __attribute__((aligned(16))) double slots[kNS * kEStride];   // 32 KB: the chunk's E slots
  __shared__ int32_t slot_obs[kNS];
  __shared__ int32_t slot_cam[kNS];       // position of the slot's camera inside its group
  __shared__ int32_t cA0[kChunkCells], cP[kChunkCells], cB0[kChunkCells], cQ[kChunkCells];   // per cell: its A / B runs in `slots`
  __shared__ int32_t chunk_cells, chunk_slots;
  if ((int)blockIdx.x >= n_pairs) return;
  const int j = order[blockIdx.x];
  const uint32_t key = (uint32_t)pair_key[j];
  const int ga = (int)(key / (uint32_t)NG), gb = (int)(key % (uint32_t)NG);
  const bool diag = ga == gb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  // dimensions of the row camera and of the eight column cameras (a group at the end of the order may be short)
  const int pa = kG * ga + w;
  const int ra = pa < nrv ? pos_red[pa] : -1;
  const int da = ra >= 0 ? red_dim[ra] : 0;
  const bool ina = lr < da && lk < 3;
  const int ea = ina ? 3 * lr + lk : 0;
  int dbs = 0;   // 4 bits per column camera
  for (int cb = 0; cb < kG; cb++) { const int pb = kG * gb + cb; if (pb < nrv) dbs |= red_dim[pos_red[pb]] << (4 * cb); }
  v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0}, acc4 = {0, 0, 0, 0}, acc5 = {0, 0, 0, 0}, acc6 = {0, 0, 0, 0},
      acc7 = {0, 0, 0, 0};
  unsigned touched = 0;
  int64_t c = pair_ptr[j];
  const int64_t cend = pair_ptr[j + 1];
  while (c < cend) {
    // ---- cut the next chunk: as many of the next <= 64 cells (in order) as fit the slot buffer; wavefront 0, one cell per lane
    if (w == 0) {
      const bool have = c + lane < cend;
      int a0 = 0, b0 = 0, p = 0, q = 0;
      if (have) { a0 = cell_a0[c + lane]; b0 = cell_b0[c + lane]; const int pq = cell_pq[c + lane]; p = pq & 0xffff; q = (pq >> 16) & 0xffff; }
      const int need = have ? (diag ? p : p + q) : 0;
      int incl = need;
#pragma unroll
      for (int s = 1; s < 64; s <<= 1) { const int t = __shfl_up(incl, s, 64); if (lane >= s) incl += t; }
      const bool fits = have && incl <= kNS;
      const unsigned long long m = __ballot(fits);       // a prefix of the lanes: the sums are monotone
      const int n = __popcll(m);
      if (lane < n) {
        const int base = incl - need;
        cA0[lane] = base; cP[lane] = p; cB0[lane] = diag ? base : base + p; cQ[lane] = diag ? p : q;
        for (int e = 0; e < p; e++) slot_obs[base + e] = gobs[a0 + e];
        if (!diag) for (int e = 0; e < q; e++) slot_obs[base + p + e] = gobs[b0 + e];
      }
      if (lane == (n > 0 ? n - 1 : 0)) { chunk_cells = n; chunk_slots = n > 0 ? incl : 0; }
    }
    __syncthreads();
    const int n = chunk_cells, ns = chunk_slots;
    if (n == 0) break;   // (a cell that does not fit: build_schur_groups refuses such a graph; never spin)
    // ---- stage: 16 lanes x 16 bytes per slot, 32 slots per pass of the workgroup
    for (int sidx = tid >> 4; sidx < ns; sidx += kThreads / 16) {
      const int o = slot_obs[sidx];
      const int part = tid & 15;
      const double2 v = *reinterpret_cast<const double2*>(E + (int64_t)kEStride * o + 2 * part);
      *reinterpret_cast<double2*>(slots + sidx * kEStride + 2 * part) = v;
      if (part == 0) slot_cam[sidx] = obs_pos[o] % kG;
    }
    __syncthreads();
    // ---- the terms of row camera w
    for (int i = 0; i < n; i++) {
      const int a0 = __builtin_amdgcn_readfirstlane(cA0[i]), p = __builtin_amdgcn_readfirstlane(cP[i]);
      const int b0 = __builtin_amdgcn_readfirstlane(cB0[i]), q = __builtin_amdgcn_readfirstlane(cQ[i]);
      for (int e = 0; e < p; e++) {
        if (__builtin_amdgcn_readfirstlane(slot_cam[a0 + e]) != w) continue;
        const double a_raw = slots[(a0 + e) * kEStride + ea];
        const double av = ina ? a_raw : 0.0;
        for (int f = 0; f < q; f++) {
          const int cb = __builtin_amdgcn_readfirstlane(slot_cam[b0 + f]);
          if (diag && cb > w) continue;
          const int db = (dbs >> (4 * cb)) & 15;
          const bool inb = lr < db && lk < 3;
          const double b_raw = slots[(b0 + f) * kEStride + (inb ? 3 * lr + lk : 0)];
          const double bv = inb ? b_raw : 0.0;
          touched |= 1u << cb;
          switch (cb) { GT_SG_CASE(0) GT_SG_CASE(1) GT_SG_CASE(2) GT_SG_CASE(3) GT_SG_CASE(4) GT_SG_CASE(5) GT_SG_CASE(6) GT_SG_CASE(7) default: break; }
        }
      }
    }
    c += n;
    __syncthreads();   // the tables and the slots are rewritten by the next chunk
  }
  // ---- S(a, b) -= the block's sum; accumulator register r holds C[row = lk + 4 r][col = lr] (as in k_schur_pairs)
  if (ra < 0) return;
  const int64_t offa = red_off[ra];
#define GT_SG_OUT(nn)                                                                                         \
  if (touched & (1u << nn)) {                                                                                 \
    const int rb = pos_red[kG * gb + nn];                                                                     \
    const int db = (dbs >> (4 * nn)) & 15;                                                                    \
    const int64_t offb = red_off[rb];                                                                         \
    _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                           \
      const int row = lk + 4 * r;                                                                             \
      if (row < da && lr < db) if (double* qd = S.at_stored(offa + row, offb + lr)) *qd -= acc##nn[r];        \
    }                                                                                                         \
  }
  GT_SG_OUT(0) GT_SG_OUT(1) GT_SG_OUT(2) GT_SG_OUT(3) GT_SG_OUT(4) GT_SG_OUT(5) GT_SG_OUT(6) GT_SG_OUT(7)
#undef GT_SG_OUT
}

// ---- the same walk with the staging of chunk c + 1 under the multiplications of chunk c (GTG_SCHUR=groups_pipe) -------------------------
// Two slot buffers and two sets of tables.  Iteration c: wavefront 0 cuts chunk c + 1 (tables of the other set) | barrier | every lane
// REQUESTS its pieces of chunk c + 1 (<= 4 x 16 bytes into registers; nothing waits for them) | the terms of chunk c out of the current
// buffer | the registers go to the other buffer | barrier.  The heaviest workgroups are the diagonal group pairs (4 186 cells = 105 chunks
// on the L1723 shape, 31 000 cells on Venice): with the fetch latency (~2 us) under the multiplications a chunk costs its MFMAs and two
// barriers.  Same sums, same order: bit-identical to k_schur_groups.
constexpr int kStagePasses = kNS / (kThreads / 16);   // 16-byte pieces per lane and chunk (4)

__device__ __forceinline__ void sg_cut_chunk(int lane, bool diag, int64_t c, int64_t cend, const int32_t* __restrict__ cell_a0,
                                             const int32_t* __restrict__ cell_b0, const int32_t* __restrict__ cell_pq,
                                             const int32_t* __restrict__ gobs, int32_t* slot_obs, int32_t* cA0, int32_t* cP, int32_t* cB0,
                                             int32_t* cQ, int32_t* n_cells, int32_t* n_slots) {
  const bool have = c + lane < cend;
  int a0 = 0, b0 = 0, p = 0, q = 0;
  if (have) { a0 = cell_a0[c + lane]; b0 = cell_b0[c + lane]; const int pq = cell_pq[c + lane]; p = pq & 0xffff; q = (pq >> 16) & 0xffff; }
  const int need = have ? (diag ? p : p + q) : 0;
  int incl = need;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) { const int t = __shfl_up(incl, s, 64); if (lane >= s) incl += t; }
  const bool fits = have && incl <= kNS;
  const unsigned long long m = __ballot(fits);
  const int n = __popcll(m);
  if (lane < n) {
    const int base = incl - need;
    cA0[lane] = base; cP[lane] = p; cB0[lane] = diag ? base : base + p; cQ[lane] = diag ? p : q;
    for (int e = 0; e < p; e++) slot_obs[base + e] = gobs[a0 + e];
    if (!diag) for (int e = 0; e < q; e++) slot_obs[base + p + e] = gobs[b0 + e];
  }
  if (lane == (n > 0 ? n - 1 : 0)) { *n_cells = n; *n_slots = n > 0 ? incl : 0; }
}

__global__ __launch_bounds__(kThreads) void k_schur_groups_pipe(int n_pairs, int NG, int nrv, const int32_t* __restrict__ order,
    const int32_t* __restrict__ pair_key, const int64_t* __restrict__ pair_ptr, const int32_t* __restrict__ cell_a0,
    const int32_t* __restrict__ cell_b0, const int32_t* __restrict__ cell_pq, const int32_t* __restrict__ gobs,
    const int32_t* __restrict__ obs_pos, const int32_t* __restrict__ pos_red, const int32_t* __restrict__ red_dim,
    const int64_t* __restrict__ red_off, const double* __restrict__ E, SMat S) {
  __shared__ __attribute__((aligned(16))) double slots[2][kNS * kEStride];   // 2 x 32 KB
  __shared__ int32_t slot_obs[2][kNS];
  __shared__ int32_t slot_cam[2][kNS];
  __shared__ int32_t cA0[2][kChunkCells], cP[2][kChunkCells], cB0[2][kChunkCells], cQ[2][kChunkCells];
  __shared__ int32_t chunk_cells[2], chunk_slots[2];
  if ((int)blockIdx.x >= n_pairs) return;
  const int j = order[blockIdx.x];
  const uint32_t key = (uint32_t)pair_key[j];
  const int ga = (int)(key / (uint32_t)NG), gb = (int)(key % (uint32_t)NG);
  const bool diag = ga == gb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int pa = kG * ga + w;
  const int ra = pa < nrv ? pos_red[pa] : -1;
  const int da = ra >= 0 ? red_dim[ra] : 0;
  const bool ina = lr < da && lk < 3;
  const int ea = ina ? 3 * lr + lk : 0;
  int dbs = 0;
  for (int cb = 0; cb < kG; cb++) { const int pb = kG * gb + cb; if (pb < nrv) dbs |= red_dim[pos_red[pb]] << (4 * cb); }
  v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0}, acc4 = {0, 0, 0, 0}, acc5 = {0, 0, 0, 0}, acc6 = {0, 0, 0, 0},
      acc7 = {0, 0, 0, 0};
  unsigned touched = 0;
  int64_t c = pair_ptr[j];
  const int64_t cend = pair_ptr[j + 1];
  const int part = tid & 15, srow = tid >> 4;   // this lane's 16-byte piece of the slots srow, srow + 32, ...
  // ---- prologue: chunk 0 into buffer 0
  if (w == 0) sg_cut_chunk(lane, diag, c, cend, cell_a0, cell_b0, cell_pq, gobs, slot_obs[0], cA0[0], cP[0], cB0[0], cQ[0], &chunk_cells[0], &chunk_slots[0]);
  __syncthreads();
  {
    const int ns = chunk_slots[0];
    for (int sidx = srow; sidx < ns; sidx += kThreads / 16) {
      const int o = slot_obs[0][sidx];
      *reinterpret_cast<double2*>(slots[0] + sidx * kEStride + 2 * part) = *reinterpret_cast<const double2*>(E + (int64_t)kEStride * o + 2 * part);
      if (part == 0) slot_cam[0][sidx] = obs_pos[o] % kG;
    }
  }
  __syncthreads();
  int cur = 0;
  for (;;) {
    const int n = chunk_cells[cur];
    if (n == 0) break;                       // (a cell that does not fit: build_schur_groups refuses such a graph)
    const int64_t cnext = c + n;
    const int nxt = cur ^ 1;
    // ---- cut chunk c + 1 (tables of the other set); an exhausted list cuts an empty chunk
    if (w == 0) sg_cut_chunk(lane, diag, cnext, cend, cell_a0, cell_b0, cell_pq, gobs, slot_obs[nxt], cA0[nxt], cP[nxt], cB0[nxt], cQ[nxt], &chunk_cells[nxt], &chunk_slots[nxt]);
    __syncthreads();
    // ---- request chunk c + 1
    const int ns_next = chunk_slots[nxt];
    double2 v[kStagePasses];
    int vcam[kStagePasses];
#pragma unroll
    for (int u = 0; u < kStagePasses; u++) {
      const int sidx = srow + u * (kThreads / 16);
      v[u].x = 0.0; v[u].y = 0.0; vcam[u] = 0;
      if (sidx < ns_next) {
        const int o = slot_obs[nxt][sidx];
        v[u] = *reinterpret_cast<const double2*>(E + (int64_t)kEStride * o + 2 * part);
        if (part == 0) vcam[u] = obs_pos[o] % kG;
      }
    }
    // ---- the terms of row camera w in chunk c
    const double* sl = slots[cur];
    for (int i = 0; i < n; i++) {
      const int a0 = __builtin_amdgcn_readfirstlane(cA0[cur][i]), p = __builtin_amdgcn_readfirstlane(cP[cur][i]);
      const int b0 = __builtin_amdgcn_readfirstlane(cB0[cur][i]), q = __builtin_amdgcn_readfirstlane(cQ[cur][i]);
      for (int e = 0; e < p; e++) {
        if (__builtin_amdgcn_readfirstlane(slot_cam[cur][a0 + e]) != w) continue;
        const double a_raw = sl[(a0 + e) * kEStride + ea];
        const double av = ina ? a_raw : 0.0;
        for (int f = 0; f < q; f++) {
          const int cb = __builtin_amdgcn_readfirstlane(slot_cam[cur][b0 + f]);
          if (diag && cb > w) continue;
          const int db = (dbs >> (4 * cb)) & 15;
          const bool inb = lr < db && lk < 3;
          const double b_raw = sl[(b0 + f) * kEStride + (inb ? 3 * lr + lk : 0)];
          const double bv = inb ? b_raw : 0.0;
          touched |= 1u << cb;
          switch (cb) { GT_SG_CASE(0) GT_SG_CASE(1) GT_SG_CASE(2) GT_SG_CASE(3) GT_SG_CASE(4) GT_SG_CASE(5) GT_SG_CASE(6) GT_SG_CASE(7) default: break; }
        }
      }
    }
    // ---- chunk c + 1 into the other buffer (nobody reads it before the barrier; its last readers passed the barrier above)
#pragma unroll
    for (int u = 0; u < kStagePasses; u++) {
      const int sidx = srow + u * (kThreads / 16);
      if (sidx < ns_next) {
        *reinterpret_cast<double2*>(slots[nxt] + sidx * kEStride + 2 * part) = v[u];
        if (part == 0) slot_cam[nxt][sidx] = vcam[u];
      }
    }
    __syncthreads();
    c = cnext; cur = nxt;
  }
  if (ra < 0) return;
  const int64_t offa = red_off[ra];
#define GT_SG_OUT(nn)                                                                                         \
  if (touched & (1u << nn)) {                                                                                 \
    const int rb = pos_red[kG * gb + nn];                                                                     \
    const int db = (dbs >> (4 * nn)) & 15;                                                                    \
    const int64_t offb = red_off[rb];                                                                         \
    _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                           \
      const int row = lk + 4 * r;                                                                             \
      if (row < da && lr < db) if (double* qd = S.at_stored(offa + row, offb + lr)) *qd -= acc##nn[r];        \
    }                                                                                                         \
  }
  GT_SG_OUT(0) GT_SG_OUT(1) GT_SG_OUT(2) GT_SG_OUT(3) GT_SG_OUT(4) GT_SG_OUT(5) GT_SG_OUT(6) GT_SG_OUT(7)
#undef GT_SG_OUT
}

#undef GT_SG_CASE

