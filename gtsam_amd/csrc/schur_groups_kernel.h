// schur_groups_kernel.h -- the kernels of schur_groups.hip, in a header of their own so that tools/kernel_emu can compile the SAME text for
// the host (512 host threads per workgroup, wave collectives by rendezvous: tests/test_schur_groups_emulated.py).  Needs, from whoever
// includes it: kSchurGroup, kSchurChunkSlots, kEStride, double2, SMat::at_stored, the HIP thread indices, __syncthreads, __shfl_up,
// __ballot, __popcll and the two amdgcn builtins used below.  Included inside namespace gt { namespace { ... } }.
#pragma once

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kG = kSchurGroup;
constexpr int kNS = kSchurChunkSlots;
constexpr int kThreads = 64 * kG;       // 512
constexpr int kChunkCells = 64;         // one cell per lane of the wavefront that cuts the chunk
constexpr int kCamShift = 28;           // an entry of the sorted observation lists: observation | (position of its camera inside its group) << 28

static_assert(kG == 8, "the accumulator switch below is written for eight column cameras (3 bits per camera, one count byte each)");
static_assert(kNS <= 128, "slot indices and run lengths of a chunk are packed into bytes");

// The tables of one chunk of cells (LDS): the observation and the camera (position inside its group) of every staged slot, and for every
// row camera w the list of ITS terms' left factors in the order in which they are accumulated (cell order, entry order):
// entry = A slot | first B slot << 8 | number of B slots << 16.
struct SgTables {
  int32_t slot_obs[kNS];
  int32_t slot_cam[kNS];
  int32_t wl[kG][kNS];
  int32_t wn[kG];
  int32_t n_cells, n_slots;
};

// Wavefront 0 cuts the next chunk: as many of the next <= 64 cells (in order) as fit the slot buffer, one cell per lane; the positions of
// the per-camera list entries come from ONE prefix sum over the lanes of eight packed byte counters (a chunk has <= 128 A slots).
__device__ __forceinline__ void sg_cut_chunk(int lane, bool diag, int64_t c, int64_t cend, const int32_t* __restrict__ cell_a0,
                                             const int32_t* __restrict__ cell_b0, const int32_t* __restrict__ cell_pq,
                                             const int32_t* __restrict__ gobs, SgTables* t) {
  const bool have = c + lane < cend;
  int a0 = 0, b0 = 0, p = 0, q = 0;
  if (have) { a0 = cell_a0[c + lane]; b0 = cell_b0[c + lane]; const int pq = cell_pq[c + lane]; p = pq & 0xffff; q = (pq >> 16) & 0xffff; }
  const int need = have ? (diag ? p : p + q) : 0;
  int incl = need;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) { const int x = __shfl_up(incl, s, 64); if (lane >= s) incl += x; }
  const bool fits = have && incl <= kNS;
  const unsigned long long m = __ballot(fits);       // a prefix of the lanes: the sums are monotone
  const int n = __popcll(m);
  const bool mine = lane < n;
  const int base = incl - need;
  const int sb0 = diag ? base : base + p, sq = diag ? p : q;    // the cell's B run in the slot buffer
  unsigned long long cnt = 0;                          // byte w: this cell's A entries of row camera w
  if (mine) {
    for (int e = 0; e < p; e++) {
      const int v = gobs[a0 + e];
      const int cam = (int)((unsigned)v >> kCamShift);
      t->slot_obs[base + e] = v & ((1 << kCamShift) - 1); t->slot_cam[base + e] = cam;
      cnt += 1ull << (8 * cam);
    }
    if (!diag)
      for (int e = 0; e < q; e++) { const int v = gobs[b0 + e]; t->slot_obs[base + p + e] = v & ((1 << kCamShift) - 1); t->slot_cam[base + p + e] = (int)((unsigned)v >> kCamShift); }
  }
  unsigned long long run = cnt;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const unsigned lo = (unsigned)__shfl_up((int)(unsigned)(run & 0xffffffffull), s, 64), hi = (unsigned)__shfl_up((int)(unsigned)(run >> 32), s, 64);
    if (lane >= s) run += ((unsigned long long)hi << 32) | lo;     // (no byte overflows: <= 128 entries in all)
  }
  if (mine) {
    unsigned long long at = run - cnt;                 // exclusive: where this cell's entries start in every camera's list
    for (int e = 0; e < p; e++) {
      const int cam = t->slot_cam[base + e];
      const int k = (int)((at >> (8 * cam)) & 0xff);
      t->wl[cam][k] = (base + e) | (sb0 << 8) | (sq << 16);
      at += 1ull << (8 * cam);
    }
  }
  const int last = n > 0 ? n - 1 : 0;
  if (lane == last) {
    t->n_cells = n; t->n_slots = n > 0 ? incl : 0;
#pragma unroll
    for (int w = 0; w < kG; w++) t->wn[w] = n > 0 ? (int)((run >> (8 * w)) & 0xff) : 0;
  }
}

#define GT_SG_CASE(n) case n: acc##n = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc##n, 0, 0, 0); break;
// the terms of row camera w in one chunk: for every entry of its list, the A slot times every B slot of the cell (diagonal group pair:
// those of cameras at positions <= its own), one MFMA per term into the accumulator of the column camera
#define GT_SG_TERMS(T, SL)                                                                                              \
  {                                                                                                                     \
    const int nw = __builtin_amdgcn_readfirstlane((T)->wn[w]);                                                          \
    for (int k = 0; k < nw; k++) {                                                                                      \
      const int ent = __builtin_amdgcn_readfirstlane((T)->wl[w][k]);                                                    \
      const int as = ent & 0xff, b0 = (ent >> 8) & 0xff, q = (ent >> 16) & 0xff;                                        \
      const double a_raw = (SL)[as * kEStride + ea];                                                                    \
      const double av = ina ? a_raw : 0.0;                                                                              \
      for (int f = 0; f < q; f++) {                                                                                     \
        const int cb = __builtin_amdgcn_readfirstlane((T)->slot_cam[b0 + f]);                                           \
        if (diag && cb > w) continue;                                                                                   \
        const int db = (dbs >> (4 * cb)) & 15;                                                                          \
        const bool inb = lr < db && lk < 3;                                                                             \
        const double b_raw = (SL)[(b0 + f) * kEStride + (inb ? 3 * lr + lk : 0)];                                       \
        const double bv = inb ? b_raw : 0.0;                                                                            \
        touched |= 1u << cb;                                                                                            \
        switch (cb) { GT_SG_CASE(0) GT_SG_CASE(1) GT_SG_CASE(2) GT_SG_CASE(3) GT_SG_CASE(4) GT_SG_CASE(5) GT_SG_CASE(6) GT_SG_CASE(7) default: break; } \
      }                                                                                                                 \
    }                                                                                                                   \
  }
// S(a, b) -= the block's sum; accumulator register r holds C[row = lk + 4 r][col = lr] (as in k_schur_pairs)
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
// what both kernels start with: the group pair of this workgroup, this wavefront's row camera, the dimensions of the eight column cameras
#define GT_SG_PROLOGUE                                                                                                                    \
  if ((int)blockIdx.x >= n_pairs) return;                                                                                                 \
  const int j = order[blockIdx.x];                                                                                                        \
  const uint32_t key = (uint32_t)pair_key[j];                                                                                             \
  const int ga = (int)(key / (uint32_t)NG), gb = (int)(key % (uint32_t)NG);                                                               \
  const bool diag = ga == gb;                                                                                                             \
  const int tid = threadIdx.x, lane = tid & 63;                                                                                           \
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);                                                                                 \
  const int lr = lane & 15, lk = lane >> 4;                                                                                               \
  const int pa = kG * ga + w;                                                                                                             \
  const int ra = pa < nrv ? pos_red[pa] : -1;                                                                                             \
  const int da = ra >= 0 ? red_dim[ra] : 0;                                                                                               \
  const bool ina = lr < da && lk < 3;                                                                                                     \
  const int ea = ina ? 3 * lr + lk : 0;                                                                                                   \
  int dbs = 0; /* 4 bits per column camera (a group at the end of the order may be short) */                                              \
  for (int cb = 0; cb < kG; cb++) { const int pb = kG * gb + cb; if (pb < nrv) dbs |= red_dim[pos_red[pb]] << (4 * cb); }                 \
  v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0}, acc4 = {0, 0, 0, 0}, acc5 = {0, 0, 0, 0},       \
      acc6 = {0, 0, 0, 0}, acc7 = {0, 0, 0, 0};                                                                                           \
  unsigned touched = 0;                                                                                                                   \
  int64_t c = pair_ptr[j];                                                                                                                \
  const int64_t cend = pair_ptr[j + 1];
#define GT_SG_EPILOGUE                                                                                                                    \
  if (ra < 0) return;                                                                                                                     \
  const int64_t offa = red_off[ra];                                                                                                       \
  GT_SG_OUT(0) GT_SG_OUT(1) GT_SG_OUT(2) GT_SG_OUT(3) GT_SG_OUT(4) GT_SG_OUT(5) GT_SG_OUT(6) GT_SG_OUT(7)

__global__ __launch_bounds__(kThreads, 4) void k_schur_groups(int n_pairs, int NG, int nrv, const int32_t* __restrict__ order,
    const int32_t* __restrict__ pair_key, const int64_t* __restrict__ pair_ptr, const int32_t* __restrict__ cell_a0,
    const int32_t* __restrict__ cell_b0, const int32_t* __restrict__ cell_pq, const int32_t* __restrict__ gobs,
    const int32_t* __restrict__ pos_red, const int32_t* __restrict__ red_dim, const int64_t* __restrict__ red_off,
    const double* __restrict__ E, SMat S) {
  __shared__ __attribute__((aligned(16))) double slots[kNS * kEStride];   // 32 KB: the chunk's E slots
  __shared__ SgTables tab;
  GT_SG_PROLOGUE
  while (c < cend) {
    if (w == 0) sg_cut_chunk(lane, diag, c, cend, cell_a0, cell_b0, cell_pq, gobs, &tab);
    __syncthreads();
    const int n = tab.n_cells, ns = tab.n_slots;
    if (n == 0) break;   // (a cell that does not fit: build_schur_groups refuses such a graph; never spin)
    // ---- stage: 16 lanes x 16 bytes per slot, 32 slots per pass of the workgroup
    for (int sidx = tid >> 4; sidx < ns; sidx += kThreads / 16) {
      const int o = tab.slot_obs[sidx];
      const int part = tid & 15;
      *reinterpret_cast<double2*>(slots + sidx * kEStride + 2 * part) = *reinterpret_cast<const double2*>(E + (int64_t)kEStride * o + 2 * part);
    }
    __syncthreads();
    GT_SG_TERMS(&tab, slots)
    c += n;
    __syncthreads();   // the tables and the slots are rewritten by the next chunk
  }
  GT_SG_EPILOGUE
}

// ---- the same walk with the staging of chunk c + 1 under the multiplications of chunk c (GTG_SCHUR=groups_pipe) -------------------------
// Two slot buffers and two sets of tables.  Iteration c: wavefront 0 cuts chunk c + 1 (tables of the other set) | barrier | every lane
// REQUESTS its pieces of chunk c + 1 (<= 4 x 16 bytes into registers; nothing waits for them) | the terms of chunk c out of the current
// buffer | the registers go to the other buffer | barrier.  The heaviest workgroups are the diagonal group pairs (4 186 cells = 105 chunks
// on the L1723 shape, 31 000 cells on Venice): with the fetch latency (~2 us) under the multiplications a chunk costs its MFMAs and two
// barriers.  Same sums, same order: bit-identical to k_schur_groups.
constexpr int kStagePasses = kNS / (kThreads / 16);   // 16-byte pieces per lane and chunk (4)

__global__ __launch_bounds__(kThreads) void k_schur_groups_pipe(int n_pairs, int NG, int nrv, const int32_t* __restrict__ order,
    const int32_t* __restrict__ pair_key, const int64_t* __restrict__ pair_ptr, const int32_t* __restrict__ cell_a0,
    const int32_t* __restrict__ cell_b0, const int32_t* __restrict__ cell_pq, const int32_t* __restrict__ gobs,
    const int32_t* __restrict__ pos_red, const int32_t* __restrict__ red_dim, const int64_t* __restrict__ red_off,
    const double* __restrict__ E, SMat S) {
  __shared__ __attribute__((aligned(16))) double slots[2][kNS * kEStride];   // 2 x 32 KB
  __shared__ SgTables tab[2];
  GT_SG_PROLOGUE
  const int part = tid & 15, srow = tid >> 4;   // this lane's 16-byte piece of the slots srow, srow + 32, ...
  // ---- prologue: chunk 0 into buffer 0
  if (w == 0) sg_cut_chunk(lane, diag, c, cend, cell_a0, cell_b0, cell_pq, gobs, &tab[0]);
  __syncthreads();
  {
    const int ns = tab[0].n_slots;
    for (int sidx = srow; sidx < ns; sidx += kThreads / 16) {
      const int o = tab[0].slot_obs[sidx];
      *reinterpret_cast<double2*>(slots[0] + sidx * kEStride + 2 * part) = *reinterpret_cast<const double2*>(E + (int64_t)kEStride * o + 2 * part);
    }
  }
  __syncthreads();
  int cur = 0;
  for (;;) {
    const int n = tab[cur].n_cells;
    if (n == 0) break;                       // (the list is exhausted -- or a cell does not fit: build_schur_groups refuses such a graph)
    const int64_t cnext = c + n;
    const int nxt = cur ^ 1;
    // ---- cut chunk c + 1 (tables of the other set); an exhausted list cuts an empty chunk
    if (w == 0) sg_cut_chunk(lane, diag, cnext, cend, cell_a0, cell_b0, cell_pq, gobs, &tab[nxt]);
    __syncthreads();
    // ---- request chunk c + 1
    const int ns_next = tab[nxt].n_slots;
    double2 v[kStagePasses];
#pragma unroll
    for (int u = 0; u < kStagePasses; u++) {
      const int sidx = srow + u * (kThreads / 16);
      v[u].x = 0.0; v[u].y = 0.0;
      if (sidx < ns_next) v[u] = *reinterpret_cast<const double2*>(E + (int64_t)kEStride * tab[nxt].slot_obs[sidx] + 2 * part);
    }
    // ---- the terms of row camera w in chunk c
    GT_SG_TERMS(&tab[cur], slots[cur])
    // ---- chunk c + 1 into the other buffer (nobody reads it before the barrier; its last readers passed the barrier above)
#pragma unroll
    for (int u = 0; u < kStagePasses; u++) {
      const int sidx = srow + u * (kThreads / 16);
      if (sidx < ns_next) *reinterpret_cast<double2*>(slots[nxt] + sidx * kEStride + 2 * part) = v[u];
    }
    __syncthreads();
    c = cnext; cur = nxt;
  }
  GT_SG_EPILOGUE
}
#undef GT_SG_PROLOGUE
#undef GT_SG_EPILOGUE
#undef GT_SG_TERMS
#undef GT_SG_OUT
#undef GT_SG_CASE
