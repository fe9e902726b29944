// cholesky.hip -- dense FP64 Cholesky of the reduced (camera / pose) system + triangular solves.
//
// This is the frontal-matrix kernel of the path: what gtsam::choleskyPartial (base/cholesky.cpp:107-158:
// Eigen LLT + TRSM + SYRK) does on the root clique(s) of the reduced camera system, and
// GaussianBayesTree::optimize (linear/linearAlgorithms-inst.h:49-155) does for the back-substitution.
//
// Layout: S BY TILES (chol_device.h::SMat): one contiguous 128 x 128 slot per stored tile of the lower triangle, plus the
// tiles of one extra tile row whose row 0 holds the right-hand side: carrying it through TRSM + trailing updates performs the
// forward solve L y = g for free (the reference does the same with its augmented [H g; g^T f] matrix,
// HessianFactor.cpp:239-252).
//
// Right-looking over 128-wide block columns, processed in PAIRS so that the big trailing update
// contracts over K = 256 (arithmetic intensity 32 flop/B against the C tile traffic):
//   k_panel128   ONE launch per block column.  Workgroup 0 factors the 128x128 diagonal tile in LDS as a stream of four
//                32-column panels: one chain wavefront walks the pivots with only v_readlane + 1/pivot on the serial
//                path and publishes (column, 1/pivot) per pivot; follower wavefronts replay the elimination on the
//                32-row blocks below and on the identity, which yields the TRSM inside the tile and the 32x32
//                inverses for free; the level-3 parts run on the FP64 matrix cores.  The other workgroups are the
//                64-row TRSM workgroups of the stored tiles below (X = A L^-T by 4-phase block substitution with
//                MFMA); they run phase p as soon as workgroup 0 releases panel p (agent-scope release/acquire), so
//                only the last phase is left when the diagonal tile is done.
//   k_syrk       C(I,J) -= sum_k L(I,k) L(J,k)^T, 128x128 tiles (or 64x64 quadrants for the small launches on the
//                serial chain), K = 128 or 256, LDS-DMA double-buffered, v_mfma_f64_16x16x4_f64
//   k_inv_tiles, k_bwd_step   backward solve: batched 128x128 tile inverses, then one small launch per block row
// MFMA is used here and for the small contractions of the assembly (Schur pairs, camera blocks: assemble.hip);
// everything else on the path is HBM-bound.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "chol_device.h"

namespace gt {


// ---- TRSM: X = A(I,k) * L(k,k)^-T for every row tile I below the diagonal -----------------------------------
// 4-phase block substitution over the 32-column sub-blocks p: R_p = A_p - sum_{q<p} X_q L(p,q)^T, X_p = R_p Xinv_pp^T.
// The kernel sits on the serial panel chain and is bound by MFMA latency, not throughput (a wavefront keeps a single
// v_mfma_f64 in flight, ~140 cycles each), so the work is spread as thin as the 16x16 MFMA tile allows: a workgroup
// is 64 rows (66 KB of LDS: fits next to a k_syrk workgroup of the overlapped update) and 8 wavefronts, wavefront
// (rg, tj) owning the 16x16 tiles of rows 16 rg.. and column half tj of every sub-block: 80 MFMAs per wavefront
// instead of 320.  The L(p,q) / Xinv operands of a wavefront (80 doubles per lane) are fetched into registers up
// front, in flight together with the tile load.
constexpr int TR = 64;   // rows per TRSM workgroup
// a dependency wait inside a launch (TRSM workgroups behind workgroup 0; the backward sweep) gives up after 100 ms of the 100 MHz constant
// clock and raises fail[1] (SC_TIMEOUT: an error, never numbers) -- the bound used to be millions of polls, seconds per stuck wait
constexpr long long kEventWaitTicks = 10000000;
__device__ __forceinline__ void trsm_body(char* smem_raw, SMat S, int k, int wg,
                                          const int32_t* __restrict__ rows, double* __restrict__ Xinv,
                                          double* __restrict__ fail, long long epoch) {
  const long long flagbase = epoch * 8;
  double* Xs = reinterpret_cast<double*>(smem_raw);   // [TR][P]
  const int I = rows[wg >> 1], half_rows = (wg & 1) * TR;
  __builtin_amdgcn_s_setprio(2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int r0 = 16 * (wave >> 1), tj = wave & 1;
  double* tile = S.tile(I, k) + half_rows * T;
  const long long* flag = reinterpret_cast<const long long*>(Xinv + kFlagOff);
  {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = u * 512 + tid;
      v[u] = *reinterpret_cast<const double2*>(tile + (e / (T / 2)) * T + 2 * (e % (T / 2)));
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = u * 512 + tid;
      double* d = Xs + (e / (T / 2)) * P + 2 * (e % (T / 2));
      d[0] = v[u].x; d[1] = v[u].y;
    }
  }
#pragma unroll
  for (int p = 0; p < 4; p++) {
    // wait until workgroup 0 has released panel p of the diagonal tile (bounded: a lost release must not hang the GPU; the bound
    // is seconds, and running into it raises the time-out flag, which the C ABI turns into an error)
    if (tid == 0) {
      int spins = 0;
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < flagbase + p + 1) {
        // a scheduling problem, not a matrix property: fail[1] is reported as an error (SC_TIMEOUT), never as "not positive definite"
        if ((++spins & 255) == 0 && wall_clock64() - t0 > kEventWaitTicks) { __hip_atomic_store(fail + 1, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        // the word stuck in this XCD's L2 with its old value (chol_dataflow.hip::st_flag): the shadow word, published behind it
        if ((spins & 1023) == 0 && __hip_atomic_load(flag + 64, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= flagbase + p + 1) break;
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __syncthreads();   // also: the tile is in LDS (p = 0) / X_{p-1} is visible (p > 0)
    // this phase's operands (images written by workgroup 0): L(p,q), q < p, and Xinv_pp
    double bl[3][8], bx[8];
#pragma unroll
    for (int q = 0; q < p; q++) {
      const double2* op = reinterpret_cast<const double2*>(Xinv + kOpndBase + (((p * (p - 1) / 2 + q) * 2 + tj) * 64 + lane) * 8);
#pragma unroll
      for (int h = 0; h < 4; h++) { const double2 t = op[h]; bl[q][2 * h] = t.x; bl[q][2 * h + 1] = t.y; }
    }
    {
      const double2* op = reinterpret_cast<const double2*>(Xinv + kOpndBase + (((6 + p) * 2 + tj) * 64 + lane) * 8);
#pragma unroll
      for (int h = 0; h < 4; h++) { const double2 t = op[h]; bx[2 * h] = t.x; bx[2 * h + 1] = t.y; }
    }
    double* mine = Xs + (r0 + lk) * P + SB * p + 16 * tj + lr;   // accumulator layout: reg r <-> row r0 + lk + 4 r
    if (p > 0) {   // R_p = A_p - sum_{q<p} X_q L(p,q)^T   (p = 0: R_0 = A_0 is already in LDS)
      v4f64 acc;
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] = mine[4 * r * P];
#pragma unroll
      for (int q = 0; q < p; q++)
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++)
          acc = MFMA(-Xs[(r0 + lr) * P + SB * q + 8 * lk + s8], bl[q][s8], acc);
#pragma unroll
      for (int r = 0; r < 4; r++) mine[4 * r * P] = acc[r];
      __syncthreads();
    }
    v4f64 xr = (v4f64){0.0, 0.0, 0.0, 0.0};   // X_p = R_p Xinv_pp^T
#pragma unroll
    for (int s8 = 0; s8 < 8; s8++) xr = MFMA(Xs[(r0 + lr) * P + SB * p + 8 * lk + s8], bx[s8], xr);
    __syncthreads();   // both column halves have read R_p before it is overwritten
#pragma unroll
    for (int r = 0; r < 4; r++) mine[4 * r * P] = xr[r];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int e = u * 512 + tid;
    const int row = e / (T / 2), pc = e % (T / 2);
    double2 v;
    v.x = Xs[row * P + 2 * pc]; v.y = Xs[row * P + 2 * pc + 1];
    *reinterpret_cast<double2*>(tile + row * T + 2 * pc) = v;
  }
}

// ---- one block column = ONE launch: workgroup 0 factors the diagonal tile, workgroups 1.. are the 64-row TRSM
// workgroups of the stored tiles below.  They run their phase p (32 columns) as soon as workgroup 0 has released panel
// p of the diagonal tile, so that when the diagonal tile is done only the last phase of the TRSM is left: the TRSM's
// latency leaves the serial panel chain.  Workgroup 0 never waits for the others (no deadlock, whatever the dispatch
// order), the release/acquire pair is agent scope (L2 write-back / invalidate across XCDs).
__global__ __launch_bounds__(512, 2) void k_panel128(SMat S, int k, const int32_t* __restrict__ rows,
                                                     double* __restrict__ Xinv, double* __restrict__ fail,
                                                     long long* __restrict__ dbg, long long epoch,
                                                     const unsigned char* __restrict__ pivot_kind, double* __restrict__ tile_exp) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (blockIdx.x == 0) potrf_body(smem_raw, S.tile(k, k), k, Xinv, fail, dbg, epoch, reinterpret_cast<long long*>(Xinv + kFlagOff), 64, false, false, pivot_kind, tile_exp);
  else trsm_body(smem_raw, S, k, (int)blockIdx.x - 1, rows, Xinv, fail, epoch);
}
__global__ void k_set_epoch(long long* epoch, long long value) { *epoch = value; }

// ---- trailing update: C(I,J) -= sum_{kt} L(I,kt) L(J,kt)^T ---------------------------------------------------
// 128x128 output tile per workgroup, 4 wavefronts in 2x2, each 64x64 = 4x4 MFMA tiles.  The contraction
// runs over KT*128 columns in chunks of 32; each chunk of the two row panels (128 x 32 doubles = 32 KiB
// each) goes HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, stays in flight
// across the MFMA block), double buffered.  The LDS image is row-linear (256 B per row, as the DMA
// requires) with the 16-byte slots XOR-swizzled by (row & 15): the swizzle is applied to the SOURCE
// address of the DMA and to the operand reads, which makes the v_mfma operand reads (16 rows x 2 slots per
// half-wave) bank-conflict free.  C is loaded into the accumulators up front (A operand negated) so the
// epilogue is a pure store.
constexpr int KC = 16;             // 2 x 2 x 16 KiB of LDS per workgroup -> two workgroups per CU
constexpr int CHB = T * KC * 8;   // bytes of one panel chunk in LDS
constexpr int ROWB = KC * 8;      // bytes per LDS row (128)
constexpr int NSLOT = KC / 2;     // 16-byte slots per row (8)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// XCD-aware tile order: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md).  The host emits the (I,J) tile list
// of an update in 8x4-tile supertile order; each XCD takes one contiguous eighth of that list, so the 32
// workgroups resident on one XCD share a few row/column panels in that XCD's 4 MiB L2 instead of streaming
// 2 x 256 KB per tile from HBM.
constexpr int STI = 8, STJ = 4;

// Wave geometry: a wavefront can keep only ONE v_mfma_f64_16x16x4_f64 in flight (measured issue interval per wave
// 138 cycles, tools/mfma_f64_peak.hip: 36 TFLOP/s at 1 wave/SIMD, 47 at 2, 70 at 4), so the matrix pipe needs
// >= 4 waves per SIMD: 8 wavefronts per workgroup (4x2, each 32x64 = 2x4 MFMA tiles, 64 accumulator registers),
// two workgroups per CU.
// Q = 2 is the latency variant for the small launches on the serial panel chain (thin update, look-ahead columns):
// a workgroup owns a 64x64 quadrant (wavefront = 16x32), four times as many workgroups each a quarter as long.
template <int KT, int ABL = 0, int Q = 1>   // ABL: ablation bits for tools/ (1 no DMA, 2 no C load, 4 no C store, 8 no MFMA)
__global__ __launch_bounds__(512, 4) void k_syrk(SMat S, int ktile0,
                                                 const int32_t* __restrict__ pairs, int npairs) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [2 buffers][A chunk | B chunk]
  // the quadrant variant only runs on the serial chain: its waves win issue arbitration (MFMA pipe, LDS) against the
  // co-resident waves of the bulk update, which would otherwise double its latency
  if (Q > 1) __builtin_amdgcn_s_setprio(3);
  constexpr int RW = T / Q;              // rows (and columns) of the output block of a workgroup
  constexpr int TI = 2 / Q, TJ = 4 / Q;  // MFMA tiles per wavefront
  constexpr int CH = RW * KC * 8;        // bytes of one panel chunk in LDS
  constexpr int NQ = RW / 8 / 8;         // DMA instructions per wavefront and panel chunk
  const int nitems = npairs * Q * Q;
  const int nper = (nitems + 7) >> 3;
  const int item = (blockIdx.x & 7) * nper + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= nper || item >= nitems) return;
  const int idx = item / (Q * Q), qi = (item % (Q * Q)) / Q, qj = item % Q;
  const int I = pairs[2 * idx], J = pairs[2 * idx + 1];
  if (Q > 1 && I == J && qj > qi) return;   // strictly-upper quadrant of a diagonal tile: never read
  // the KT operand tiles of the two row panels (contraction over the block columns ktile0 .. ktile0 + KT - 1) and the target
  const double* Ap[KT]; const double* Bp[KT];
#pragma unroll
  for (int t = 0; t < KT; t++) { Ap[t] = S.tile(I, ktile0 + t) + qi * RW * T; Bp[t] = S.tile(J, ktile0 + t) + qj * RW * T; }
  double* C = S.tile(I, J) + qi * RW * T + qj * RW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;          // 4 x 2 waves: rows 16 TI wr .., cols 16 TJ wc ..
  const int lr = lane & 15, lk = lane >> 4;
  constexpr int NCH = KT * T / KC;

  // DMA map: one instruction = 1 KiB = 8 rows x 8 slots; instruction q of wave w fills rows 8(NQ w+q) .. +7;
  // lane l -> row + l/8, stored slot l%8, which holds logical slot (l%8) ^ ((row >> 1) & 7)
  const int drow = lane >> 3, dslot = lane & 7;
  auto stage = [&](int ch, int buf) {
    char* base = smem_raw + buf * 2 * CH;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int row = 8 * (NQ * wave + q) + drow;
      const int logical = dslot ^ ((row >> 1) & 7);
      const double* ga = Ap[(ch * KC) / T] + row * T + (ch * KC) % T + 2 * logical;
      const double* gb = Bp[(ch * KC) / T] + row * T + (ch * KC) % T + 2 * logical;
      __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)(base + (NQ * wave + q) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)gb, (lptr_t)(base + CH + (NQ * wave + q) * 1024), 16, 0, 0);
    }
  };

  if (!(ABL & 1)) stage(0, 0);
  v4f64 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ti++)
#pragma unroll
    for (int tj = 0; tj < TJ; tj++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        acc[ti][tj][r] = (ABL & 2) ? 0.0 : C[(wr * 16 * TI + ti * 16 + lk + 4 * r) * T + wc * 16 * TJ + tj * 16 + lr];
  __syncthreads();
  // operand byte offsets inside a chunk: row R = 16 TI wr (or 16 TJ wc) + 16 t + lr ((R >> 1) & 7 == lr >> 1), column kk + lk;
  // half-wave = 16 rows x 2 halves of one slot: bank = 32 (lr & 1) + 4 (slot ^ (lr >> 1)) + 2 (lk & 1): all distinct
  const int a_row_off = (wr * 16 * TI + lr) * ROWB, b_row_off = (wc * 16 * TJ + lr) * ROWB;
  const int half = (lk & 1) * 8, hi = lk >> 1, sw = lr >> 1;
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) {
    const int cur = ch & 1;
    if (ch + 1 < NCH && !(ABL & 1)) stage(ch + 1, cur ^ 1);
    const char* Ac = smem_raw + cur * 2 * CH;
    const char* Bc = Ac + CH;
#pragma unroll
    for (int kk = 0; kk < KC; kk += 4) {
      const int so = (((kk >> 1) + hi) ^ sw) * 16 + half;
      double a[TI], b[TJ];
#pragma unroll
      for (int t = 0; t < TI; t++) a[t] = -*reinterpret_cast<const double*>(Ac + a_row_off + t * 16 * ROWB + so);
#pragma unroll
      for (int t = 0; t < TJ; t++) b[t] = *reinterpret_cast<const double*>(Bc + b_row_off + t * 16 * ROWB + so);
#pragma unroll
      for (int ti = 0; ti < TI; ti++)
#pragma unroll
        for (int tj = 0; tj < TJ; tj++) { if (ABL & 8) acc[ti][tj][0] += a[ti] * b[tj]; else acc[ti][tj] = MFMA(a[ti], b[tj], acc[ti][tj]); }
    }
    __syncthreads();   // drains the DMA of chunk ch+1 (vmcnt) and fences the buffer just read
  }
#pragma unroll
  for (int ti = 0; ti < TI; ti++)
#pragma unroll
    for (int tj = 0; tj < TJ; tj++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (!(ABL & 4) || acc[ti][tj][r] == 123.456) C[(wr * 16 * TI + ti * 16 + lk + 4 * r) * T + wc * 16 * TJ + tj * 16 + lr] = acc[ti][tj][r];
}

long long* g_potrf_dbg = nullptr;   // debug: cycle stamps of the last k_potrf128 (see gtg_debug_potrf_stamps)
long long* g_potrf_dbg_set(long long* p) { g_potrf_dbg = p; return p; }

static inline int syrk_grid(int64_t npairs) { return (int)(((npairs + 7) / 8) * 8); }
constexpr int64_t kLatencyTiles = 320;   // launches up to this many tiles use the quadrant (latency) variant of k_syrk

// ---- host: tile schedule ------------------------------------------------------------------------------------
// Symbolic factorisation at the granularity of 256-wide column pairs (the unit the trailing update contracts
// over), then the per-pair tile lists.  This is the reduced system's elimination structure: what the reference
// rebuilds as EliminationTree + JunctionTree on every lambda try (inference/EliminationTree-inst.h:77-155,
// JunctionTree-inst.h:63-151), computed once here.
void build_chol_plan(CholPlan& plan, int nt, const std::vector<uint8_t>* pair_struct, hipStream_t stream,
                     const std::vector<int32_t>* pair_part, const std::vector<int32_t>* part_parent) {
  const int np = (nt + 1) / 2;
  const bool tree = pair_part && !pair_part->empty();
  plan.pair_part.clear(); plan.part_parent.clear();
  if (tree) { plan.pair_part = *pair_part; plan.part_parent = *part_parent; }
  plan.anc_off.assign(np, 0); plan.anc_cnt.assign(np, 0);
  std::vector<uint8_t> B((size_t)np * np, 0);
  for (int q = 0; q < np; q++)
    for (int p = 0; p <= q; p++) B[(size_t)q * np + p] = pair_struct ? (*pair_struct)[(size_t)q * np + p] : 1;
  for (int p = 0; p < np; p++) B[(size_t)p * np + p] = 1;
  for (int p = 0; p < np; p++) {   // symbolic elimination: the rows of column p become a clique
    std::vector<int> r;
    for (int q = p + 1; q < np; q++) if (B[(size_t)q * np + p]) r.push_back(q);
    for (size_t a = 0; a < r.size(); a++)
      for (size_t b = 0; b <= a; b++) B[(size_t)r[a] * np + r[b]] = 1;
  }
  auto tiles_of = [&](int q, std::vector<int32_t>& out) { out.push_back(2 * q); if (2 * q + 1 < nt) out.push_back(2 * q + 1); };
  std::vector<int32_t> rows, pairs, bcols, stored_list;
  plan.nt = nt;
  plan.trsm_off.assign(nt, 0); plan.trsm_cnt.assign(nt, 0);
  plan.s1_off.assign(np, 0); plan.s1_cnt.assign(np, 0); plan.nar_off.assign(np, 0); plan.nar_cnt.assign(np, 0);
  plan.rest_off.assign(np, 0); plan.rest_cnt.assign(np, 0);
  plan.bwd_off.assign(nt, 0); plan.bwd_cnt.assign(nt, 0);
  const double t3 = (double)T * T * T;
  double flops = 0.0; int64_t stored = 0;
  // (the SYRK pair lists of the stream schedule -- a sorted list of (I, J) per column pair, 0.4 M entries on the L1723 shape -- are built
  // by ensure_stream_lists when that schedule, the fall-back and A/B of the dataflow pass, is first used: 0.5 of this function's 0.7 ms)
  plan.h_fill = B; plan.stream_lists = false;
  for (int p = 0; p < np; p++) {
    const int k = 2 * p; const bool two = k + 1 < nt;
    std::vector<int32_t> R;
    for (int q = p + 1; q < np; q++) if (B[(size_t)q * np + p]) tiles_of(q, R);
    stored += (two ? 3 : 1) + (int64_t)R.size() * (two ? 2 : 1);
    for (int cc = k; cc <= (two ? k + 1 : k); cc++) {
      for (int rr = cc; rr <= (two ? k + 1 : k); rr++) { stored_list.push_back(rr); stored_list.push_back(cc); }
      for (int32_t I : R) { stored_list.push_back(I); stored_list.push_back(cc); }
      stored_list.push_back(nt); stored_list.push_back(cc);
    }
    // column k: rows below = [k+1] + R + [rhs]
    plan.trsm_off[k] = (int64_t)rows.size();
    if (two) rows.push_back(k + 1);
    rows.insert(rows.end(), R.begin(), R.end()); rows.push_back(nt);
    plan.trsm_cnt[k] = (int64_t)rows.size() - plan.trsm_off[k];
    flops += t3 / 3.0 + (double)(plan.trsm_cnt[k] - 1) * t3;
    if (two) {
      flops += (double)((int64_t)R.size() + 1) * 2.0 * t3;     // (the thin update's pairs: (k + 1, k + 1), (I, k + 1) for I in R; the rhs row's excluded)
      plan.trsm_off[k + 1] = (int64_t)rows.size();
      rows.insert(rows.end(), R.begin(), R.end()); rows.push_back(nt);
      plan.trsm_cnt[k + 1] = (int64_t)rows.size() - plan.trsm_off[k + 1];
      flops += t3 / 3.0 + (double)(plan.trsm_cnt[k + 1] - 1) * t3;
    }
    const double kk = two ? 2.0 : 1.0;
    flops += (double)((int64_t)R.size() * ((int64_t)R.size() + 1) / 2) * 2.0 * t3 * kk;   // rhs row excluded
  }
  for (int kt = 0; kt < nt; kt++) {   // backward solve: non-zero column tiles of row tile kt
    plan.bwd_off[kt] = (int64_t)bcols.size();
    const int q = kt / 2;
    for (int p = 0; p < q; p++) if (B[(size_t)q * np + p]) tiles_of(p, bcols);
    if (kt & 1) bcols.push_back(kt - 1);
    plan.bwd_cnt[kt] = (int64_t)bcols.size() - plan.bwd_off[kt];
  }
  {   // the same tiles by column, rows descending: what the one-launch backward sweep walks
    std::vector<std::vector<int32_t>> by_col(nt);
    for (int kt = nt - 1; kt >= 0; kt--)
      for (int64_t e = plan.bwd_off[kt]; e < plan.bwd_off[kt] + plan.bwd_cnt[kt]; e++) by_col[bcols[e]].push_back(kt);
    std::vector<int32_t> off(nt + 1, 0), list;
    for (int jt = 0; jt < nt; jt++) { list.insert(list.end(), by_col[jt].begin(), by_col[jt].end()); off[jt + 1] = (int32_t)list.size(); }
    if (list.empty()) list.push_back(0);
    plan.bwd_col_off.upload(off.data(), off.size(), stream);
    plan.bwd_col_rows.upload(list.data(), list.size(), stream);
  }
  plan.critical_pairs = np;
  if (tree) {   // longest leaf-to-root path in pairs
    const int nparts = (int)plan.part_parent.size();
    std::vector<int> len(nparts, 0), path(nparts, 0);
    for (int p = 0; p < np; p++) len[plan.pair_part[p]]++;
    int best = 0;
    for (int x = nparts - 1; x >= 0; x--) {   // parents have larger indices: walk down from the root
      const int par = plan.part_parent[x];
      path[x] = len[x] + (par >= 0 ? path[par] : 0);
      best = std::max(best, path[x]);
    }
    plan.critical_pairs = best;
  }
  plan.flops = flops;
  plan.dense_fraction = (double)stored / ((double)nt * (nt + 1) / 2.0);
  if (rows.empty()) rows.push_back(0);
  if (bcols.empty()) bcols.push_back(0);
  plan.rows.upload(rows.data(), rows.size(), stream);
  plan.pairs.free();
  plan.bcols.upload(bcols.data(), bcols.size(), stream);
  plan.n_stored = (int64_t)stored_list.size() / 2;
  plan.stored.upload(stored_list.data(), stored_list.size(), stream);
  // tile -> slot (chol_device.h::SMat): the stored tiles in list order
  plan.h_slot.assign((size_t)(nt + 1) * nt, -1);
  for (int64_t q = 0; q < plan.n_stored; q++) {
    int32_t& sl = plan.h_slot[(size_t)stored_list[2 * q] * nt + stored_list[2 * q + 1]];
    if (sl >= 0) throw std::runtime_error("cholesky plan: a tile is listed twice");
    sl = (int32_t)q;
  }
  plan.slot.upload(plan.h_slot.data(), plan.h_slot.size(), stream);
  check_hip(hipStreamSynchronize(stream), "plan upload");
}

// The stream schedule's SYRK pair lists (see build_chol_plan), from the fill structure the plan keeps: per column pair p the thin update
// of its second column (s1), the look-ahead part of the trailing update (targets in pair p + 1: nar), the rest, and -- with parts -- the
// targets in ancestor parts (anc); every list in supertile order.
void ensure_stream_lists(CholPlan& plan, hipStream_t stream) {
  if (plan.stream_lists) return;
  const int nt = plan.nt, np = (nt + 1) / 2;
  const bool tree = !plan.pair_part.empty();
  const std::vector<uint8_t>& B = plan.h_fill;
  std::vector<int32_t> pairs;
  auto tiles_of = [&](int q, std::vector<int32_t>& out) { out.push_back(2 * q); if (2 * q + 1 < nt) out.push_back(2 * q + 1); };
  auto emit_pairs = [&](std::vector<std::pair<int32_t, int32_t>>& v, int64_t& off, int64_t& cnt) {
    // supertile order: (J / STJ, I / STI, J % STJ, I % STI)
    std::sort(v.begin(), v.end(), [](const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) {
      const int64_t ka = (((int64_t)(a.second / STJ) * 65536 + a.first / STI) * STJ + a.second % STJ) * STI + a.first % STI;
      const int64_t kb = (((int64_t)(b.second / STJ) * 65536 + b.first / STI) * STJ + b.second % STJ) * STI + b.first % STI;
      return ka < kb; });
    off = (int64_t)pairs.size() / 2; cnt = (int64_t)v.size();
    for (auto& ij : v) { pairs.push_back(ij.first); pairs.push_back(ij.second); }
  };
  for (int p = 0; p < np; p++) {
    const int k = 2 * p; const bool two = k + 1 < nt;
    std::vector<int32_t> R;
    for (int q = p + 1; q < np; q++) if (B[(size_t)q * np + p]) tiles_of(q, R);
    if (two) {
      std::vector<std::pair<int32_t, int32_t>> s1;
      s1.emplace_back(k + 1, k + 1);
      for (int32_t I : R) s1.emplace_back(I, k + 1);
      s1.emplace_back(nt, k + 1);
      emit_pairs(s1, plan.s1_off[p], plan.s1_cnt[p]);
    }
    std::vector<std::pair<int32_t, int32_t>> nar, rest, anc;
    for (size_t b = 0; b < R.size(); b++) {
      const int32_t J = R[b];
      // with parts: targets in another part can only be in an ancestor (no tiles between independent subtrees)
      auto& dst = (tree && plan.pair_part[J / 2] != plan.pair_part[p]) ? anc : (J / 2 == p + 1) ? nar : rest;
      for (size_t a = b; a < R.size(); a++) dst.emplace_back(R[a], J);
      dst.emplace_back(nt, J);
    }
    emit_pairs(nar, plan.nar_off[p], plan.nar_cnt[p]);
    emit_pairs(rest, plan.rest_off[p], plan.rest_cnt[p]);
    emit_pairs(anc, plan.anc_off[p], plan.anc_cnt[p]);
  }
  if (pairs.empty()) { pairs.push_back(0); pairs.push_back(0); }
  plan.pairs.upload(pairs.data(), pairs.size(), stream);
  check_hip(hipStreamSynchronize(stream), "plan upload");
  plan.stream_lists = true;
}

// Multi-GPU exchange helper: copy the stored tiles of S into / out of a contiguous buffer, so that the all-reduce of
// the partial reduced systems carries only the stored lower tiles (not the dense (NP+128) x NP array).
__global__ __launch_bounds__(256) void k_pack_tiles(SMat S, const int32_t* __restrict__ tiles,
                                                    double* __restrict__ buf, int unpack) {
  const int I = tiles[2 * blockIdx.x], J = tiles[2 * blockIdx.x + 1];
  double* tile = S.tile(I, J);
  double* b = buf + (int64_t)blockIdx.x * T * T;
#pragma unroll 8
  for (int e = threadIdx.x; e < T * (T / 2); e += 256) {
    const int r = e / (T / 2), c2 = 2 * (e % (T / 2));
    double2* g = reinterpret_cast<double2*>(tile + r * T + c2);
    double2* p = reinterpret_cast<double2*>(b + r * T + c2);
    if (unpack) *g = *p; else *p = *g;
  }
}
// The same exchange at block granularity: the structurally non-zero d x d blocks of the reduced system (81-double slots),
// then the rhs row (NP doubles), then the padding diagonal.  HBM-bound gather / scatter of 72-byte row segments.
__global__ __launch_bounds__(256) void k_pack_blocks(SMat S, int NP, int64_t n_xb, const int64_t* __restrict__ row_off,
                                                     const int64_t* __restrict__ col_off, const int32_t* __restrict__ dims,
                                                     const int64_t* __restrict__ pad, int64_t npad, double* __restrict__ buf, int unpack) {
  const int64_t nblk = 81 * n_xb, total = nblk + NP + npad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    double* s;
    if (idx < nblk) {
      const int64_t b = idx / 81;
      const int e = (int)(idx - 81 * b), i = e / 9, j = e - 9 * i, d = dims[b];
      if (i >= (d & 255) || j >= (d >> 8)) { if (!unpack) buf[idx] = 0.0; continue; }
      s = S.at_stored(row_off[b] + i, col_off[b] + j);
      if (!s) { if (!unpack) buf[idx] = 0.0; continue; }   // (an entry of a straddling block in the upper triangle: no slot, never read)
    } else if (idx < nblk + NP) {
      s = S.at(NP, idx - nblk);
    } else {
      const int64_t q = pad[idx - nblk - NP];
      s = S.at(q, q);
    }
    if (unpack) *s = buf[idx]; else buf[idx] = *s;
  }
}
int64_t exchange_block_doubles(const gtg_context& c) { return 81 * c.n_xb + c.NP + (int64_t)c.h_pad_index.size(); }
void launch_pack_blocks(gtg_context& c, SMat S, int NP, double* buf, bool unpack) {
  const int64_t total = exchange_block_doubles(c);
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 64);
  hipLaunchKernelGGL(k_pack_blocks, dim3(grid), dim3(256), 0, c.stream, S, NP, c.n_xb, c.xb_row_off.p, c.xb_col_off.p, c.xb_dim.p,
                     c.pad_index.p, (int64_t)c.h_pad_index.size(), buf, unpack ? 1 : 0);
  check_hip(hipGetLastError(), "pack_blocks");
}
// zero the stored tiles of S (the others are never read): what the per-try rebuild of the reduced system needs
// (the slots ARE the stored tiles: one memset of the slot array)
void launch_zero_tiles(gtg_context& c, SMat S, const CholPlan& plan) {
  check_hip(hipMemsetAsync(S.p, 0, sizeof(double) * (size_t)plan.n_stored * TT, c.stream), "zero tiles");
}
void launch_pack_tiles(gtg_context& c, SMat S, const CholPlan& plan, double* buf, bool unpack) {
  hipLaunchKernelGGL(k_pack_tiles, dim3((unsigned)plan.n_exch), dim3(256), 0, c.stream, S, plan.exch.p, buf, unpack ? 1 : 0);
  check_hip(hipGetLastError(), "pack_tiles");
}

// Two-stream schedule with look-ahead.  Pair p = block columns (k, k+1); its trailing update is split by target column
// into nar(p) (the columns of pair p+1) and rest(p) (everything right of them: the bulk of the flops):
//   panel stream (high priority): panel(k) | thin update of column k+1 | panel(k+1) | wait P[p-1] | nar(p) -> event N[p]
//   update stream               : wait N[p]; rest(p)                                                  -> event P[p]
// The whole serial chain (panel, thin update, panel, look-ahead update, next panel) is in stream order on the panel
// stream.  Its only cross-stream dependency, "rest(p-1) has finished the columns of pair p+1", is a whole pair old
// when it is needed, so it costs one barrier packet (~3 us measured) instead of a fresh signal round trip (~13 us).
// rest(p) starts after nar(p) so that the two never share CUs (two small MFMA launches side by side double each
// other's latency).  panel(k) is ONE launch (k_panel128): the diagonal tile and, streamed behind it, the TRSM below.
void launch_cholesky(gtg_context& c, SMat S, int NP, const CholPlan& plan, double* Xinv, double* fail,
                     const unsigned char* pivot_kind, double* tile_exp) {
  CholStreams& g_cs = c.cs;
  const int nt = NP / T;
  if (plan.nt != nt) throw std::runtime_error("cholesky plan does not match the matrix");
  ensure_stream_lists(const_cast<CholPlan&>(plan), c.stream);   // (the pair lists are this schedule's alone: built on its first use)
  const size_t smem_potrf = sizeof(double) * kPotrfSmemDoubles;
  const size_t smem_trsm = sizeof(double) * (TR * P);
  const size_t smem_syrk = 4 * (size_t)CHB;
  static std::set<int> attr_set;   // function attributes are per device
  static std::mutex attr_mutex;
  std::unique_lock<std::mutex> attr_lock(attr_mutex);
  if (!attr_set.count(c.device)) {
    check_hip(hipFuncSetAttribute((const void*)k_panel128, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)std::max(smem_potrf, smem_trsm)), "smem attr");
    check_hip(hipFuncSetAttribute((const void*)k_syrk<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_syrk), "smem attr");
    check_hip(hipFuncSetAttribute((const void*)k_syrk<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_syrk), "smem attr");
    check_hip(hipFuncSetAttribute((const void*)k_syrk<1, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_syrk / 2), "smem attr");
    check_hip(hipFuncSetAttribute((const void*)k_syrk<2, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_syrk / 2), "smem attr");
    attr_set.insert(c.device);
  }
  attr_lock.unlock();
  const int npairs = (nt + 1) / 2;
  const long long flagbase = ++c.chol_epoch;   // host-counted, passed to every kernel by value (chol_device.h::potrf_body)
  hipLaunchKernelGGL(k_set_epoch, dim3(1), dim3(1), 0, c.stream, c.chol_epoch_dev.p, flagbase);
  if (!g_cs.panel) {
    int lo = 0, hi = 0;
    check_hip(hipDeviceGetStreamPriorityRange(&lo, &hi), "priority range");
    check_hip(hipStreamCreateWithPriority(&g_cs.panel, hipStreamNonBlocking, hi), "panel stream");
    check_hip(hipEventCreateWithFlags(&g_cs.start, hipEventDisableTiming), "event");
  }
  while ((int)g_cs.P.size() < npairs) {
    hipEvent_t e1, e2;
    check_hip(hipEventCreateWithFlags(&e1, hipEventDisableTiming), "event");
    check_hip(hipEventCreateWithFlags(&e2, hipEventDisableTiming), "event");
    g_cs.P.push_back(e1); g_cs.N.push_back(e2);
  }
  // (a CU-masked update stream that keeps CUs free for the chain was measured in round 1: no gain, removed)
  hipStream_t su = c.stream, sp = g_cs.panel;
  const int32_t* rows = plan.rows.p;
  const int32_t* pairs = plan.pairs.p;
  auto panel = [&](int k) {    // factor block column k: diagonal tile, then every stored row tile below
    double* Xk = Xinv + (size_t)k * T * T;
    hipLaunchKernelGGL(k_panel128, dim3(1 + 2 * (unsigned)plan.trsm_cnt[k]), dim3(512), std::max(smem_potrf, smem_trsm), sp,
                       S, k, rows + plan.trsm_off[k], Xk, fail, (long long*)g_potrf_dbg, flagbase, pivot_kind, tile_exp);
  };
  // everything queued on the update stream so far (building S) must precede the first panel
  check_hip(hipEventRecord(g_cs.start, c.stream), "record");
  check_hip(hipStreamWaitEvent(sp, g_cs.start, 0), "wait");
  auto update = [&](hipStream_t st, int k, const std::vector<int64_t>& off, const std::vector<int64_t>& cnt, int pi, bool latency) {
    if (cnt[pi] <= 0) return;
    if (latency && cnt[pi] <= kLatencyTiles)
      hipLaunchKernelGGL((k_syrk<2, 0, 2>), dim3(syrk_grid(4 * cnt[pi])), dim3(512), smem_syrk / 2, st, S, k,
                         pairs + 2 * off[pi], (int)cnt[pi]);
    else
      hipLaunchKernelGGL(k_syrk<2>, dim3(syrk_grid(cnt[pi])), dim3(512), smem_syrk, st, S, k, pairs + 2 * off[pi], (int)cnt[pi]);
  };
  // A plan with PARTS (nested-dissection ordering; build_chol_plan's `anc` lists = the updates that cross into a separator) runs as the
  // same single chain: the parts one after the other in elimination order, the cross-part updates behind the pair's own bulk updates
  // on the update stream.  What is new at a part boundary: the first panel of the next part waits for EVERYTHING older on the update
  // stream (the separator's columns collect `anc` updates from every pair of the subtrees below it, not only from the pair before).
  // (Rounds 1-4 ran the parts as independent chains on their own stream pairs here -- up to 18 streams; it was never faster than the
  // dataflow schedule, one run of it did not return on a shared box and was never explained (profiles/r04_streams_tree_hang.txt):
  // removed in round 5.  This schedule is the A/B of the dataflow pass and the last attempt after a timed-out one, not a fast path.)
  const bool tree = !plan.pair_part.empty();
  for (int pi = 0, k = 0; k < nt; k += 2, pi++) {
    if (tree && pi > 0 && plan.pair_part[pi] != plan.pair_part[pi - 1]) check_hip(hipStreamWaitEvent(sp, g_cs.P[pi - 1], 0), "wait");
    panel(k);
    if (k + 1 < nt) {
      if (plan.s1_cnt[pi] <= kLatencyTiles)
        hipLaunchKernelGGL((k_syrk<1, 0, 2>), dim3(syrk_grid(4 * plan.s1_cnt[pi])), dim3(512), smem_syrk / 2, sp, S, k,
                           pairs + 2 * plan.s1_off[pi], (int)plan.s1_cnt[pi]);
      else
        hipLaunchKernelGGL(k_syrk<1>, dim3(syrk_grid(plan.s1_cnt[pi])), dim3(512), smem_syrk, sp, S, k,
                           pairs + 2 * plan.s1_off[pi], (int)plan.s1_cnt[pi]);
      panel(k + 1);
      // the next pair's columns: everything older pairs owe them (rest(<= p-1), update stream) must be in; that event
      // is a whole pair old by now, so the wait costs a packet (~3 us), not a cross-stream round trip (~13 us)
      if (pi > 0) check_hip(hipStreamWaitEvent(sp, g_cs.P[pi - 1], 0), "wait");
      update(sp, k, plan.nar_off, plan.nar_cnt, pi, true);
    }
    check_hip(hipEventRecord(g_cs.N[pi], sp), "record");
    check_hip(hipStreamWaitEvent(su, g_cs.N[pi], 0), "wait");
    if (k + 1 < nt) update(su, k, plan.rest_off, plan.rest_cnt, pi, false);
    if (tree) update(su, k, plan.anc_off, plan.anc_cnt, pi, false);
    check_hip(hipEventRecord(g_cs.P[pi], su), "record");
  }
  check_hip(hipGetLastError(), "cholesky");
}

// debug (tools/syrk_ablation.py): time `reps` launches of the K=256 update over a dense m x m tile grid with ablations
float debug_time_syrk(gtg_context& c, SMat S, int m, int abl, int reps) {
  const size_t smem_syrk = 4 * (size_t)CHB;
  auto set = [&](const void* f) { check_hip(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_syrk), "smem attr"); };
  set((const void*)k_syrk<2, 0>); set((const void*)k_syrk<2, 1>); set((const void*)k_syrk<2, 3>); set((const void*)k_syrk<2, 7>); set((const void*)k_syrk<2, 6>); set((const void*)k_syrk<2, 15>);
  std::vector<int32_t> lst;
  for (int j = 0; j < m; j++) for (int i = j; i < m; i++) { lst.push_back(2 + i); lst.push_back(2 + j); }
  DevBuf<int32_t> dl; dl.upload(lst.data(), lst.size(), c.stream);
  const int np = (int)(lst.size() / 2);
  hipEvent_t e0, e1;
  check_hip(hipEventCreate(&e0), "event"); check_hip(hipEventCreate(&e1), "event");
  const dim3 grid(syrk_grid(np)), blk(512);
  check_hip(hipEventRecord(e0, c.stream), "record");
  for (int r = 0; r < reps; r++) {
    switch (abl) {
      case 0: hipLaunchKernelGGL((k_syrk<2, 0>), grid, blk, smem_syrk, c.stream, S, 0, dl.p, np); break;
      case 1: hipLaunchKernelGGL((k_syrk<2, 1>), grid, blk, smem_syrk, c.stream, S, 0, dl.p, np); break;
      case 3: hipLaunchKernelGGL((k_syrk<2, 3>), grid, blk, smem_syrk, c.stream, S, 0, dl.p, np); break;
      case 6: hipLaunchKernelGGL((k_syrk<2, 6>), grid, blk, smem_syrk, c.stream, S, 0, dl.p, np); break;
      case 7: hipLaunchKernelGGL((k_syrk<2, 7>), grid, blk, smem_syrk, c.stream, S, 0, dl.p, np); break;
      default: hipLaunchKernelGGL((k_syrk<2, 15>), grid, blk, smem_syrk, c.stream, S, 0, dl.p, np); break;
    }
  }
  check_hip(hipEventRecord(e1, c.stream), "record"); check_hip(hipStreamSynchronize(c.stream), "sync");
  float ms = 0;
  check_hip(hipEventElapsedTime(&ms, e0, e1), "elapsed");
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  dl.free();
  return ms / reps;
}

// ---- backward solve L^T x = y -------------------------------------------------------------------------
// The 122 steps (block rows, descending) are inherently serial, so each step is ONE small launch and does as little
// as possible:
//   k_inv_tiles   once per factorisation, all diagonal tiles in parallel: the strictly-lower 32x32 blocks of
//                 L(k,k)^-1 (the diagonal ones come out of k_panel128), written into the UPPER sub-blocks of the
//                 diagonal tile of S, which nobody reads: block position (q,p) holds Linv(p,q), p > q
//   k_bwd_step    step k: every workgroup first forms x_k = L(k,k)^-T y_k itself (a 128x128 mat-vec out of L2: cheaper
//                 than a second launch or a grid-wide dependency), then updates its 64 columns
//                 y[j] -= sum_r L(k*128 + r, j) x_k[r]  (row panel read coalesced along j); workgroup 0 stores x_k
__device__ __forceinline__ void blk_mul(double* __restrict__ C, const double* __restrict__ A, const double* __restrict__ B,
                                        double alpha, bool accumulate, int tid) {   // C (+)= alpha A B, 32x32 row-major
  const int i = tid >> 3, c0 = (tid & 7) * 4;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
  for (int kk = 0; kk < SB; kk++) {
    const double av = A[i * SB + kk];
#pragma unroll
    for (int u = 0; u < 4; u++) acc[u] += av * B[kk * SB + c0 + u];
  }
#pragma unroll
  for (int u = 0; u < 4; u++) C[i * SB + c0 + u] = (accumulate ? C[i * SB + c0 + u] : 0.0) + alpha * acc[u];
}

// An x entry that has not been produced yet: a signalling-NaN pattern no computation yields (arithmetic quiets NaNs).
constexpr unsigned long long kBwdUnset = 0x7FF4A5C3D2E1F00DULL;

__global__ __launch_bounds__(256) void k_inv_tiles(SMat S, int NP, const double* __restrict__ Xinv_all,
                                                   double* __restrict__ x) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* Lb = reinterpret_cast<double*>(smem_raw);   // [6] L(p,q), p > q, at p(p-1)/2 + q
  double* Xb = Lb + 6 * SB * SB;                       // [4] Linv(p,p)
  double* Wb = Xb + 4 * SB * SB;                       // [6] Linv(p,q), p > q
  double* Tm = Wb + 6 * SB * SB;                       // scratch
  const int k = blockIdx.x, tid = threadIdx.x;
  double* tile = S.tile(k, k);
  const double* Xinv = Xinv_all + (size_t)k * T * T;
  if (x && tid < T) {   // k_bwd_sweep waits on the entries themselves (and, when a wait drags on, on their shadow copies NP entries further on)
    reinterpret_cast<unsigned long long*>(x)[k * T + tid] = kBwdUnset;
    reinterpret_cast<unsigned long long*>(x)[NP + k * T + tid] = kBwdUnset;
  }
  for (int e = tid; e < 10 * 512; e += 256) {
    const int blk = e >> 9, w = e & 511, r = w >> 4, c2 = 2 * (w & 15);
    double2 v;
    if (blk < 6) {
      int p = 1, q = blk;
      while (q >= p) { q -= p; p++; }
      v = *reinterpret_cast<const double2*>(tile + (SB * p + r) * T + SB * q + c2);
      Lb[blk * SB * SB + r * SB + c2] = v.x; Lb[blk * SB * SB + r * SB + c2 + 1] = v.y;
    } else {
      v = *reinterpret_cast<const double2*>(Xinv + (blk - 6) * SB * SB + r * SB + c2);
      Xb[(blk - 6) * SB * SB + r * SB + c2] = v.x; Xb[(blk - 6) * SB * SB + r * SB + c2 + 1] = v.y;
    }
  }
  __syncthreads();
  // Linv(p,q) = -Linv(p,p) sum_{r=q}^{p-1} L(p,r) Linv(r,q), by increasing distance p - q
  for (int d = 1; d < 4; d++)
    for (int q = 0; q + d < 4; q++) {
      const int p = q + d;
      for (int r = q; r < p; r++) {
        const double* Lpr = Lb + (p * (p - 1) / 2 + r) * SB * SB;
        const double* Irq = (r == q) ? Xb + q * SB * SB : Wb + (r * (r - 1) / 2 + q) * SB * SB;
        blk_mul(Tm, Lpr, Irq, 1.0, r > q, tid);
        __syncthreads();
      }
      blk_mul(Wb + (p * (p - 1) / 2 + q) * SB * SB, Xb + p * SB * SB, Tm, -1.0, false, tid);
      __syncthreads();
    }
  for (int e = tid; e < 6 * 512; e += 256) {
    const int blk = e >> 9, w = e & 511, r = w >> 4, c2 = 2 * (w & 15);
    int p = 1, q = blk;
    while (q >= p) { q -= p; p++; }
    double2 v;
    v.x = Wb[blk * SB * SB + r * SB + c2]; v.y = Wb[blk * SB * SB + r * SB + c2 + 1];
    *reinterpret_cast<double2*>(tile + (SB * q + r) * T + SB * p + c2) = v;   // upper position (q,p)
  }
}

__global__ __launch_bounds__(256) void k_bwd_step(SMat S, int k, const int32_t* __restrict__ cols,
                                                  int ncols, const double* __restrict__ Xinv, double* __restrict__ y,
                                                  double* __restrict__ x) {
  __shared__ double ys[T];
  __shared__ double xh[2][T];
  __shared__ double xs[T];
  __shared__ double part[4][64];
  const int tid = threadIdx.x;
  const double* tile = S.tile(k, k);
  if (tid < T) ys[tid] = y[k * T + tid];
  __syncthreads();
  {  // x_q[cc] = sum_i Linv(q,q)[i][cc] y_q[i] + sum_{p>q} sum_i Linv(p,q)[i][cc] y_p[i]; two threads per entry (i halves)
    const int c = tid & (T - 1), hf = tid >> 7, q = c >> 5, cc = c & 31;
    double a0 = 0.0, a1 = 0.0;
    const double* Xq = Xinv + q * SB * SB + cc;
#pragma unroll 8
    for (int i = 16 * hf; i < 16 * hf + 16; i += 2) {
      a0 += Xq[i * SB] * ys[SB * q + i];
      a1 += Xq[(i + 1) * SB] * ys[SB * q + i + 1];
    }
    for (int p = q + 1; p < 4; p++) {
      const double* Wp = tile + (SB * q) * T + SB * p + cc;   // Linv(p,q) sits at upper position (q,p)
#pragma unroll 8
      for (int i = 16 * hf; i < 16 * hf + 16; i += 2) {
        a0 += Wp[i * T] * ys[SB * p + i];
        a1 += Wp[(i + 1) * T] * ys[SB * p + i + 1];
      }
    }
    xh[hf][c] = a0 + a1;
  }
  __syncthreads();
  if (tid < T) {
    const double v = xh[0][tid] + xh[1][tid];
    xs[tid] = v;
    if (blockIdx.x == 0) x[k * T + tid] = v;
  }
  __syncthreads();
  if (ncols == 0) return;
  const int c = tid & 63, g = tid >> 6;
  const int jt = cols[blockIdx.x >> 1], jc = (blockIdx.x & 1) * 64 + c;   // stored column tiles only
  const int j = jt * T + jc;
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  {
    const double* Lr = S.tile(k, jt) + (32 * g) * T + jc;
#pragma unroll
    for (int r = 0; r < 32; r += 4) {
      acc0 += Lr[r * T] * xs[32 * g + r];
      acc1 += Lr[(r + 1) * T] * xs[32 * g + r + 1];
      acc2 += Lr[(r + 2) * T] * xs[32 * g + r + 2];
      acc3 += Lr[(r + 3) * T] * xs[32 * g + r + 3];
    }
  }
  part[g][c] = (acc0 + acc1) + (acc2 + acc3);
  __syncthreads();
  if (g == 0) y[j] -= ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
}


// k_bwd_sweep: the whole backward solve in ONE launch.  Workgroup b owns block row j = nt - 1 - b of x (left-looking):
//   x_j = L(j,j)^-T ( y_j - sum_{i > j, (i,j) stored} L(i,j)^T x_i ),
// taking its tiles (i,j) in descending i, the order in which the x_i appear.  Dependencies travel through x itself: k_inv_tiles
// fills x with kBwdUnset, a producer stores its 128 entries write-through (aligned 8-byte stores: each entry is either unset or
// final), and a consumer's first wavefront polls the 128 entries it needs until none is unset -- one memory round trip per
// step of the chain instead of flag-then-data.  A workgroup only ever waits for workgroups with a smaller blockIdx, which the
// dispatcher started before it, so the sweep cannot deadlock however many workgroups are resident; a wait that runs into its
// bound raises fail[1] (reported as an error, never as numbers).  The tile of the NEXT dependency is already in registers when
// a wait ends (two register buffers), L(j,j)^-1 sits in LDS from the start: the serial chain x_{j+1} -> x_j costs one poll,
// two 128x128 mat-vecs out of registers / LDS and three barriers.
constexpr int kSweepThreads = 512;
constexpr size_t kSweepSmem = sizeof(double) * (10 * SB * SB + 2 * T + T + 8 * T);

__device__ __forceinline__ void sweep_load(double2 (&t)[16], const SMat& S, int i, int j, int g, int c2) {
  const double* src = S.tile(i, j) + (16 * g) * T + c2;
#pragma unroll
  for (int r = 0; r < 16; r++) t[r] = *reinterpret_cast<const double2*>(src + r * T);
}

__device__ __forceinline__ void sweep_wait(const double* x, int NP, int i, double* xs, double* fail, int tid) {
  if (tid < 64) {
    const unsigned long long* px = reinterpret_cast<const unsigned long long*>(x) + (int64_t)i * T + 2 * tid;
    unsigned long long a, b;
    const long long t0 = wall_clock64();
    for (int spins = 0;; spins++) {
      a = __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b = __hip_atomic_load(px + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(a != kBwdUnset && b != kBwdUnset)) break;
      if ((spins & 255) == 255 && wall_clock64() - t0 > kEventWaitTicks) { if (tid == 0) __hip_atomic_store(fail + 1, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      if ((spins & 1023) == 1023) {   // the entries stuck in this XCD's L2 as unset (chol_dataflow.hip::st_flag): their shadow copies
        a = __hip_atomic_load(px + NP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = __hip_atomic_load(px + NP + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(a != kBwdUnset && b != kBwdUnset)) break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    xs[2 * tid] = __longlong_as_double((long long)a);
    xs[2 * tid + 1] = __longlong_as_double((long long)b);
  }
  __syncthreads();
}

__device__ __forceinline__ void sweep_fma(const double2 (&t)[16], const double* xs, int g, double& a0, double& a1) {
#pragma unroll
  for (int r = 0; r < 16; r++) { const double xv = xs[16 * g + r]; a0 += t[r].x * xv; a1 += t[r].y * xv; }
}

__global__ __launch_bounds__(kSweepThreads) void k_bwd_sweep(SMat S, int NP, int nt,
                                                             const int32_t* __restrict__ col_off, const int32_t* __restrict__ col_rows,
                                                             const double* __restrict__ Xinv_all, const double* __restrict__ y,
                                                             double* x, double* fail) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* Lv = reinterpret_cast<double*>(smem_raw);   // L(j,j)^-1: blocks (q,q) at q, blocks (p,q), p > q, at 4 + p(p-1)/2 + q; [i][cc]
  double* xs = Lv + 10 * SB * SB;                       // [2][T] the x_i in use / arriving
  double* w = xs + 2 * T;                               // [T]
  double* part = w + T;                                 // [8][T]
  const int tid = threadIdx.x, j = nt - 1 - (int)blockIdx.x;
  const int c2 = 2 * (tid & 63), g = tid >> 6;
  const int32_t* rows = col_rows + col_off[j];
  const int n = col_off[j + 1] - col_off[j];
  double2 ta[16], tb[16];
  if (n > 0) sweep_load(ta, S, rows[0], j, g, c2);
  if (n > 1) sweep_load(tb, S, rows[1], j, g, c2);
  {
    const double* tile = S.tile(j, j);
    const double* Xinv = Xinv_all + (size_t)j * T * T;
    for (int e = tid; e < 10 * 512; e += kSweepThreads) {
      const int blk = e >> 9, u = e & 511, r = u >> 4, cc = 2 * (u & 15);
      double2 v;
      if (blk < 4) v = *reinterpret_cast<const double2*>(Xinv + blk * SB * SB + r * SB + cc);
      else {
        int p = 1, q = blk - 4;
        while (q >= p) { q -= p; p++; }
        v = *reinterpret_cast<const double2*>(tile + (SB * q + r) * T + SB * p + cc);   // Linv(p,q) sits at upper position (q,p)
      }
      *reinterpret_cast<double2*>(Lv + blk * SB * SB + r * SB + cc) = v;
    }
  }
  double a0 = 0.0, a1 = 0.0;
  for (int idx = 0; idx < n; idx += 2) {   // ta holds tile idx, tb tile idx + 1
    sweep_wait(x, NP, rows[idx], xs, fail, tid);
    sweep_fma(ta, xs, g, a0, a1);
    if (idx + 2 < n) sweep_load(ta, S, rows[idx + 2], j, g, c2);
    if (idx + 1 < n) {
      sweep_wait(x, NP, rows[idx + 1], xs + T, fail, tid);
      sweep_fma(tb, xs + T, g, a0, a1);
      if (idx + 3 < n) sweep_load(tb, S, rows[idx + 3], j, g, c2);
    }
  }
  part[g * T + c2] = a0; part[g * T + c2 + 1] = a1;
  __syncthreads();   // (also: Lv is staged)
  if (tid < T) {
    double sum = 0.0;
#pragma unroll
    for (int h = 0; h < 8; h++) sum += part[h * T + tid];
    w[tid] = y[j * T + tid] - sum;
  }
  __syncthreads();
  {
    const int c = tid & (T - 1), hf = tid >> 7, q = c >> 5, cc = c & 31;
    double b0 = 0.0, b1 = 0.0;
    const double* Lq = Lv + q * SB * SB + cc;
#pragma unroll
    for (int i = 8 * hf; i < 8 * hf + 8; i += 2) { b0 += Lq[i * SB] * w[SB * q + i]; b1 += Lq[(i + 1) * SB] * w[SB * q + i + 1]; }
    for (int p = q + 1; p < 4; p++) {
      const double* Lp = Lv + (4 + p * (p - 1) / 2 + q) * SB * SB + cc;
#pragma unroll
      for (int i = 8 * hf; i < 8 * hf + 8; i += 2) { b0 += Lp[i * SB] * w[SB * p + i]; b1 += Lp[(i + 1) * SB] * w[SB * p + i + 1]; }
    }
    part[hf * T + c] = b0 + b1;
  }
  __syncthreads();
  if (tid < T) {
    const double v = (part[tid] + part[T + tid]) + (part[2 * T + tid] + part[3 * T + tid]);
    __hip_atomic_store(x + j * T + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through: what the waiting workgroups poll
    __hip_atomic_store(x + NP + j * T + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // and its shadow copy
  }
}

// gtg_destroy: the handle's schedule streams and events
void destroy_chol_streams(gtg_context& c) {
  CholStreams& cs = c.cs;
  if (cs.panel) (void)hipStreamDestroy(cs.panel);
  if (cs.start) (void)hipEventDestroy(cs.start);
  for (auto* v : {&cs.P, &cs.N}) { for (hipEvent_t e : *v) (void)hipEventDestroy(e); v->clear(); }
  cs = CholStreams();
}

// y = L^-1 g sits in row 0 of the rhs tiles after the factorisation: gathered into a contiguous vector for the backward solve
__global__ __launch_bounds__(T) void k_gather_rhs(SMat S, double* __restrict__ y) { y[blockIdx.x * T + threadIdx.x] = S.tile(S.nt, blockIdx.x)[threadIdx.x]; }

void launch_backward_solve(gtg_context& c, SMat S, int NP, const CholPlan& plan, const double* Xinv, double* x, double* fail) {
  const int nt = NP / T;
  if ((int64_t)c.yred.n != NP) c.yred.alloc(NP);
  double* y = c.yred.p;
  hipLaunchKernelGGL(k_gather_rhs, dim3((unsigned)nt), dim3(T), 0, c.stream, S, y);
  const size_t smem_inv = sizeof(double) * 17 * SB * SB;
  const char* bwd_env = std::getenv("GTG_BWD");   // (read per call: the A/B test switches it inside one process)
  const bool per_row = bwd_env && std::string(bwd_env) == "steps";
  {
    static std::set<int> attr_set;
    static std::mutex attr_mutex;
    std::lock_guard<std::mutex> attr_lock(attr_mutex);
    if (!attr_set.count(c.device)) {
      check_hip(hipFuncSetAttribute((const void*)k_inv_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_inv), "smem attr");
      check_hip(hipFuncSetAttribute((const void*)k_bwd_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSweepSmem), "smem attr");
      attr_set.insert(c.device);
    }
  }
  hipLaunchKernelGGL(k_inv_tiles, dim3((unsigned)nt), dim3(256), smem_inv, c.stream, S, NP, Xinv, per_row ? nullptr : x);
  if (!per_row) {
    hipLaunchKernelGGL(k_bwd_sweep, dim3((unsigned)nt), dim3(kSweepThreads), kSweepSmem, c.stream, S, NP, nt, plan.bwd_col_off.p,
                       plan.bwd_col_rows.p, Xinv, y, x, fail);
  } else {   // GTG_BWD=steps: one launch per block row (the round-1 form, kept as the A/B of the sweep)
    for (int k = nt - 1; k >= 0; k--) {
      const int ncols = (int)plan.bwd_cnt[k];
      hipLaunchKernelGGL(k_bwd_step, dim3(ncols > 0 ? 2 * (unsigned)ncols : 1u), dim3(256), 0, c.stream, S, k,
                         plan.bcols.p + plan.bwd_off[k], ncols, Xinv + (size_t)k * T * T, y, x);
    }
  }
  check_hip(hipGetLastError(), "backward_solve");
}

// gtg_prewarm: this unit's kernels of the default path (the stream schedule's k_panel128 / k_syrk are the fall-back and the A/B: first use pays)
static void prewarm_cholesky(int) {
  prewarm_kernels({(const void*)k_set_epoch, (const void*)k_pack_blocks, (const void*)k_inv_tiles, (const void*)k_bwd_sweep, (const void*)k_gather_rhs});
}
static PrewarmUnit prewarm_cholesky_registered(prewarm_cholesky);

}  // namespace gt
