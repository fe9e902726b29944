// chol_dataflow.hip -- the tile-sparse FP64 Cholesky of the reduced camera / pose system as ONE dataflow pass.
//
// Same mathematics and the same tile storage as cholesky.hip (gtsam::choleskyPartial, base/cholesky.cpp:107-158, on the
// root of the reduced system; rhs carried as an extra tile row = forward solve for free), different schedule:
//
//   * LEFT-LOOKING, one task per stored 128x128 tile.  Task (I, J), I > J:  R = A(I,J) - sum_k L(I,k) L(J,k)^T over the
//     column tiles k < J where both operands are stored, accumulated in the MFMA accumulators over the whole contraction
//     (the tile is read once and written once: no re-read / re-write of C per column pair as in the right-looking
//     schedule), then L(I,J) = R L(J,J)^-T by block substitution in registers, streamed behind the panels of the
//     diagonal tile.  Task PD(J): the same accumulation for the diagonal tile.
//   * TWO PERSISTENT KERNELS per factorisation instead of ~440 launches: k_df_bulk (two workgroups per CU) takes the
//     tile tasks from a ticket counter in a fixed topological order (column by column); k_df_chain (one workgroup on a
//     CU the bulk stream's CU mask leaves free) factors the diagonal tiles one after the other with the streamed
//     32-column panel of chol_device.h.  Dependencies are epoch-stamped flags in HBM (release / acquire at agent
//     scope): a task only ever waits for tasks that precede it in the ticket order, and tickets are taken in order
//     by running workgroups, so the earliest unfinished task always runs -- no deadlock whatever the dispatch order.
//   * The serial chain per block column is: diagonal tile (k_df_chain) -> last substitution step of tile (J+1, J) ->
//     its contribution to the next diagonal tile, applied by k_df_chain itself in 32-column slices as the substitution
//     publishes them (PD(J+1) leaves that one contraction step out) -> next diagonal tile; everything else runs beside it.
//
// All sums run in a fixed order (the k list of a task): results are bit-reproducible and independent of which workgroup
// ran which task.  A dependency wait that runs into its bound raises fail[1] (reported as an error by the C ABI, never
// as "not positive definite"), after which every wait falls through and both kernels drain.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "chol_device.h"

namespace gt {

namespace {

constexpr int KC = 32;            // contraction chunk staged per LDS-DMA round (columns): one 32-column block of the operand tiles
constexpr int ROWB = KC * 8;      // bytes per LDS row of a chunk
constexpr int CH = T * KC * 8;    // bytes of one 128-row panel chunk
constexpr int PX = SB + 2;        // LDS pitch (doubles) of a 128 x 32 block of the substitution
// A dependency wait gives up after kWaitTicks of the 100 MHz constant clock (20 ms: four factorisations of the headline problem; the
// try is then repeated with the other schedule, api.hip) -- the bound used to be 4 M polls, more than a second per stuck wait.
constexpr long long kWaitTicks = 2000000;
constexpr long long kPieceBase = 4096;   // part_flag = kPieceBase epoch + finished pieces of the tile's contraction (< kPieceBase pieces per tile)
constexpr int kChainWords = 4;     // words per diagonal tile in the chain kernel's table: slot of (J, J), slot of (J, J-1) or -1, the 64-bit sub-tile mask of (J, J-1)
constexpr int kStepWords = 6;      // words of a contraction step in the device klist: slots of (I, k), (J, k), their two 64-bit sub-tile masks
constexpr int kImgDoubles = 2 * 64 * 8;   // one MFMA operand image of a 32x32 block (chol_device.h opnd_off): 8 KB
constexpr size_t kSmemBulk = std::max<size_t>(4 * (size_t)CH, sizeof(double) * (T * PX + 4 * kImgDoubles));
constexpr size_t kSmemPotrf = sizeof(double) * kPotrfSmemDoubles;
constexpr size_t kSmemChain = (kSmemPotrf + 15) / 16 * 16 + sizeof(double) * 4 * SB * PB;   // + one 128 x 32 slice of the tile below
// flag values: epoch * 8 + steps; a tile (I, J) is final at 4 steps (its four 32-column blocks), a diagonal tile's word in
// its Dinv slot counts released panels, pd_flag is epoch * 8 + 4 when the accumulated diagonal tile is in
__device__ __forceinline__ long long final_of(long long epoch) { return epoch * 8 + 4; }

#ifndef GT_KERNEL_EMU
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#endif
constexpr int kHandoffAux = 16;   // cache-policy bits of the LDS-DMA that reads handed-over data: sc1 (agent scope, past the CU's L1)

// ---- cross-workgroup visibility without cache-wide fences -----------------------------------------------------------
// The two kernels exchange tiles through HBM while they run, between XCDs whose L2s are not coherent with each other.  The
// textbook protocol (release fence = write back the whole L2, acquire fence = invalidate it) costs microseconds per use with
// 60 workgroups per XCD producing dirty lines (first version: 45-80 us per tile in the substitution, all of it fences).
// Instead (GTG_DF_FENCES=0, the default):
//   * every datum another workgroup will read is stored WRITE-THROUGH (sc1: agent-scope atomic store, relaxed), so it is
//     in memory when the store is acknowledged; the producer waits for its own stores (s_waitcnt vmcnt(0)), a workgroup
//     barrier collects the wavefronts, then one lane stores the flag (write-through as well);
//   * a consumer polls the flag with agent-scope (sc1) loads and then reads the data with agent-scope (sc1) loads as well: the
//     LDS-DMA of the operand panels and of the operand images carries the sc1 bit (kHandoffAux), the register loads of the chain
//     kernel and of the partial tiles are relaxed agent-scope atomic loads.  sc1 stores + drained flag on the producer side and
//     sc1 loads on the consumer side is one of the two valid forms of the hand-off on this target (MI355X_MICROARCH.md,
//     "Workgroup dispatch, XCD placement & inter-workgroup visibility"; cdna_hip_programming.md Guideline 16 R1); the other one --
//     one agent-scope acquire after a matched flag, then plain loads -- was built and measured as an A/B in round 4 (bit-identical).
// Rounds 1-3 read the data with PLAIN loads and no acquire ("nobody can hold a stale copy of a tile that is written once"): an
// argument about this schedule, not about the memory model, and the verdict of round 3 was right to reject it.
// -DGTG_DF_FENCES=1 builds the textbook release / acquire protocol for comparison.
#ifndef GTG_DF_FENCES
#define GTG_DF_FENCES 0
#endif
__device__ __forceinline__ void st_wt(double* p, double v) {
#if GTG_DF_FENCES
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void stores_done() {   // this wavefront's stores are acknowledged (resp. released)
#if GTG_DF_FENCES
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
  GT_DRAIN_STORES();
#endif
}
// Flags and stale lines (rounds 3 - 6).  In a process with several handles on one device a polled line is occasionally left STUCK in the
// poller's XCD L2 with the value it had when the polling began -- sc1 loads (served by that L2) return the old flag for seconds while memory
// holds the new one and every other XCD sees it (tools/df_contention_diag.py, profiles/r03_df_contention.txt).  Rounds 3 - 5 therefore
// published every flag twice (a shadow word in another page, looked at after 1024 fruitless polls) and, since round 4, also asked the
// device's point of coherence with a read-modify-write atomic after 512 (wait_flags: fetch_max with 0; agent-scope RMW atomics execute
// at the point of coherence -- the ticket counter of the bulk kernel relies on exactly that across the 8 XCDs).  Round 6 counted what
// ends those long waits (profiles/r06e_poll_statistics.txt; 3.5e8 waits of >= 256 polls in three 60 s stress runs): of 1.43 M waits that
// ended on the RMW poll NONE found its word still old with the next sc1 load -- they are flags that changed in the microsecond between the
// iteration's sc1 poll and the RMW (0.4 % of the long waits, the ratio of that window to a wait), not stale lines, and the RMW leaves
// the line current --; of 0.21 M waits that ended on the shadow word 1 397 DID still read the old value from the word itself (all but one
// with three handles in flight): that is the stuck line, rare (1.6e-5 of the long waits) and real, and the shadow read ends the wait
// without curing it.  So the shadow words are gone: the flag is ONE word published by an exchange, a long wait asks the point of coherence
// every 512 polls.
__device__ __forceinline__ void st_flag(long long* p, long long v) {
#if GTG_DF_FENCES
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
  (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// the word as the point of coherence holds it (flags only grow: max with 0 leaves them alone); wave-uniform address: one lane asks
__device__ __forceinline__ long long ld_flag_rmw(const long long* p) {
  long long v = 0;
  if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0)
    v = __hip_atomic_fetch_max(const_cast<long long*>(p), 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int lo = __builtin_amdgcn_readfirstlane((int)v), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ void acquired() {
#if GTG_DF_FENCES
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
  asm volatile("" ::: "memory");
#endif
}
// (polling EVERY flag with a returning read-modify-write atomic instead was measured in round 5: + 1.5 % on the L1723 factorisation --
// a slower poll than an sc1 load; profiles/r05a_variants_ab.txt.  The RMW poll stays what a wait falls back to after 512 misses.)
__device__ __forceinline__ long long ld_flag(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool timed_out(const double* fail) {
  return __hip_atomic_load(reinterpret_cast<const long long*>(fail + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
// Every lane of the calling wavefront polls the same words (one broadcast load); bounded: see the file comment.
// The first wait that gives up leaves a record (what it waited for) in dbg[0..7] (gtg_debug_df_ctrl).
__device__ __forceinline__ void wait_flags(const long long* f1, long long v1, const long long* f2, long long v2, double* fail,
                                           int32_t* dbg = nullptr, int kind = 0, int a = 0, int b = 0, int c = 0) {
  int spins = 0;
  long long t0 = 0;
  // (two polls in flight, half a round trip apart, were measured in round 6: 4.63 -> 4.71 ms -- the polling traffic costs more than the
  // earlier notice saves; profiles/r06z_poll_depth.txt)
  while (ld_flag(f1) < v1 || ld_flag(f2) < v2) {
    __builtin_amdgcn_s_sleep(4);
    if ((++spins & 255) == 0) {
      if (timed_out(fail)) break;
      if (spins == 256) { t0 = wall_clock64(); if (dbg && (threadIdx.x & 63) == 0) atomicAdd(dbg + 10, 1); }   // ctrl[18]: waits that got this long
      if ((spins & 511) == 0 && ld_flag_rmw(f1) >= v1 && ld_flag_rmw(f2) >= v2) {    // ask the point of coherence (see the comment above st_flag)
        // ctrl[7]: waits that ended here; ctrl[16]: ... and whose word the next sc1 load still finds old (0 of 1.43 M in round 6's count)
        if (dbg && (threadIdx.x & 63) == 0) { atomicAdd(dbg - 1, 1); if (ld_flag(f1) < v1 || ld_flag(f2) < v2) atomicAdd(dbg + 8, 1); }
        break;
      }
      if (wall_clock64() - t0 > kWaitTicks) {
        if (dbg && atomicCAS(dbg, 0, kind) == 0) {
          dbg[1] = a; dbg[2] = b; dbg[3] = c; dbg[4] = (int)ld_flag(f1); dbg[5] = (int)ld_flag(f2); dbg[6] = (int)v1; dbg[7] = (int)v2;
          // post-mortem (ctrl[2..3]): where the waiter runs
          unsigned xcc, hw;
          GT_XCC_ID(xcc);
          GT_HW_ID(hw);
          dbg[-6] = (int)(xcc & 0xf); dbg[-5] = (int)hw;
        }
        // (write-through: the other XCDs' waiters poll this word and must see it while the kernels are still running)
        __hip_atomic_store(fail + 1, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
      }
    }
  }
  acquired();
}

// Published 32-column blocks (0..4) of the two operand tiles of a contraction step: the smaller of their two flags relative
// to this factorisation's base (flags of earlier factorisations count as 0).  Wave-uniform.
__device__ __forceinline__ int tile_progress(const long long* f1, const long long* f2, long long flagbase) {
  const long long a = ld_flag(f1) - flagbase, b = ld_flag(f2) - flagbase;
  const long long m = a < b ? a : b;
  return __builtin_amdgcn_readfirstlane((int)(m < 0 ? 0 : m));
}
__device__ __forceinline__ int wait_progress(const long long* f1, const long long* f2, long long flagbase, int need, double* fail,
                                             int32_t* dbg, int kind, int a, int b, int c) {
  wait_flags(f1, flagbase + need, f2, flagbase + need, fail, dbg, kind, a, b, c);
  return need;   // (at least; the caller asks again when it needs more)
}

// ---- one bulk task ------------------------------------------------------------------------------------------------
// ONE workgroup of 16 wavefronts per CU (the four waves per SIMD the FP64 matrix pipe needs, all from the same task).  Two
// 8-wave workgroups per CU gave the same contraction rate, but the END of a task -- the substitution, which is a link of a
// serial chain -- then shared the matrix pipe with a neighbour in the middle of its contraction: 12 us from the last panel of the
// diagonal tile to the finished tile, against 2.5 us with the CU to itself (measured with the single-kernel form).
// Wavefront (rt, h) = (bulk_rt(wave), bulk_h(wave)) owns rows 16 rt .. 16 rt + 15 and the column half h (64 columns = the 32-column
// blocks 2 h and 2 h + 1): 1 x 4 MFMA tiles, 32 accumulator registers -- the substitution runs in the same layout (a row of
// L(I,J) depends on the same row of R only), every register index is a compile-time constant, nothing spills.
// Panel chunks (128 rows x 16 columns of L(I,k) and of L(J,k)) go L2/HBM -> LDS by LDS-DMA, double buffered, 16-byte slots
// XOR-swizzled on the source address and on the operand reads.
constexpr int kBulkThreads = 1024;
// wavefront -> (row tile, column half).  Wavefronts go to the four SIMDs of the CU with period four.  Two things are wanted of the
// wavefronts that share a SIMD: (i) both column halves -- the substitution's later steps occupy ONE half only (blocks 2, 3: half 1); with
// h = wave & 1 (rounds 2 - 5) those eight wavefronts shared two SIMDs and the other two matrix pipes idled, 2.8 us for the 32 MFMAs of
// step 2 (profiles/r06t_substitution_steps.txt); (ii) row tiles that lie apart -- the structurally empty sub-tiles of an operand tile are
// a staircase (the rows near the band's edge start further right), so the row tiles with work in a 16-column strip are a prefix 0 .. n-1,
// and a SIMD that holds two neighbouring row tiles in both halves is fully busy as soon as n > 4.  SIMD s holds the row tiles
// s & 1, 2 + (s & 1), 4 + (s & 1), 6 + (s & 1) with alternating halves.
__device__ __forceinline__ int bulk_rt(int wave) { return 2 * (wave >> 2) + (wave & 1); }
__device__ __forceinline__ int bulk_h(int wave) { return ((wave >> 2) + (wave >> 1)) & 1; }

template <int H>
__device__ __forceinline__ void substitute(char* smem_raw, v4f64 (&x)[4], double* __restrict__ C, int I, int J,
                                           long long* __restrict__ myflag, const long long* __restrict__ pflag, double* __restrict__ Xinv_all,
                                           double* __restrict__ fail, long long epoch, int32_t* __restrict__ dbg,
                                           long long* __restrict__ tr, bool row_live) {
  // (row_live: the 16 rows of this wavefront can be non-zero in tile (I, J) -- its sub-tile mask; a row tile without a live sub-tile stays
  // exactly zero through the substitution and its wavefronts leave the MFMAs out: half of the row tiles of a pose graph's tiles)
  // ---- L(I,J) = R L(J,J)^-T: right-looking block substitution over the 32-column blocks q.
  // Step q runs when k_df_chain has released panel q of the diagonal tile (inverse of its diagonal block; the operand images
  // L(p, q-1), p >= q, are out by then as well):
  //   R_p -= X_{q-1} L(p, q-1)^T for the blocks p >= q,  then  X_q = R_q Linv(q,q)^T  (stored, and published: flag = 8 epoch + q + 1)
  // Block p = 2 h + l lives in x[2 l + t] of the wavefronts of column half h.  The A operand of a step is a 16 x 32 patch of LDS
  // per row tile (W[rt]: -X_{q-1}; then R_q -> operand layout; then -X_q), shared by the two wavefronts of the row tile; the B
  // operand images (<= 3 blocks L(p, q-1) + Linv(q,q)) come memory -> LDS by LDS-DMA, one memory latency per step.
  int tid_ = threadIdx.x;
  GT_PIN(tid_);
  const int tid = tid_, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = bulk_rt(wave);
  constexpr int h = H;
  const int lr = lane & 15, lk = lane >> 4;
  const long long flagbase = epoch * 8;
  const unsigned lane_off = (unsigned)(lk * T + lr);
  double* Crow = C + (16 * rt) * T + 64 * h;
  const double* Xinv = Xinv_all + (size_t)J * T * T;
  // pflag: the diagonal tile's progress word (released panels) = the flag word of tile (J, J); myflag: the word of this tile
  double* W = reinterpret_cast<double*>(smem_raw) + rt * (16 * PX);
  double* img = reinterpret_cast<double*>(smem_raw) + 8 * 16 * PX;
  long long pf = ld_flag(pflag);   // released panels of the diagonal tile, as last seen (monotonic)
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (pf < flagbase + q + 1) {
      wait_flags(pflag, flagbase + q + 1, pflag, flagbase + q + 1, fail, dbg, 3, I, J, q);
      pf = flagbase + q + 1;
    }
    if (tr && tid == 0) tr[4 + q] = wall_clock64();   // panel q seen
    // (the previous step's images have been consumed: barrier at the end of the loop body)
    {  // images of this step: slots 0 .. 3-q = L(q + s, q - 1) (q > 0), slot 3 = Linv(q,q); wavefront w moves 1 KiB pieces
      // (w & 7) of the slots (w >> 3) and (w >> 3) + 2
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int sl = (wave >> 3) + 2 * j;           // uniform
        const int p = q + sl;
        const int blk = (sl == 3) ? 6 + q : p * (p - 1) / 2 + (q - 1);
        if ((sl == 3) || (q > 0 && p < 4))
          __builtin_amdgcn_global_load_lds((gptr_t)(Xinv + kOpndBase + (size_t)blk * kImgDoubles + 128 * (wave & 7) + 2 * lane),
                                           (lptr_t)(img + sl * kImgDoubles + 128 * (wave & 7)), 16, 0, kHandoffAux);
      }
    }
    // The image fetch and the acknowledgement of X_{q-1}'s write-through stores are two memory round trips; rounds 3 - 5 took them one after
    // the other (drain, barrier, flag, fetch, barrier), now the fetch is in flight while the stores drain (s_waitcnt vmcnt(0) covers both)
    if (q > 0) stores_done();   // X_{q-1} of this wavefront is in memory
    __syncthreads();   // (drains the DMA); -X_{q-1} is complete in the W patches and out of every wavefront
    if (q > 0 && tid == 0) st_flag(myflag, flagbase + q);
    if (q > 0) {
      double a[8];
#pragma unroll
      for (int s = 0; s < 8; s++) a[s] = W[lr * PX + 8 * lk + s];
#pragma unroll
      for (int l = 0; l < 2; l++) {
        constexpr int dummy_ = 0; (void)dummy_;
        const int p = 2 * h + l;
        if (p < q) continue;   // (compile-time)
#pragma unroll
        for (int t = 0; t < 2; t++) {
          double bl[8];
          const double2* op = reinterpret_cast<const double2*>(img + (p - q) * kImgDoubles + (t * 64 + lane) * 8);
#pragma unroll
          for (int u = 0; u < 4; u++) { const double2 v = op[u]; bl[2 * u] = v.x; bl[2 * u + 1] = v.y; }
          if (row_live)
#pragma unroll
            for (int s = 0; s < 8; s++) x[2 * l + t] = MFMA(a[s], bl[s], x[2 * l + t]);
        }
      }
    }
    __syncthreads();   // both wavefronts of a row tile have read -X_{q-1}: the patch is free for R_q
    // the next panel's flag, requested a step's arithmetic ahead of its use: a poll is a memory round trip even when the panel is long out.
    // (In front of this step's write-through stores: the memory counter is in order, a load issued behind them returns with their acknowledgements)
    if (q < 3 && pf < flagbase + q + 2) pf = ld_flag(pflag);
#pragma unroll
    for (int l = 0; l < 2; l++) {
      if (2 * h + l != q) continue;   // (compile-time)
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) W[(lk + 4 * r) * PX + 16 * t + lr] = x[2 * l + t][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-local: written and read by this wavefront
      double a[8];
#pragma unroll
      for (int s = 0; s < 8; s++) a[s] = W[lr * PX + 8 * lk + s];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        double bi[8];
        const double2* op = reinterpret_cast<const double2*>(img + 3 * kImgDoubles + (t * 64 + lane) * 8);
#pragma unroll
        for (int u = 0; u < 4; u++) { const double2 v = op[u]; bi[2 * u] = v.x; bi[2 * u + 1] = v.y; }
        x[2 * l + t] = (v4f64){0.0, 0.0, 0.0, 0.0};
        if (row_live)
#pragma unroll
          for (int s = 0; s < 8; s++) x[2 * l + t] = MFMA(a[s], bi[s], x[2 * l + t]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // operand reads of R_q before -X_q overwrites the patch
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double v = x[2 * l + t][r];
          W[(lk + 4 * r) * PX + 16 * t + lr] = -v;   // negated: the A operand of the next step's updates
          st_wt((Crow + (4 * r) * T + 32 * l + 16 * t) + lane_off, v);
        }
    }
    if (q < 3) __syncthreads();   // this step's images have been consumed by every wavefront: the next step may fetch over them
  }
  stores_done();
  __syncthreads();
  if (tid == 0) st_flag(myflag, flagbase + 4);
}


__device__ __forceinline__ void run_task(char* smem_raw, double* __restrict__ S, int I, int J, int slotC, int slotD, int G, int scratch,
                                         const int32_t* __restrict__ kl, int kcnt, int piece, int pieces, unsigned long long out_mask,
                                         long long* __restrict__ tile_flag, long long* __restrict__ part_flag,
                                         long long* __restrict__ pd_flag, double* __restrict__ Xinv_all,
                                         double* __restrict__ fail, long long epoch, int32_t* __restrict__ dbg, long long* __restrict__ tr) {
  // the thread index is laundered per task: everything derived from it is recomputed here instead of being hoisted out of
  // the persistent task loop and kept alive across it
  int tid_ = threadIdx.x;
  GT_PIN(tid_);
  const int tid = tid_, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = bulk_rt(wave), h = bulk_h(wave);
  const int lr = lane & 15, lk = lane >> 4;
  // slotC: the slot of tile (I, J), slotD: of the diagonal tile (J, J); a contraction step is the pair of slots of its two operand
  // tiles (I, k), (J, k) -- tiles and their flag words are both indexed by slot
  double* C = S + (int64_t)slotC * TT;
  const long long fin = final_of(epoch);
  const long long flagbase = epoch * 8;

  // One DMA instruction = 1 KiB = 4 rows x 16 slots of 16 bytes (a row of a chunk is 256 bytes): lane l -> row + l / 16, stored
  // slot l % 16, which holds the logical slot (l % 16) ^ (row & 15).  With rows 256 bytes apart every row starts at bank 0, and
  // the 16 operand rows of an MFMA read (rows = 16 x + lr) then hit 16 different slots: conflict-free ds_read_b64.
  const int drow = lane >> 4, dslot = lane & 15;
  auto stage = [&](const double* Ap, const double* Bp, int ch, int buf) {   // wavefront w moves rows 8 w .. 8 w + 7 of both panels
    char* base = smem_raw + buf * 2 * CH;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int row = 8 * wave + 4 * q + drow;
      const int logical = dslot ^ (row & 15);
      const double* ga = Ap + row * T + ch * KC + 2 * logical;
      const double* gb = Bp + row * T + ch * KC + 2 * logical;
      __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)(base + (2 * wave + q) * 1024), 16, 0, kHandoffAux);
      __builtin_amdgcn_global_load_lds((gptr_t)gb, (lptr_t)(base + CH + (2 * wave + q) * 1024), 16, 0, kHandoffAux);
    }
  };

  // element (c, r) of this lane: row 16 rt + lk + 4 r, column 64 h + 16 c + lr -- uniform part + one 32-bit lane offset
  const unsigned lane_off = (unsigned)(lk * T + lr);
  double* Crow = C + (16 * rt) * T + 64 * h;
  v4f64 x[4];
  long long waited = 0;   // GTG_DF_TRACE: ticks this workgroup spent in the flag waits of the accumulation / contraction phase
#define GT_TIMED(call) do { if (tr) { const long long w0_ = wall_clock64(); call; waited += wall_clock64() - w0_; } else { call; } } while (0)
  // ---- where this piece's partial result lives.  The early pieces of a tile accumulate IN PLACE, one after the other (lane 0).  A tile
  // whose contraction list is very long -- the diagonal tiles of a nested-dissection separator collect an update from every block
  // column of the subtrees below them: 300 steps for the root of the 20 000-pose graph, all of them waiting for ONE chain of
  // pieces while eight leaf chains produce their operands side by side -- has G accumulator LANES instead: early piece r belongs
  // to lane r mod G, the lanes 1 .. G-1 accumulate from zero into scratch slots behind the stored tiles, and the final piece adds
  // the lanes up in lane order (a fixed order: bit-reproducible).  G = 1 (every tile of the camera systems) is the in-place chain.
  const int E = pieces - 1;                    // early pieces of the tile
  const bool early = piece < E;
  const int lane_g = (G > 1 && early) ? piece % G : 0, q = (G > 1 && early) ? piece / G : piece;
  const int acc_slot = lane_g == 0 ? slotC : scratch + lane_g - 1;
  double* AccRow = S + (int64_t)acc_slot * TT + (16 * rt) * T + 64 * h;
  long long* pflag_mine = part_flag + acc_slot;
  auto load_acc = [&](const double* row, bool add) {     // a partial result written (write-through) by another workgroup: read past the L1 (sc1)
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const double v = __hip_atomic_load((row + (4 * r) * T + 16 * c) + lane_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x[c][r] = add ? x[c][r] + v : v;
      }
  };
#pragma unroll
  for (int c = 0; c < 4; c++) x[c] = (v4f64){0.0, 0.0, 0.0, 0.0};
  if (early ? q > 0 : E > 0) {
    // lane 0 / this lane: the pieces before this one
    const long long have = early ? q : (E + G - 1) / G;
    GT_TIMED(wait_flags(pflag_mine, epoch * kPieceBase + have, pflag_mine, epoch * kPieceBase + have, fail, dbg, 7, I, J, piece));
    load_acc(AccRow, false);
    if (!early)
      for (int g = 1; g < G; g++) {            // the other lanes, in lane order
        const long long ng = (E - g + G - 1) / G;
        const long long* fg = part_flag + scratch + g - 1;
        GT_TIMED(wait_flags(fg, epoch * kPieceBase + ng, fg, epoch * kPieceBase + ng, fail, dbg, 7, I, J, piece));
        load_acc(S + (int64_t)(scratch + g - 1) * TT + (16 * rt) * T + 64 * h, true);
      }
  } else if (lane_g == 0) {
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int r = 0; r < 4; r++) x[c][r] = (Crow + (4 * r) * T + 16 * c)[lane_off];
  }

  if (kcnt > 0) {
    // A contraction step consumes its two operand tiles 16 columns at a time, and a tile is published in 32-column blocks
    // while its substitution runs: a chunk only waits for the block it reads.  For every step but the most recent ones both
    // tiles are long final and the test is a scalar compare; for the last step (block column J-1, whose tiles become final
    // behind the diagonal tile that is being factored right now) the contraction streams behind the substitution.
    int ka = kl[0], kb = kl[1];
    int cp = 0;
    GT_TIMED(cp = wait_progress(tile_flag + ka, tile_flag + kb, flagbase, 1, fail, dbg, 1, I, J, ka));   // progress known for the current step
    const double* Ak = S + (int64_t)ka * TT;
    const double* Bk = S + (int64_t)kb * TT;
    stage(Ak, Bk, 0, 0);
    __syncthreads();
    const int a_row_off = (16 * rt + lr) * ROWB, b_row_off = (64 * h + lr) * ROWB;
    // Structural zeros inside the operand tiles (round 6).  A stored 128 x 128 tile is rarely full: at the granularity of MFMA tiles the
    // contraction steps of the L1723 shape carry 21 % structurally-zero products (tools/df_subtile_count.py).  Every step brings the
    // 8 x 8-bit masks of its two operand tiles (strip-level symbolic factorisation, analysis.hip); per 16-column strip a wavefront
    // multiplies only if ITS 16 rows of L(I, k) can be non-zero there, and only into the column tiles t whose rows of L(J, k) can: `live`
    // holds those four bits for each of the eight strips.  A skipped product is a sum of exact zeros: the factor is unchanged.  (The
    // right-hand-side row, tile row nt, is the extreme case: one live row tile -- fourteen wavefronts multiplied zeros, 6 % of the bulk
    // kernel's resident time, tools/df_wait_analysis.py.)
    auto live_bits = [&](const int32_t* w) {
      const unsigned long long ma = (unsigned)w[2] | ((unsigned long long)(unsigned)w[3] << 32), mb = (unsigned)w[4] | ((unsigned long long)(unsigned)w[5] << 32);
      const unsigned arow = (unsigned)(ma >> (8 * rt)) & 0xFFu;
      unsigned lv = 0;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const unsigned bcol = (unsigned)(mb >> (8 * (4 * h + t))) & 0xFFu & arow;   // strips where column tile t of this wavefront has work
#pragma unroll
        for (int sp = 0; sp < 8; sp++) lv |= ((bcol >> sp) & 1u) << (4 * sp + t);
      }
      return __builtin_amdgcn_readfirstlane(lv);
    };
    unsigned live = live_bits(kl);
    const int half = (lk & 1) * 8, hi = lk >> 1, sw = lr;
    for (int ki = 0; ki < kcnt; ki++) {
      // the flags of the next contraction step, fetched a whole step ahead of their use
      const int32_t* kn = kl + kStepWords * (ki + 1 < kcnt ? ki + 1 : ki);
      const int kna = kn[0], knb = kn[1];
      const unsigned live_next = live_bits(kn);
      int np = tile_progress(tile_flag + kna, tile_flag + knb, flagbase);
      const double* An = S + (int64_t)kna * TT;
      const double* Bn = S + (int64_t)knb * TT;
#pragma unroll
      for (int ch = 0; ch < T / KC; ch++) {
        const int cur = ch & 1;
        if (ch + 1 < T / KC) {
          const int need = ch + 2;   // chunk ch + 1 = the operand tiles' 32-column block ch + 1
          if (cp < need) { cp = tile_progress(tile_flag + ka, tile_flag + kb, flagbase); if (cp < need) GT_TIMED(cp = wait_progress(tile_flag + ka, tile_flag + kb, flagbase, need, fail, dbg, 2, I, J, ka)); }
          stage(Ak, Bk, ch + 1, cur ^ 1);
        } else if (ki + 1 < kcnt) {
          if (np < 1) GT_TIMED(np = wait_progress(tile_flag + kna, tile_flag + knb, flagbase, 1, fail, dbg, 2, I, J, kna));
          stage(An, Bn, 0, cur ^ 1);
        }
        const char* Ac = smem_raw + cur * 2 * CH;
        const char* Bc = Ac + CH;
#pragma unroll
        for (int sp = 0; sp < KC / kSub; sp++) {   // the chunk's two 16-column strips
          const unsigned m = (live >> (4 * (ch * (KC / kSub) + sp))) & 15u;
          if (m == 15u) {
#pragma unroll
            for (int kk = kSub * sp; kk < kSub * (sp + 1); kk += 4) {
              const int so = (((kk >> 1) + hi) ^ sw) * 16 + half;
              const double a = -lds_ld(reinterpret_cast<const double*>(Ac + a_row_off + so));   // (single 8-byte reads: chol_device.h::lds_ld)
              double b[4];
#pragma unroll
              for (int t = 0; t < 4; t++) b[t] = lds_ld(reinterpret_cast<const double*>(Bc + b_row_off + t * 16 * ROWB + so));
#pragma unroll
              for (int t = 0; t < 4; t++) x[t] = MFMA(a, b[t], x[t]);
            }
          } else if (m != 0u) {
#pragma unroll
            for (int kk = kSub * sp; kk < kSub * (sp + 1); kk += 4) {
              const int so = (((kk >> 1) + hi) ^ sw) * 16 + half;
              const double a = -lds_ld(reinterpret_cast<const double*>(Ac + a_row_off + so));
#pragma unroll
              for (int t = 0; t < 4; t++)
                if (m & (1u << t)) x[t] = MFMA(a, lds_ld(reinterpret_cast<const double*>(Bc + b_row_off + t * 16 * ROWB + so)), x[t]);
            }
          }
        }
        __syncthreads();   // drains the DMA of the next chunk (vmcnt) and fences the buffer just read
      }
      ka = kna; kb = knb; Ak = An; Bk = Bn; cp = np; live = live_next;
    }
  }
  if (tr && tid == 0) { tr[1] = wall_clock64(); tr[3] |= waited << 40; }   // (bits 40..: ticks waited; below: where the workgroup runs)
#undef GT_TIMED

  if (early) {   // an early piece: the partial result goes back into its lane's tile
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int r = 0; r < 4; r++) st_wt((AccRow + (4 * r) * T + 16 * c) + lane_off, x[c][r]);
    stores_done();
    __syncthreads();
    if (tid == 0) st_flag(pflag_mine, epoch * kPieceBase + q + 1);
    return;
  }
  if (I == J) {
    // PD(J): the diagonal tile with its updates in (all but block column J-1's, which k_df_chain applies itself)
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int r = 0; r < 4; r++) st_wt((Crow + (4 * r) * T + 16 * c) + lane_off, x[c][r]);
    stores_done();
    __syncthreads();
    if (tid == 0) st_flag(pd_flag + J, fin);
    return;
  }

  // the substitution, specialised for the column half (the wavefront-uniform branch keeps every "is block p mine / still open"
  // test a compile-time constant: with run-time tests the compiler merged the accumulators through scratch at every step)
  const bool row_live = ((out_mask >> (8 * rt)) & 0xFFull) != 0;
  if (h == 0) substitute<0>(smem_raw, x, C, I, J, tile_flag + slotC, tile_flag + slotD, Xinv_all, fail, epoch, dbg, tr, row_live);
  else substitute<1>(smem_raw, x, C, I, J, tile_flag + slotC, tile_flag + slotD, Xinv_all, fail, epoch, dbg, tr, row_live);
}

__device__ __forceinline__ void bulk_loop(char* smem_raw, double* __restrict__ S, const int32_t* __restrict__ tasks,
                                          int ntasks, const int32_t* __restrict__ klist,
                                          long long* __restrict__ tile_flag, long long* __restrict__ part_flag,
                                          long long* __restrict__ pd_flag,
                                          double* __restrict__ Xinv_all, int32_t* __restrict__ ctrl,
                                          double* __restrict__ fail, const long long epoch,
                                          long long* __restrict__ trace) {
  __shared__ int s_task;
  for (;;) {
    if (threadIdx.x == 0) s_task = atomicAdd(ctrl, 1);
    __syncthreads();
    const int t = s_task;
    __syncthreads();
    if (t >= ntasks) return;
    const int32_t* d = tasks + 12 * (int64_t)t;   // I, J, offset / count of the step list, piece r of R, slot of (I, J), slot of (J, J), accumulator lanes, first scratch slot, [10, 11]: the tile's own sub-tile mask
    long long* tr = trace ? trace + 8 * (int64_t)t : nullptr;   // GTG_DF_TRACE: 100 MHz stamps (taken, contraction done, done), place
    if (tr && threadIdx.x == 0) {
      unsigned hw, xcc;
      GT_HW_ID(hw);
      GT_XCC_ID(xcc);
      tr[0] = wall_clock64(); tr[3] = ((long long)(xcc & 0xf) << 32) | hw;
    }
    run_task(smem_raw, S, d[0], d[1], d[6], d[7], d[8], d[9], klist + kStepWords * (int64_t)d[2], d[3], d[4], d[5], (unsigned)d[10] | ((unsigned long long)(unsigned)d[11] << 32), tile_flag, part_flag, pd_flag, Xinv_all, fail, epoch, ctrl + 8, tr);
    __syncthreads();   // the substitution buffers / staging buffers are reused by the next task
    if (tr && threadIdx.x == 0) tr[2] = wall_clock64();
  }
}

#ifndef GT_KERNEL_EMU   // (the emulation calls bulk_loop / chain_loop itself: dynamic LDS is a host buffer there)
__global__ __launch_bounds__(kBulkThreads) void k_df_bulk(double* __restrict__ S, const int32_t* __restrict__ tasks,
                                                    int ntasks, const int32_t* __restrict__ klist,
                                                    long long* __restrict__ tile_flag, long long* __restrict__ part_flag,
                                                    long long* __restrict__ pd_flag,
                                                    double* __restrict__ Xinv_all, int32_t* __restrict__ ctrl,
                                                    double* __restrict__ fail, const long long epoch,
                                                    long long* __restrict__ trace) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bulk_loop(smem_raw, S, tasks, ntasks, klist, tile_flag, part_flag, pd_flag, Xinv_all, ctrl, fail, epoch, trace);
}
#endif

// The diagonal tiles.  Two workgroups (even / odd J) take turns, so that everything before the last slice of tile J -- waiting for
// PD(J), bringing the tile into LDS, the first three slices -- happens while the partner factors tile J-1 (with one workgroup
// those 13 us per tile were on the serial chain).  Tile J: wait until PD(J) is in (every update but the one of block column J-1),
// bring it into LDS, apply  C -= L(J,J-1) L(J,J-1)^T  in four 32-column slices as the substitution of tile (J, J-1) publishes
// them (the last slice is the only one left when that tile is final), factor (potrf_body releases its four panels to the
// substitution steps of the tiles below through the tile's progress word).
__device__ __forceinline__ void chain_loop(char* smem_raw, double* __restrict__ S, double* __restrict__ Xinv_all,
                                           const long long* __restrict__ pd_flag, long long* tile_flag,
                                           const int32_t* __restrict__ chain_slots, double* __restrict__ fail,
                                           const long long epoch, int32_t* __restrict__ ctrl,
                                           long long* __restrict__ trace, const int32_t* __restrict__ my_tiles, int n_mine,
                                           const unsigned char* __restrict__ pivot_kind, double* __restrict__ tile_exp,
                                           long long* __restrict__ stamps = nullptr) {   // GTG_DF_TRACE: 64 stamps per diagonal tile (potrf_body's STAMPs + per-wavefront stage ends)
  double* A = reinterpret_cast<double*>(smem_raw);
  double* X = reinterpret_cast<double*>(smem_raw + (kSmemPotrf + 15) / 16 * 16);   // [4][SB][PB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  for (int it = 0; it < n_mine; it++) {
    const int J = my_tiles[it];
    if (tid < 64) wait_flags(pd_flag + J, final_of(epoch), pd_flag + J, final_of(epoch), fail, ctrl + 8, 4, J, J, 0);
    __syncthreads();
    acquired();
    if (tid == 0) { atomicAdd(ctrl + 1, 1); if (trace) trace[2 * J] = wall_clock64(); }   // debug: diagonal tiles started (all chains)
    const int dslot = chain_slots[kChainWords * J];   // slot of (J, J); [+ 1]: of (J, J-1) (-1: not stored); [+ 2, + 3]: that tile's sub-tile mask
    double* tile = S + (int64_t)dslot * TT;
    bool deferred = false;
    long long* sdbg = stamps ? stamps + 64 * (int64_t)J : nullptr;
    diag_tile_to_lds<GTG_DF_FENCES == 0>(tile, A, tid);   // PD(J)'s result, handed over by a bulk workgroup
    // The update of the block column right before this tile is applied HERE, in 32-column slices as the substitution of tile (J, J-1)
    // publishes them (the last slice is the only thing left when that tile is final).
    const int sslot = chain_slots[kChainWords * J + 1];
    if (sslot >= 0) {
      const double* sub = S + (int64_t)sslot * TT;   // tile (J, J-1)
      const long long* sflag = tile_flag + sslot;
      // Sub-tile mask of tile (J, J-1) (analysis.hip): a 16 x 16 MFMA tile of the update C -= X_q X_q^T whose two operand patches have no
      // 16-column strip of slice q in common is a sum of exact zeros and is left out.  On a camera system the tile next to the diagonal is
      // full; on a pose graph a fifth of it is (sphere2500: 13.5 live sub-tiles of 64, w20000: 8).
      const unsigned long long xmask = (unsigned)chain_slots[kChainWords * J + 2] | ((unsigned long long)(unsigned)chain_slots[kChainWords * J + 3] << 32);
      auto live = [&](int q, int ra, int rb) {   // row strips ra, rb (16 rows each) of X, slice q = column strips 2 q, 2 q + 1
        const unsigned a = (unsigned)(xmask >> (8 * ra + 2 * q)) & 3u, b = (unsigned)(xmask >> (8 * rb + 2 * q)) & 3u;
        return (a & b) != 0u;
      };
      long long seen = 0;   // the tile's progress word as last read (monotonic)
#pragma unroll 1
      for (int q = 0; q < 4; q++) {
        // (the word was read once more before the previous slice's tasks -- a poll is a memory round trip even when the flag has long been there)
        if (tid < 64 && seen < epoch * 8 + q + 1) wait_flags(sflag, epoch * 8 + q + 1, sflag, epoch * 8 + q + 1, fail, ctrl + 8, 5, J, J - 1, q);
        __syncthreads();   // also: the tile image is complete (q = 0) / the slice buffer is free (q > 0)
        acquired();
        if (sdbg && tid == 0 && q >= 2) sdbg[60 + 2 * (q - 2)] = wall_clock64();   // trace: slice q seen ...
        {  // slice q: rows 0..127, columns 32 q .. 32 q + 31 of the tile below-left -> X[4][SB][PB], 16 bytes x 4 per thread
          double2 v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int e = u * 512 + tid, row = e >> 4, c2 = 2 * (e & 15);
            const double* src = sub + row * T + SB * q + c2;   // X_q of the substitution: handed over (sc1 loads)
#if GTG_DF_FENCES
            v[u] = *reinterpret_cast<const double2*>(src);
#else
            v[u].x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[u].y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int e = u * 512 + tid, row = e >> 4, c2 = 2 * (e & 15);
            double* d = X + (row >> 5) * SB * PB + (row & 31) * PB + c2;
            d[0] = v[u].x; d[1] = v[u].y;
          }
        }
        __syncthreads();
        if (sdbg && tid == 0 && q >= 2) sdbg[61 + 2 * (q - 2)] = wall_clock64();   // ... and in LDS
        if (tid < 64 && q < 3) seen = ld_flag(sflag);   // in flight under the slice tasks
        if (q == 3) {
          // The LAST slice is on the serial chain of the factorisation (the tile left of this one became final a moment ago): only its
          // contribution to the four blocks (ib, 0) -- all that panel 0 of the diagonal tile reads -- and to (1,1), (2,1) is applied here (3 MFMA
          // tiles per wavefront instead of 5), the rest inside potrf_body under panel 0's pivot chain (Xdef).  Bit-identical; measured in round 4
          // (5.12 -> 5.07 ms on L1723, profiles/r04_df_defer_ab.txt) and again in round 5 (profiles/r05a_variants_ab.txt).
          {   // 16 tiles of the blocks (ib, 0), two per wavefront, then one each of the blocks (1,1) and (2,1), which the update of panel 1 reads
            // (8 more tiles here cost 0.6 us; left to potrf_body's panel 0 they needed a third wavefront there, on the pivot chain's SIMD)
            const int t = wave, u = wave + 8;
            const TilePatch p0 = slice_patch(A, X, t >> 2, 0, (t >> 1) & 1, t & 1), p1 = slice_patch(A, X, u >> 2, 0, (u >> 1) & 1, u & 1);
            const bool l0 = live(3, 2 * (t >> 2) + ((t >> 1) & 1), t & 1), l1 = live(3, 2 * (u >> 2) + ((u >> 1) & 1), u & 1);
            if (l0 && l1) upd_tiles2(p0.C, p0.A, p0.B, p1.C, p1.A, p1.B, true, lr, lk);
            else if (l0) upd_tiles2(p0.C, p0.A, p0.B, p0.C, p0.A, p0.B, false, lr, lk);
            else if (l1) upd_tiles2(p1.C, p1.A, p1.B, p1.C, p1.A, p1.B, false, lr, lk);
            const int ib2 = 1 + (wave >> 2);
            const TilePatch p2 = slice_patch(A, X, ib2, 1, (wave >> 1) & 1, wave & 1);
            if (live(3, 2 * ib2 + ((wave >> 1) & 1), 2 + (wave & 1))) upd_tiles2(p2.C, p2.A, p2.B, p2.C, p2.A, p2.B, false, lr, lk);
          }
          deferred = true;
        } else {
          {   // 10 lower blocks: one whole block per wavefront (0 .. 7), then one tile each of the blocks 8 and 9
            int ib, cb;
            lower_block(wave, ib, cb);
            if (live(q, 2 * ib, 2 * cb) || live(q, 2 * ib, 2 * cb + 1) || live(q, 2 * ib + 1, 2 * cb) || live(q, 2 * ib + 1, 2 * cb + 1))
              upd_block4(A + boff(ib, cb), X + ib * SB * PB, X + cb * SB * PB, lr, lk);
            lower_block(8 + (wave >> 2), ib, cb);
            const TilePatch p0 = slice_patch(A, X, ib, cb, (wave >> 1) & 1, wave & 1);
            if (live(q, 2 * ib + ((wave >> 1) & 1), 2 * cb + (wave & 1))) upd_tiles2(p0.C, p0.A, p0.B, p0.C, p0.A, p0.B, false, lr, lk);
          }
        }
      }
    }
    potrf_body(smem_raw, tile, J, Xinv_all + (size_t)J * T * T, fail, stamps ? stamps + 64 * (int64_t)J : nullptr, epoch, tile_flag + dslot, 0, true, GTG_DF_FENCES == 0, pivot_kind, tile_exp,
               deferred ? X : nullptr, deferred ? chain_slots[kChainWords * J + 2] : -1, deferred ? chain_slots[kChainWords * J + 3] : -1);
    __syncthreads();
    if (trace && tid == 0) trace[2 * J + 1] = wall_clock64();
  }
}

#ifndef GT_KERNEL_EMU
__global__ __launch_bounds__(512, 2) void k_df_chain(double* __restrict__ S, double* __restrict__ Xinv_all,
                                                     const long long* __restrict__ pd_flag, long long* tile_flag,
                                                     const int32_t* __restrict__ chain_slots, double* __restrict__ fail,
                                                     const long long epoch, int32_t* __restrict__ ctrl,
                                                     long long* __restrict__ trace, const unsigned char* __restrict__ pivot_kind,
                                                     double* __restrict__ tile_exp, const int32_t* __restrict__ chain_off,
                                                     const int32_t* __restrict__ chain_tiles, long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  chain_loop(smem_raw, S, Xinv_all, pd_flag, tile_flag, chain_slots, fail, epoch, ctrl, trace, chain_tiles + chain_off[blockIdx.x],
             chain_off[blockIdx.x + 1] - chain_off[blockIdx.x], pivot_kind, tile_exp, stamps);
}

// Both roles in ONE kernel (GTG_DF_SINGLE=1): the first n_chain workgroups are the chain (dispatched first, so they are resident before
// anybody waits for them; their upper eight wavefronts leave at once), the others take the tile tasks.  The chain's code is
// compiled for the bulk kernel's 128 registers here (it spills: slower than the two-kernel form) but it is a single dispatch: this is the form rocprofv3's counter collection, which
// serialises kernels, can measure -- two kernels that wait for each other never finish under it.
__global__ __launch_bounds__(kBulkThreads) void k_df_single(double* __restrict__ S, const int32_t* __restrict__ tasks,
                                                      int ntasks, const int32_t* __restrict__ klist,
                                                      long long* __restrict__ tile_flag, long long* __restrict__ part_flag,
                                                      long long* __restrict__ pd_flag,
                                                      const int32_t* __restrict__ chain_slots, double* __restrict__ Xinv_all,
                                                      int32_t* __restrict__ ctrl, double* __restrict__ fail,
                                                      const long long epoch, long long* __restrict__ trace,
                                                      const unsigned char* __restrict__ pivot_kind, double* __restrict__ tile_exp,
                                                      const int32_t* __restrict__ chain_off, const int32_t* __restrict__ chain_tiles, int n_chain) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if ((int)blockIdx.x < n_chain) { if (threadIdx.x < 512) chain_loop(smem_raw, S, Xinv_all, pd_flag, tile_flag, chain_slots, fail, epoch, ctrl, trace ? trace + 8 * (int64_t)ntasks : nullptr, chain_tiles + chain_off[blockIdx.x], chain_off[blockIdx.x + 1] - chain_off[blockIdx.x], pivot_kind, tile_exp); }
  else bulk_loop(smem_raw, S, tasks, ntasks, klist, tile_flag, part_flag, pd_flag, Xinv_all, ctrl, fail, epoch, trace);
}

__global__ void k_df_begin(long long* epoch, long long value, int32_t* ctrl) { *epoch = value; ctrl[0] = 0; ctrl[1] = 0; ctrl[8] = 0; ctrl[2] = ctrl[3] = ctrl[4] = ctrl[5] = -1; }   // (ctrl[6], ctrl[7]: counted over the handle's life)

// The plan's device tables from its host lists (upload_df_plan): one lane per task.  A task (I, J, offset / count of its steps, piece, pieces)
// gets the slots of its tile and of the diagonal tile, its accumulator lanes and the tile's own sub-tile mask; every contraction step k of
// its list the slots of the operand tiles (I, k), (J, k) and their two sub-tile masks.  acc[0] += the MFMAs the step leaves out on
// structurally empty sub-tiles, in half units of 4 MFMAs (a diagonal tile's step counts once, any other twice: integers, so the sum does
// not depend on the order); acc[1] += tiles of the lists without a slot (the host throws).
__global__ __launch_bounds__(256) void k_df_resolve(int64_t n_tasks, int nt, const int32_t* __restrict__ t6, const int32_t* __restrict__ kl,
                                                    const int32_t* __restrict__ slot, const unsigned long long* __restrict__ sub16,
                                                    const int32_t* __restrict__ lane_tab, int32_t* __restrict__ t12, int32_t* __restrict__ steps,
                                                    unsigned long long* __restrict__ acc) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tasks) return;
  const int32_t* d = t6 + 6 * t;
  const int I = d[0], J = d[1];
  int bad = 0;
  auto slot_of = [&](int a, int b) { const int q = slot[(int64_t)a * nt + b]; if (q < 0) bad++; return q; };
  auto mask_of = [&](int a, int k) -> unsigned long long { return a >= nt ? 0xFFull : sub16 ? sub16[(int64_t)a * nt + k] : ~0ull; };
  int32_t* o = t12 + 12 * t;
  const int q = slot_of(I, J);
  for (int x = 0; x < 6; x++) o[x] = d[x];
  o[6] = q; o[7] = slot_of(J, J);
  o[8] = q >= 0 ? lane_tab[2 * q] : 1; o[9] = q >= 0 ? lane_tab[2 * q + 1] : -1;
  { const unsigned long long m = I == J ? ~0ull : mask_of(I, J); o[10] = (int32_t)(uint32_t)m; o[11] = (int32_t)(uint32_t)(m >> 32); }
  unsigned long long skipped = 0;
  for (int32_t e = d[2]; e < d[2] + d[3]; e++) {
    const int k = kl[e];
    int32_t* w = steps + kStepWords * (int64_t)e;
    const unsigned long long ma = mask_of(I, k), mb = mask_of(J, k);
    w[0] = slot_of(I, k); w[1] = slot_of(J, k);
    w[2] = (int32_t)(uint32_t)ma; w[3] = (int32_t)(uint32_t)(ma >> 32); w[4] = (int32_t)(uint32_t)mb; w[5] = (int32_t)(uint32_t)(mb >> 32);
    if (I < nt) {
      int live = 0;
      for (int sp = 0; sp < 8; sp++) live += 4 * __popcll(ma & (0x0101010101010101ull << sp)) * __popcll(mb & (0x0101010101010101ull << sp));
      skipped += (unsigned long long)(2048 - live) * (I == J ? 1u : 2u);
    }
  }
  if (skipped) atomicAdd(&acc[0], skipped);
  if (bad) atomicAdd(&acc[1], (unsigned long long)bad);
}
#endif   // GT_KERNEL_EMU

}  // namespace

#ifndef GT_KERNEL_EMU   // (host code from here on)

// ---- host: task lists --------------------------------------------------------------------------------------------
// tile_struct: (nt x nt) row-major bytes, lower triangle: tile (I, J) holds something before the factorisation
// (nullptr = dense).  Symbolic elimination at tile granularity adds the fill; the rhs row (tile row nt) is dense.
// Two halves: build_df_plan_host is pure host code (no runtime call: analysis.hip runs it on its own thread beside build_chol_plan),
// upload_df_plan resolves the tiles to slots -- which come from the stream schedule's plan -- and makes the device copies.
void build_df_plan(DfPlan& df, int nt, const std::vector<uint8_t>* tile_struct, hipStream_t stream,
                   const std::vector<int32_t>& slot, int64_t n_slots,
                   const std::vector<int32_t>* tile_part, const std::vector<int32_t>* part_parent) {
  build_df_plan_host(df, nt, tile_struct, tile_part, part_parent);
  upload_df_plan(df, stream, slot, n_slots);
}

void build_df_plan_host(DfPlan& df, int nt, const std::vector<uint8_t>* tile_struct,
                        const std::vector<int32_t>* tile_part, const std::vector<int32_t>* part_parent) {
  const bool sect_on = std::getenv("GTG_DF_PLAN_TIMING") != nullptr;
  auto sect_t = std::chrono::high_resolution_clock::now();
  auto sect = [&](const char* what) { if (!sect_on) return; const auto n = std::chrono::high_resolution_clock::now(); std::fprintf(stderr, "[df plan] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(n - sect_t).count()); sect_t = n; };
  std::vector<uint8_t> B((size_t)nt * nt, 0);
  for (int i = 0; i < nt; i++)
    for (int j = 0; j <= i; j++) B[(size_t)i * nt + j] = tile_struct ? (*tile_struct)[(size_t)i * nt + j] : 1;
  for (int i = 0; i < nt; i++) B[(size_t)i * nt + i] = 1;
  for (int k = 0; k < nt; k++) {
    std::vector<int> r;
    for (int i = k + 1; i < nt; i++) if (B[(size_t)i * nt + k]) r.push_back(i);
    for (size_t a = 0; a < r.size(); a++)
      for (size_t b = 0; b <= a; b++) B[(size_t)r[a] * nt + r[b]] = 1;
  }
  std::vector<std::vector<int32_t>> rowcols(nt);
  const int bw = (nt + 63) / 64;
  std::vector<uint64_t> rowbits((size_t)nt * bw, 0);     // the same rows as bit sets: a tile's contraction list is the AND of two of them
  for (int i = 0; i < nt; i++)
    for (int k = 0; k < i; k++) if (B[(size_t)i * nt + k]) { rowcols[i].push_back(k); rowbits[(size_t)i * bw + (k >> 6)] |= 1ull << (k & 63); }

  sect("symbolic fill + row lists");
  // ---- elimination-tree parallelism: several diagonal chains --------------------------------------------------------
  // With a nested-dissection ordering (analysis.hip) the block columns fall into PARTS: leaves that do not touch each other and the
  // separators above them (part_parent; children are numbered before their parent, a part is a contiguous range of tiles).  The
  // reference eliminates independent subtrees of its junction tree concurrently (inference/ClusterTree-inst.h:218-317,
  // base/treeTraversal/parallelTraversalTasks.h:35-156); here every leaf gets its own chain of diagonal tiles:
  //   * the block columns are taken in an order `seq` that interleaves the parts that are ready (all children done) column by
  //     column, pos[c] = place of column c in it.  Ticket order, contraction lists and the placement of early pieces below are all
  //     stated in terms of pos (without parts pos[c] = c and everything is what it was);
  //   * the chain kernel runs 2 workgroups per SLOT (<= 4 slots on the 8 reserved CUs); a leaf takes the next slot round robin,
  //     a separator the slot of its first child; a slot's tiles are taken in pos order, alternating between its two workgroups.
  // Deadlock freedom is unchanged: a task only waits for tasks with smaller tickets and for diagonal tiles whose own inputs have
  // smaller tickets; a chain workgroup walks its tiles in pos order, so the tile it waits for is always the one whose inputs come first.
  const bool tree = tile_part && part_parent && part_parent->size() > 1 && (int)tile_part->size() == nt;
  const int nparts = tree ? (int)part_parent->size() : 1;
  std::vector<int32_t> part_of(nt, 0), seq, pos(nt, 0);
  if (tree) for (int c = 0; c < nt; c++) part_of[c] = (*tile_part)[c];
  {
    std::vector<std::vector<int32_t>> cols(nparts);
    for (int c = 0; c < nt; c++) cols[part_of[c]].push_back(c);
    std::vector<int> pending(nparts, 0), next(nparts, 0);
    if (tree) for (int x = 0; x < nparts; x++) if ((*part_parent)[x] >= 0) pending[(*part_parent)[x]]++;
    std::vector<int> active;
    for (int x = 0; x < nparts; x++) if (pending[x] == 0) active.push_back(x);
    while (!active.empty()) {
      std::vector<int> still;
      for (int x : active) {
        if (next[x] < (int)cols[x].size()) seq.push_back(cols[x][next[x]++]);
        if (next[x] < (int)cols[x].size()) { still.push_back(x); continue; }
        const int par = tree ? (*part_parent)[x] : -1;
        if (par >= 0 && --pending[par] == 0) still.push_back(par);
      }
      active.swap(still);
    }
    if ((int)seq.size() != nt) throw std::runtime_error("dataflow plan: the parts do not cover the block columns");
    for (int q = 0; q < nt; q++) pos[seq[q]] = q;
    // slots and chain lists
    std::vector<int> slot(nparts, 0);
    int nslots = 1;
    if (tree) {
      // (round 4: 8 slots = 16 reserved CUs by default -- with the accumulator lanes of the separators' diagonal tiles in place the leaf
      // chains are the critical path of a pose graph, and they run side by side; up to 16 slots = 32 reserved CUs on request)
      // (16 slots = 32 reserved CUs since round 6: one slot per leaf of a 4-level dissection with room to spare; measured against 8 / 12
      // slots, profiles/r06j_slots_sweep.txt: sphere2500 0.913 / 0.892 / 0.892 ms, w20000 2.71 / 2.78 / 2.58 ms -- the pose graphs' bulk
      // work is a few GFLOP, the CUs cost nothing; what bounds them is the serial top of the tree, profiles/r06k_trace_sphere2500.json)
      constexpr int max_slots = 16;
      std::vector<int> first_child(nparts, -1);
      for (int x = nparts - 1; x >= 0; x--) if ((*part_parent)[x] >= 0) first_child[(*part_parent)[x]] = x;
      int leaves = 0;
      for (int x = 0; x < nparts; x++) if (first_child[x] < 0) slot[x] = leaves++ % max_slots;
      for (int x = 0; x < nparts; x++) if (first_child[x] >= 0) slot[x] = slot[first_child[x]];   // (children precede their parent)
      nslots = std::min(leaves, max_slots);
    }
    std::vector<std::vector<int32_t>> per_slot(nslots);
    for (int q = 0; q < nt; q++) per_slot[slot[part_of[seq[q]]]].push_back(seq[q]);      // pos order
    df.h_chain_off.assign(1, 0); df.h_chain_tiles.clear();
    const int per = nt > 1 ? 2 : 1;
    int longest = 0;
    for (int sl = 0; sl < nslots; sl++) {
      longest = std::max(longest, (int)per_slot[sl].size());
      for (int w = 0; w < per; w++) {
        for (size_t i = w; i < per_slot[sl].size(); i += per) df.h_chain_tiles.push_back(per_slot[sl][i]);
        df.h_chain_off.push_back((int32_t)df.h_chain_tiles.size());
      }
    }
    df.n_chain = (int)df.h_chain_off.size() - 1;
    df.h_seq = seq;
    df.critical_tiles = longest;
  }
  auto by_pos = [&](std::vector<int32_t>& v) { if (tree) std::sort(v.begin(), v.end(), [&](int32_t a, int32_t b) { return pos[a] < pos[b]; }); };

  df.h_tasks.clear(); df.h_klist.clear();
  const double t3 = (double)T * T * T;
  double flops = 0.0; int64_t stored = 0;
  // A tile's contraction is serial on one CU (14 us per step, up to ~30 steps) and the workgroups in flight cover only ~9 block
  // columns: a long contraction taken when its column comes up would not be done when the column's diagonal tile is factored.
  // So a long list is cut into PIECES of at most kPiece steps; piece r accumulates in place (tile -= its steps, flag
  // part_flag = kPieceBase epoch + r + 1) and is queued EARLIER than the tile's own column: right behind the block column of its
  // youngest operand, where everything it reads is final and it runs without waiting.  Only the last piece (the youngest
  // kFinal steps + the substitution) sits in the tile's column and streams behind the columns before it.
  // (kPiece, kFinal) swept on the L1723 shape in round 4 (Cholesky ms): (6,3) 5.17, (4,4) 5.09, (6,4) 5.10, (4,5) 5.11, (3,4) 5.14,
  // (4,3) 5.15, (4,6) 5.16, (8,3) 5.32, (2,4) 5.32, (6,2) 5.48, (10,3) 5.51.
  constexpr int kPiece = 4, kFinal = 4;
  // The DIAGONAL tile's last piece is shorter: PD(J) is on the serial chain -- the chain workgroup of tile J cannot start before it -- and a
  // last piece of 4 steps is 72 us of contraction (18 us per step) that only starts when a workgroup takes its ticket.  Round 5 trace
  // (profiles/r05f_late_pd.txt): on 13 of 120 block columns of the L1723 shape the ticket was taken 53 - 74 us before the tile was needed
  // instead of the usual 77 - 120, PD(J) came 7 - 20 us late, and those 13 periods (57 us instead of 41) were 0.27 ms of the 5.05.
  // The same holds for the tiles right below it: (J + 1, J) feeds the next diagonal tile through the chain kernel, (J + 2, J) is the youngest
  // operand of PD(J + 2) AND of the critical tile (J + 2, J + 1) -- when its last piece was taken 40 - 60 us before its column's diagonal
  // tile was out (normally: 100), it was final 25 us late and both were late with it -- and (J + 3, J) is the youngest operand of that one.
  // Swept on the L1723 shape (profiles/r05_plan_cut_sweep.txt; Cholesky ms): last piece of the diagonal tile 1 / 2 / 3 / 4 steps 5.17 / 5.04 / 5.03 /
  // 5.09; short pieces + pulled for 0 / 2 / 3 rows below it 5.07 / 5.03 / 5.00.
#ifndef GT_DF_FINAL_DIAG
#define GT_DF_FINAL_DIAG 2
#endif
#ifndef GT_DF_NEAR_ROWS
#define GT_DF_NEAR_ROWS 3
#endif
  constexpr int kFinalDiag = GT_DF_FINAL_DIAG, kNearRows = GT_DF_NEAR_ROWS;
  struct Rec { int32_t I, J, koff, kcnt, r, R; };
  std::vector<std::vector<Rec>> finals(nt), early(nt);      // by place in `seq`
  std::vector<int> diag_last_g(nt + 1, -1);                  // group of the last early piece of PD(J) (-1: none)
  std::vector<int> cut;
  { size_t tiles = 0; for (int i = 0; i < nt; i++) tiles += rowcols[i].size();       // (room up front: a tile has a last piece and ~ a piece per 4 steps)
    for (int g = 0; g < nt; g++) { finals[g].reserve(rowcols[g].size() + nt / 2 + 4); early[g].reserve(2 * (tiles / nt + 1) + 16); }
    df.h_klist.reserve(16 * tiles); }
  auto emit = [&](int I, int J, std::vector<int32_t>& ks) {
    by_pos(ks);                                        // the steps in the order in which their operands come into being
    const int n = (int)ks.size(), G = pos[J];
    const bool near = I - J <= kNearRows;              // the diagonal tile and the kNearRows tiles below it: on or next to the serial chain
    int m = std::max(0, n - (near ? kFinalDiag : kFinal));   // older steps, in early pieces of at most kPiece; the last piece: the youngest
    while (m > 0 && pos[ks[m - 1]] > G - 3) m--;     // (an early piece sits two groups before its tile's column at the latest)
    const int f = n - m;
    // the early steps cut into pieces of kPiece from the old end; a DIAGONAL tile's last early piece is short as well (kFinalDiag steps):
    // the last piece cannot start before it is done, and 4 steps are 73 us (round 5 trace: PD(J) late whenever that piece had 4)
    cut.clear();             // piece r = steps [cut[r], cut[r + 1])
    {
      const int tail = (near && m > kFinalDiag) ? kFinalDiag : 0;
      for (int b = 0; b < m - tail; b += kPiece) cut.push_back(b);
      if (tail) cut.push_back(m - tail);
      cut.push_back(m);
    }
    const int R = (int)cut.size();     // early pieces + the last one
    if (R >= (int)kPieceBase) throw std::runtime_error("dataflow plan: a contraction list needs more pieces than the part flags can count");
    const int32_t off = (int32_t)df.h_klist.size();
    df.h_klist.insert(df.h_klist.end(), ks.begin(), ks.end());
    int gprev = 0;
    for (int r = 0; r + 1 < R; r++) {
      const int b = cut[r], e = cut[r + 1];
      const int last_k = ks[e - 1];
      const int g = std::max(pos[last_k] + 1, gprev);   // as early as its operands exist (<= G - 2)
      early[g].push_back(Rec{I, J, off + b, e - b, r, R}); gprev = g;
      if (I == J) diag_last_g[J] = g;
    }
    finals[G].push_back(Rec{I, J, off + m, f, R - 1, R});
  };
  sect("parts, chains");
  std::vector<int32_t> ks;
  std::vector<int32_t> has_sub(nt, 0);
  for (int J = 0; J < nt; J++) {
    // the contribution of block column J-1 to the diagonal tile is applied by k_df_chain itself (streamed): not in PD's list
    ks = rowcols[J];
    if (!ks.empty() && ks.back() == J - 1) { ks.pop_back(); has_sub[J] = 1; }
    // (also streaming column J-2's contribution in the chain workgroup was measured in round 4: PD(J) is never late then -- the p90 of
    // the chain period falls from 59 to 54 us -- but the extra 16 us of slices push the MEDIAN period from 40.0 to 41.6 us: 5.25 ->
    // 5.49 ms on L1723.  Removed.)
    emit(J, J, ks);
    flops += t3 / 3.0 + (double)rowcols[J].size() * t3; stored++;
    for (int I = J + 1; I < nt; I++) {
      if (!B[(size_t)I * nt + J]) continue;
      ks.clear();
      for (int w = 0; w < bw; w++) {
        uint64_t both = rowbits[(size_t)I * bw + w] & rowbits[(size_t)J * bw + w];
        while (both) { ks.push_back(64 * w + __builtin_ctzll(both)); both &= both - 1; }
      }
      emit(I, J, ks);
      flops += t3 + (double)ks.size() * 2.0 * t3; stored++;
    }
    ks = rowcols[J];
    emit(nt, J, ks);   // rhs row: y_J (flops not counted, as in the right-looking plan)
  }
  // (Round 6, measured and removed -- profiles/r06i_level_sweep.txt, tools/df_plan_load.py: early pieces are queued As Soon As Possible, and
  // on a banded system they arrive in bursts: the work of a ticket-order iteration swings between 0.4 and 1.9 times what 246 workgroups get
  // done in a chain period, 39 of 122 iterations of the L1723 shape are over.  A forward sweep that moved the pieces with the latest
  // deadlines out of overfull iterations (same pieces, same order inside a tile: bit-identical) levelled that profile completely -- and the
  // factorisation went from 5.00 to 5.97 ms (capacity = 100 % of a period's work; 9.2 ms at 90 %, no change at 120 - 150 %).  The aggregate
  // is not what binds: a tile's pieces are a serial chain of ~70 us links, and every deferral shortens the time that chain has.)
  // ticket order: the own tasks of the column in place q (diagonal accumulation, the tile below it, ...), then the early pieces whose
  // youngest operand is in place q - 2: the latency-critical tasks of a column are taken a whole group of background work ahead of the
  // pieces that merely have to be done some columns later (with the early pieces of group q - 1 in front of them, the diagonal
  // accumulation was taken 30 us before it was needed and the chain waited 20 us for it every few columns)
  sect("contraction lists, pieces");
  { size_t recs = 0; for (int g = 0; g < nt; g++) recs += finals[g].size() + early[g].size(); df.h_tasks.reserve(6 * recs); }
  auto put1 = [&](const Rec& t) { const int32_t f[6] = {t.I, t.J, t.koff, t.kcnt, t.r, t.R}; df.h_tasks.insert(df.h_tasks.end(), f, f + 6); };
  auto put = [&](const std::vector<Rec>& v) { for (const Rec& t : v) put1(t); };
  // One chain: the two tasks of a column that the serial chain waits for -- PD(J) and the tile right below the diagonal tile -- are
  // queued one group EARLIER than the rest of their column, in front of the previous group's burst of early pieces (and behind their
  // own early pieces of that burst, so that a task still only waits for smaller tickets).  A burst is several hundred pieces of
  // ~100 us: behind it PD(J) was taken 12 - 30 us too late on every sixth column of the L1723 shape and the chain period stretched
  // from 40 to 60 - 76 us (round 4 trace: 19 of 121 periods, 0.4 ms of 5.2).  Taken early the two tasks just hold two of the 248
  // workgroups a period longer.
  // (pulled with the diagonal tile: the kNearRows tiles below it; measured on L1723 in round 4, with last pieces of 4 steps everywhere: 0 rows
  // below 5.17 ms, 1 row 5.17, 3 rows 5.19; no pulling at all 5.30)
  const int pull_rows = tree ? -1 : kNearRows;
  const bool pull = pull_rows >= 0;
  auto critical = [&](const Rec& t, int c) { return t.J == c && t.I >= c && t.I <= c + pull_rows && t.I < nt; };
  // PD(c) itself goes one group further ahead (round 5): its last piece -- two contraction steps, 36 us, the second one streaming behind the
  // substitution of tile (c, c - 2) -- was still taken only 20 - 25 us before the chain needed it on some columns of the L1723 shape
  // (230 tickets behind the tile (c, c - 1) of the group before: profiles/r05i trace).  Allowed when all its early pieces sit in groups
  // <= q - 1 (they are queued in front of it then: a task still only waits for smaller tickets; its operand tiles (c, c - 3), (c, c - 2)
  // belong to columns <= q, whose own tasks are queued by then).
  std::vector<char> pulled(nt + 1, 0), pd_out(nt + 2, 0);
  auto is_pd = [&](const Rec& t, int c) { return t.I == c && t.J == c; };
  for (int q = 0; q < nt; q++) {
    for (const Rec& t : finals[q]) { if (pulled[q] && critical(t, q)) continue; put1(t); }
    if (q >= 1) {
      const int c = q + 1, c2 = q + 2;            // the columns whose critical tasks are pulled in front of early[q - 1]
      if (pull && c < nt) {
        const bool pd2 = c2 < nt && diag_last_g[c2] <= q - 1;
        for (const Rec& t : early[q - 1]) if (critical(t, c) || (pd2 && is_pd(t, c2))) put1(t);
        for (const Rec& t : finals[c]) if (critical(t, c) && !(is_pd(t, c) && pd_out[c])) put1(t);
        pulled[c] = 1;
        if (pd2) { for (const Rec& t : finals[c2]) if (is_pd(t, c2)) put1(t); pd_out[c2] = 1; }
        for (const Rec& t : early[q - 1]) if (!(critical(t, c) || (pd2 && is_pd(t, c2)))) put1(t);
      } else put(early[q - 1]);
    }
  }
  put(early[nt - 1]);
  df.nt = nt; df.n_tasks = (int64_t)df.h_tasks.size() / 6;
  df.flops = flops; df.dense_fraction = (double)stored / ((double)nt * (nt + 1) / 2.0);
  if (df.h_klist.empty()) df.h_klist.push_back(0);
  df.h_has_sub = has_sub;
  sect("ticket order");
}

void upload_df_plan(DfPlan& df, hipStream_t stream, const std::vector<int32_t>& slot, int64_t n_slots, const std::vector<uint64_t>* sub16, const int32_t* d_slot) {
  const int nt = df.nt;
  const std::vector<int32_t>& has_sub = df.h_has_sub;
  // Device form: tiles AND their flag words are addressed by SLOT (context.h::SMat; the slots come from the stream schedule's plan,
  // whose stored set -- fill at the granularity of column pairs -- contains this one).  A task carries the slots of its tile and of
  // its diagonal tile, a contraction step the slots of its two operand tiles; h_tasks / h_klist keep the tile coordinates (debug
  // getters, tests/test_chol_plan.py).
  auto slot_of = [&](int I, int J) {
    const int32_t q = slot[(size_t)I * nt + J];
    if (q < 0) throw std::runtime_error("dataflow plan: a tile of the task list has no slot in the stored-tile list");
    return q;
  };
  {
    // accumulator lanes of the tiles with very long contraction lists (run_task): G - 1 scratch slots each, behind the stored tiles
    constexpr int lane_min = 8;   // early pieces from which a tile gets lanes
    constexpr int lane_max = 8;
    std::map<int32_t, std::pair<int32_t, int32_t>> lanes;   // slot of the tile -> (G, first scratch slot)
    df.n_scratch = 0;
    for (int64_t t = 0; t < df.n_tasks; t++) {
      const int32_t* d = df.h_tasks.data() + 6 * t;
      const int E = d[5] - 1;
      if (d[4] != d[5] - 1 || E < lane_min || lane_max < 2) continue;       // (looked at once per tile: at its final piece)
      const int G = std::min(lane_max, E / 4);
      if (G < 2) continue;
      lanes[slot_of(d[0], d[1])] = {G, (int32_t)(n_slots + df.n_scratch)};
      df.n_scratch += G - 1;
    }
    // sub-tile masks of a contraction step's operand tiles (analysis.hip: strip-level symbolic factorisation; none = every sub-tile): the
    // right-hand-side row has one row of sub-tiles
    auto mask_of = [&](int I, int k) -> uint64_t { return I >= nt ? 0xFFull : sub16 ? (*sub16)[(size_t)I * nt + k] : ~0ull; };
    double skipped = 0.0;
    if (d_slot && df.n_tasks >= 1024 && !getenv("GTG_HOST_SYMBOLIC")) {
      // On the DEVICE (k_df_resolve, one lane per task; d_slot = the device copy of `slot`): the host lists go up as they are (6 words per
      // task, one per step) instead of the resolved tables (12 and 6 words), and the 64 x 64 bit products of the skipped-MFMA count run
      // there as well (2.4 ms of host loop on the L1723 shape).  GTG_HOST_SYMBOLIC=1 keeps the loop below: same tables, same count
      // (tests/test_gpu_device_analysis.py); so do the dry-run runtime of the CPU tests and small plans.
      std::vector<int32_t> lane_tab(2 * (size_t)n_slots);
      for (int64_t q = 0; q < n_slots; q++) { lane_tab[2 * q] = 1; lane_tab[2 * q + 1] = -1; }
      for (const auto& kv : lanes) { lane_tab[2 * (size_t)kv.first] = kv.second.first; lane_tab[2 * (size_t)kv.first + 1] = kv.second.second; }
      auto al = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
      const size_t b_t6 = al(4 * df.h_tasks.size()), b_kl = al(4 * df.h_klist.size()), b_lane = al(4 * lane_tab.size()), b_sub = sub16 ? al(8 * sub16->size()) : 0;
      DevBuf<unsigned char> ws; ws.alloc(b_t6 + b_kl + b_lane + b_sub + 256);
      struct Release { DevBuf<unsigned char>& b; ~Release() { b.free(); } } release{ws};
      int32_t* d_t6 = reinterpret_cast<int32_t*>(ws.p); int32_t* d_kl = reinterpret_cast<int32_t*>(ws.p + b_t6);
      int32_t* d_lane = reinterpret_cast<int32_t*>(ws.p + b_t6 + b_kl);
      unsigned long long* d_sub = sub16 ? reinterpret_cast<unsigned long long*>(ws.p + b_t6 + b_kl + b_lane) : nullptr;
      unsigned long long* d_acc = reinterpret_cast<unsigned long long*>(ws.p + b_t6 + b_kl + b_lane + b_sub);
      check_hip(hipMemcpyAsync(d_t6, df.h_tasks.data(), 4 * df.h_tasks.size(), hipMemcpyHostToDevice, stream), "H2D");
      check_hip(hipMemcpyAsync(d_kl, df.h_klist.data(), 4 * df.h_klist.size(), hipMemcpyHostToDevice, stream), "H2D");
      check_hip(hipMemcpyAsync(d_lane, lane_tab.data(), 4 * lane_tab.size(), hipMemcpyHostToDevice, stream), "H2D");
      if (sub16) check_hip(hipMemcpyAsync(d_sub, sub16->data(), 8 * sub16->size(), hipMemcpyHostToDevice, stream), "H2D");
      check_hip(hipMemsetAsync(d_acc, 0, 16, stream), "memset");
      df.tasks.alloc(12 * (size_t)df.n_tasks); df.klist.alloc(kStepWords * df.h_klist.size());
      hipLaunchKernelGGL(k_df_resolve, dim3((unsigned)((df.n_tasks + 255) / 256)), dim3(256), 0, stream, df.n_tasks, nt, d_t6, d_kl, d_slot, d_sub, d_lane,
                         df.tasks.p, df.klist.p, d_acc);
      check_hip(hipGetLastError(), "df plan resolve");
      unsigned long long acc[2] = {0, 0};
      check_hip(hipMemcpyAsync(acc, d_acc, 16, hipMemcpyDeviceToHost, stream), "D2H");
      check_hip(hipStreamSynchronize(stream), "df plan resolve");
      if (acc[1]) throw std::runtime_error("dataflow plan: a tile of the task list has no slot in the stored-tile list");
      skipped = (double)acc[0] * 0.5 * 2048.0;
    } else {
    std::vector<int32_t> dt; dt.reserve((size_t)df.n_tasks * 12);
    std::vector<int32_t> dk(kStepWords * df.h_klist.size(), 0);
    std::vector<uint8_t> seen(df.h_klist.size(), 0);
    for (int64_t t = 0; t < df.n_tasks; t++) {
      const int32_t* d = df.h_tasks.data() + 6 * t;
      const int I = d[0], J = d[1];
      for (int x = 0; x < 6; x++) dt.push_back(d[x]);
      dt.push_back(slot_of(I, J)); dt.push_back(slot_of(J, J));
      { const auto it = lanes.find(slot_of(I, J)); dt.push_back(it == lanes.end() ? 1 : it->second.first); dt.push_back(it == lanes.end() ? -1 : it->second.second); }
      { const uint64_t m = I == J ? ~0ull : mask_of(I, J); dt.push_back((int32_t)(uint32_t)m); dt.push_back((int32_t)(uint32_t)(m >> 32)); }   // sub-tile mask of the tile itself (substitute)
      for (int32_t e = d[2]; e < d[2] + d[3]; e++) {   // (the pieces of a tile share one list: every entry is visited once)
        const int k = df.h_klist[e];
        int32_t* w = dk.data() + kStepWords * (size_t)e;
        const uint64_t ma = mask_of(I, k), mb = mask_of(J, k);
        w[0] = slot_of(I, k); w[1] = slot_of(J, k);
        w[2] = (int32_t)(uint32_t)ma; w[3] = (int32_t)(uint32_t)(ma >> 32); w[4] = (int32_t)(uint32_t)mb; w[5] = (int32_t)(uint32_t)(mb >> 32);
        if (!seen[e]) {   // MFMAs the step leaves out: per 16-column strip (row tiles with a live A sub-tile) x (column tiles with a live B sub-tile), 4 each
          int live = 0;
          for (int sp = 0; sp < 8; sp++) live += 4 * __builtin_popcountll(ma & (0x0101010101010101ull << sp)) * __builtin_popcountll(mb & (0x0101010101010101ull << sp));
          const int rows = I >= nt ? 1 : 8;   // (the flop count has always taken the right-hand-side row as a full tile row; its 7 idle row tiles are not counted as skipped work here)
          skipped += (I == J ? 0.5 : 1.0) * (double)(4 * 8 * rows * 8 - live) * 2048.0 * (I >= nt ? 0.0 : 1.0);
        }
        seen[e] = 1;
      }
    }
    df.tasks.upload(dt.data(), dt.size(), stream);
    df.klist.upload(dk.data(), dk.size(), stream);
    }
    df.flops_executed = df.flops - skipped;
    std::vector<int32_t> cs(kChainWords * (size_t)nt, -1);
    for (int J = 0; J < nt; J++) {
      cs[kChainWords * J] = slot_of(J, J);
      if (has_sub[J] & 1) {
        cs[kChainWords * J + 1] = slot_of(J, J - 1);
        const uint64_t m = mask_of(J, J - 1);
        cs[kChainWords * J + 2] = (int32_t)(uint32_t)m; cs[kChainWords * J + 3] = (int32_t)(uint32_t)(m >> 32);
      }
    }
    df.has_sub.upload(cs.data(), cs.size(), stream);   // (per diagonal tile: kChainWords)
    check_hip(hipStreamSynchronize(stream), "df plan upload");
  }
  df.chain_off.upload(df.h_chain_off.data(), df.h_chain_off.size(), stream);
  df.chain_tiles.upload(df.h_chain_tiles.data(), df.h_chain_tiles.size(), stream);
  df.shadow = (n_slots + df.n_scratch + 511) / 512 * 512;   // flag words by slot, the scratch slots of the accumulator lanes included (the name is from rounds 3 - 5, when every flag had a shadow word this far behind it)
  df.tile_flag.alloc((size_t)df.shadow); df.part_flag.alloc((size_t)df.shadow); df.pd_flag.alloc((size_t)df.shadow); df.ctrl.alloc(32);
  if (getenv("GTG_DF_TRACE")) { df.trace.alloc(8 * (size_t)df.n_tasks + 66 * (size_t)nt); check_hip(hipMemsetAsync(df.trace.p, 0, sizeof(long long) * df.trace.n, stream), "memset"); }
  check_hip(hipMemsetAsync(df.tile_flag.p, 0, sizeof(long long) * df.tile_flag.n, stream), "memset");
  check_hip(hipMemsetAsync(df.pd_flag.p, 0, sizeof(long long) * df.pd_flag.n, stream), "memset");
  check_hip(hipMemsetAsync(df.part_flag.p, 0, sizeof(long long) * df.part_flag.n, stream), "memset");
  check_hip(hipMemsetAsync(df.ctrl.p, 0, sizeof(int32_t) * 32, stream), "memset");
  check_hip(hipStreamSynchronize(stream), "df plan upload");
}

void free_df_plan(DfPlan& df) {
  df.tasks.free(); df.klist.free(); df.tile_flag.free(); df.part_flag.free(); df.pd_flag.free(); df.ctrl.free(); df.trace.free(); df.has_sub.free();
  df.chain_off.free(); df.chain_tiles.free();
}

// GTG_CHOL=streams selects the per-column launch sequence of cholesky.hip instead of the dataflow pass (the A/B of the tests; read per
// call: a test switches it inside one process)
bool dataflow_schedule_selected() {
  const char* sched = getenv("GTG_CHOL");
  return !(sched && std::string(sched) == "streams");
}

// The dynamic-LDS attributes of the three kernels and the device's two CU-masked streams: once per (device, reserved CUs), created by the
// first factorisation of a process -- or ahead of it by gtg_prewarm (below).
struct DfStreams { hipStream_t bulk = nullptr, chain = nullptr; hipEvent_t ev_start = nullptr, ev_chain = nullptr, ev_bulk = nullptr; int grid = 0; };
static DfStreams& df_streams(int device, int reserve) {
  static std::set<int> attr_set;
  static std::mutex attr_mutex;
  {
    std::lock_guard<std::mutex> lock(attr_mutex);
    if (!attr_set.count(device)) {
      check_hip(hipFuncSetAttribute((const void*)k_df_bulk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmemBulk), "smem attr");
      check_hip(hipFuncSetAttribute((const void*)k_df_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmemChain), "smem attr");
      check_hip(hipFuncSetAttribute((const void*)k_df_single, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(kSmemChain, kSmemBulk)), "smem attr");
      attr_set.insert(device);
    }
  }
  // The two CU-masked streams and their events belong to the DEVICE, not to the handle: a process runs one dataflow
  // factorisation per device at a time (the caller holds that device's lock, api.hip), and creating masked streams costs
  // milliseconds -- per handle that was 5 ms on the first lambda try of every new optimizer.
  static std::map<std::pair<int, int>, DfStreams> per_device;   // (device, reserved CUs)
  static std::mutex per_device_mutex;
  std::lock_guard<std::mutex> lock(per_device_mutex);            // (held over the creation: gtg_prewarm's thread and a first factorisation may meet here)
  DfStreams& ds = per_device[{device, reserve}];
  if (!ds.bulk) {
    // Two CU-masked streams with complementary masks: the bulk kernel's workgroups stay off a few CUs, and k_df_chain (97 KB
    // of LDS, the whole register file of its SIMDs) can only be placed on exactly those -- so it is placed at once, whatever
    // the order in which the two kernels reach the dispatcher.  (With an unmasked chain stream the dispatcher may pick a
    // shader engine whose CUs the persistent bulk workgroups have filled, and the chain then waits for the end of the
    // factorisation it is needed for: measured as a 2 s stall that ends in the wait bound.)
    // Mask semantics measured on MI355X (tools/cu_mask_probe.hip): bit i = XCD i % 8, shader engine (i / 8) % 4, CU i / 32 of
    // it; an XCD whose bits are ALL zero is not excluded but fully enabled.  A mask that really confines a kernel therefore
    // needs a bit in every XCD: the chain's mask is the last CU of every XCD (8 CUs, 3 % of the chip), the bulk's the rest.
    hipDeviceProp_t prop;
    check_hip(hipGetDeviceProperties(&prop, device), "props");
    const int ncu = std::max(prop.multiProcessorCount, 16);
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u), inv((ncu + 31) / 32, 0u);
    for (int i = 0; i < ncu; i++) (i < ncu - reserve ? mask : inv)[i >> 5] |= 1u << (i & 31);
    check_hip(hipExtStreamCreateWithCUMask(&ds.bulk, (uint32_t)mask.size(), mask.data()), "masked stream");
    check_hip(hipExtStreamCreateWithCUMask(&ds.chain, (uint32_t)inv.size(), inv.data()), "masked stream");
    check_hip(hipEventCreateWithFlags(&ds.ev_start, hipEventDisableTiming), "event");
    check_hip(hipEventCreateWithFlags(&ds.ev_chain, hipEventDisableTiming), "event");
    check_hip(hipEventCreateWithFlags(&ds.ev_bulk, hipEventDisableTiming), "event");
    const char* g = getenv("GTG_DF_GRID");
    ds.grid = g ? atoi(g) : (ncu - reserve);
  }
  return ds;
}

// The masked stream pair a plan needs (and the kernels' dynamic-LDS attributes) ahead of its first factorisation: analysis.hip calls this as
// soon as the plan's chains are known, and the creation -- 22 ms in a cold process, all of it inside the first lambda try before (round 6,
// GTG_DEBUG_TIMING's "first" lines) -- runs on a helper thread under the rest of the upload, the caller's set-up and the first
// linearisation; launch_cholesky_df joins it.  A pair that exists already costs a map lookup.  (One thread per (device, reserved CUs) for
// the life of the process; gtg_destroy joins what is still running, so that no thread outlives the library's last handle.)
static int df_reserve_for(int n_chain) { return n_chain > 16 ? 32 : n_chain > 8 ? 16 : 8; }   // one CU per chain workgroup, a bit in every XCD (see df_streams)
// (the thread objects live in a map that is never destroyed: a process that uploads a problem and exits without a factorisation or a
// gtg_destroy must not run into std::terminate for a joinable thread; exit joins them as well)
static std::mutex g_prepare_mu;
static std::map<std::pair<int, int>, std::thread>& prepare_threads() { static auto* m = new std::map<std::pair<int, int>, std::thread>; return *m; }
void df_join_prepared() {
  std::lock_guard<std::mutex> lock(g_prepare_mu);
  for (auto& kv : prepare_threads()) if (kv.second.joinable()) kv.second.join();
}
void df_prepare_streams_async(int device, int n_chain) {
  if (std::getenv("GTG_SYNC_STREAM_PAIR")) return;   // (A/B: the pair is created by the first factorisation, as before round 6)
  const int reserve = df_reserve_for(n_chain);
  std::lock_guard<std::mutex> lock(g_prepare_mu);
  auto& threads = prepare_threads();
  if (threads.count({device, reserve})) return;
  if (threads.empty()) std::atexit([] { df_join_prepared(); });
  threads[{device, reserve}] = std::thread([device, reserve] {
    try { check_hip(hipSetDevice(device), "hipSetDevice"); (void)df_streams(device, reserve); } catch (...) { (void)hipGetLastError(); }   // (a failure shows again, with its message, in the first factorisation)
  });
}

// fail[0]: non-positive pivot (Eigen LLT NumericalIssue); fail[1]: a dependency wait hit its bound
void launch_cholesky_df(gtg_context& c, SMat Sm, int NP, DfPlan& df, double* Xinv, double* fail,
                        const unsigned char* pivot_kind, double* tile_exp) {
  const int nt = NP / T;
  double* S = Sm.p;
  if (df.nt != nt) throw std::runtime_error("dataflow cholesky plan does not match the matrix");
  const int reserve = df_reserve_for(df.n_chain);
  df_join_prepared();
  const auto pair_t0 = std::chrono::high_resolution_clock::now();
  DfStreams& ds = df_streams(c.device, reserve);
  { static std::atomic<int> first{0};    // GTG_DEBUG_TIMING: what the first factorisation of the process pays for the kernels' attributes and its masked stream pair
    if (first.fetch_add(1) == 0 && std::getenv("GTG_DEBUG_TIMING"))
      std::fprintf(stderr, "[gtsam_amd first ] %-40s %8.2f ms\n", "(attributes + masked stream pair)", std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - pair_t0).count()); }
  // the factorisation's epoch: counted on the host and handed to both kernels BY VALUE -- the ticket counter is only ever touched by
  // atomics, the flags by write-through stores and sc1 loads, and nothing the two kernels synchronise through is a word that a
  // kernel of the previous factorisation wrote with a plain store
  const long long epoch = ++c.chol_epoch;
  hipLaunchKernelGGL(k_df_begin, dim3(1), dim3(1), 0, c.stream, c.chol_epoch_dev.p, epoch, df.ctrl.p);
  static const bool single = getenv("GTG_DF_SINGLE") != nullptr;
  if (single) {
    hipDeviceProp_t prop;
    check_hip(hipGetDeviceProperties(&prop, c.device), "props");
    const int g1 = (int)std::min<int64_t>(prop.multiProcessorCount, df.n_tasks + df.n_chain);
    hipLaunchKernelGGL(k_df_single, dim3(g1), dim3(kBulkThreads), std::max(kSmemChain, kSmemBulk), c.stream, S, df.tasks.p, (int)df.n_tasks,
                       df.klist.p, df.tile_flag.p, df.part_flag.p, df.pd_flag.p, df.has_sub.p, Xinv, df.ctrl.p, fail, epoch, df.trace.p, pivot_kind, tile_exp,
                       df.chain_off.p, df.chain_tiles.p, df.n_chain);
    check_hip(hipGetLastError(), "cholesky (dataflow, single kernel)");
    return;
  }
  check_hip(hipEventRecord(ds.ev_start, c.stream), "record");
  check_hip(hipStreamWaitEvent(ds.chain, ds.ev_start, 0), "wait");
  check_hip(hipStreamWaitEvent(ds.bulk, ds.ev_start, 0), "wait");
  // test hook (tests/test_gpu_dataflow_protocol.py): GTG_DF_TEST_TIMEOUT=n leaves the chain kernel out of the process's n-th dataflow
  // factorisation -- the bulk kernel's waits then run into their bound, exactly what a chain kernel that was never placed looks like
  // ("n:m": out of m consecutive ones from the n-th on -- two in a row make the repeated try time out as well)
  static const char* const drop_env = getenv("GTG_DF_TEST_TIMEOUT");
  static const int drop_at = drop_env ? atoi(drop_env) : 0;
  static const int drop_n = (drop_env && strchr(drop_env, ':')) ? atoi(strchr(drop_env, ':') + 1) : 1;
  static std::atomic<int> launches{0};
  const int launch_no = ++launches;
  const bool drop_chain = drop_at > 0 && launch_no >= drop_at && launch_no < drop_at + drop_n;
  if (!drop_chain)
  hipLaunchKernelGGL(k_df_chain, dim3(df.n_chain), dim3(512), kSmemChain, ds.chain, S, Xinv, df.pd_flag.p, df.tile_flag.p, df.has_sub.p, fail, epoch, df.ctrl.p,
                     df.trace.p ? df.trace.p + 8 * df.n_tasks : nullptr, pivot_kind, tile_exp, df.chain_off.p, df.chain_tiles.p,
                     df.trace.p ? df.trace.p + 8 * df.n_tasks + 2 * (int64_t)nt : nullptr);
  const int grid = (int)std::min<int64_t>(ds.grid, df.n_tasks);
  hipLaunchKernelGGL(k_df_bulk, dim3(grid), dim3(kBulkThreads), kSmemBulk, ds.bulk, S, df.tasks.p, (int)df.n_tasks, df.klist.p,
                     df.tile_flag.p, df.part_flag.p, df.pd_flag.p, Xinv, df.ctrl.p, fail, epoch, df.trace.p);
  // A short second launch of the bulk kernel BEHIND the chain kernel in its stream (six workgroups on the reserved CUs, same
  // ticket counter).  It was meant to share the tail of the factorisation; the profile shows that it finds next to nothing to do
  // (4 us: the tail after the last diagonal tile is 5 us of work) -- and yet the factorisation is reproducibly 1.5 % shorter
  // with it (L1723, interleaved A/B on fresh boxes: 5.28-5.32 ms against 5.38-5.39 ms): the
  // end of a stream is observed sooner behind a short kernel than directly behind a persistent one (an empty kernel behind the
  // bulk kernel instead has the same effect: 5.28 ms).  It must not
  // start earlier: a third persistent kernel beside the chain on the reserved CUs (tried: its own stream with the chain's
  // mask) starved the chain -- wait bounds hit.
  constexpr int extra = 6;
  if (df.n_tasks > grid)
    hipLaunchKernelGGL(k_df_bulk, dim3(extra), dim3(kBulkThreads), kSmemBulk, ds.chain, S, df.tasks.p, (int)df.n_tasks, df.klist.p,
                       df.tile_flag.p, df.part_flag.p, df.pd_flag.p, Xinv, df.ctrl.p, fail, epoch, df.trace.p);
  check_hip(hipEventRecord(ds.ev_chain, ds.chain), "record");
  check_hip(hipEventRecord(ds.ev_bulk, ds.bulk), "record");
  check_hip(hipStreamWaitEvent(c.stream, ds.ev_chain, 0), "wait");
  check_hip(hipStreamWaitEvent(c.stream, ds.ev_bulk, 0), "wait");
  check_hip(hipGetLastError(), "cholesky (dataflow)");
}

// gtg_prewarm: this unit's kernels (kernels.h).  NOT the masked streams: which pair a graph needs (8, 16 or 32 reserved CUs) is only known
// after its analysis, and an idle second pair costs the running one 10 % -- sphere2500 0.89 -> 0.99 ms per factorisation with a prewarmed
// 8-CU pair beside its own 32-CU pair (round 6; the runtime multiplexes streams onto a few hardware queues).  Destroying the pair that is
// not needed any more was tried and is not an option either: in the full GPU suite (three builds of the library in one process) it ended
// in a memory access fault of the device.  A process that factorises camera systems AND pose graphs keeps both pairs and pays that 10 %.
static void prewarm_chol_dataflow(int) {
  prewarm_kernels({(const void*)k_df_bulk, (const void*)k_df_chain, (const void*)k_df_single, (const void*)k_df_begin, (const void*)k_df_resolve});
}
static PrewarmUnit prewarm_chol_dataflow_registered(prewarm_chol_dataflow);

}  // namespace gt
#else
}  // namespace gt   (emulation: the device code above only)
#endif   // GT_KERNEL_EMU
