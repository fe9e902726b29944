// chol_device.h -- device code shared by the two schedules of the reduced-system Cholesky (cholesky.hip: stream / event
// schedule of per-column launches; chol_dataflow.hip: persistent dataflow kernels): MFMA helpers, the streamed 32-column
// panel (chain wavefront + followers) and the diagonal-tile body built from it.
#pragma once
// (GT_KERNEL_EMU: tools/kernel_emu compiles this header for the host -- one thread per work-item -- and supplies the device vocabulary,
// the constants of context.h it needs and the three macros below itself)
#ifndef GT_KERNEL_EMU
#include "kernels.h"
#define GT_PIN(x) asm volatile("" : "+v"(x))                           // keeps a value in its register at this point of the schedule
#define GT_DRAIN_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")   // this wavefront's stores are acknowledged
#define GT_LDS_VOLATILE(T) __attribute__((address_space(3))) volatile T*    // explicit LDS pointers for volatile accesses
#define GT_XCC_ID(x) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x))   // where a workgroup runs (traces, post-mortems)
#define GT_HW_ID(x) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x))
#define GT_WAVE_SYNC() ((void)0)   // lanes of ONE wavefront exchange data through LDS here: in order on the hardware (the LDS operations of a
                                   // wavefront complete in program order), a rendezvous of the wavefront's threads in the host emulation
#endif

namespace gt {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int T = kTile;        // 128
constexpr int TT = kTileDoubles;  // doubles of one tile slot (context.h::SMat)
constexpr int SB = 32;          // sub-block of the diagonal tile
constexpr int P = T + 2;        // LDS pitch of a full tile: (2*P) % 64 == 4 -> MFMA operand reads conflict-free
constexpr int PB = SB + 2;      // LDS pitch inside a 32x32 sub-block of the packed diagonal tile
// The diagonal tile lives in LDS as its 10 lower 32x32 sub-blocks (87 KB instead of 133 KB) so that k_potrf128 can
// share a CU with a k_syrk workgroup of the overlapped trailing update (look-ahead).
__device__ __forceinline__ constexpr int boff(int ib, int cb) { return (ib * (ib + 1) / 2 + cb) * SB * PB; }

__device__ __forceinline__ double readlane_f64(double v, int l) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
  return u.d;
}

// MFMA fragment helpers (cdna_hip_programming.md section 3, f64 16x16x4): lane l supplies
// A[row = l&15][k = l>>4], B[k = l>>4][col = l&15]; accumulator reg r holds C[row = (l>>4) + 4r][col = l&15].
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// 128x128 tile (global, ld = NP) <-> LDS (pitch PL doubles); 16-byte accesses, 16 loads in flight per lane
template <int PL>
__device__ __forceinline__ void tile_to_lds(const double* __restrict__ tile, double* __restrict__ L, int tid) {
#pragma unroll
  for (int e0 = 0; e0 < T * (T / 2); e0 += 256 * 16) {
    double2 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int e = e0 + u * 256 + tid;
      v[u] = *reinterpret_cast<const double2*>(tile + (e / (T / 2)) * T + 2 * (e % (T / 2)));
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int e = e0 + u * 256 + tid;
      double* d = L + (e / (T / 2)) * PL + 2 * (e % (T / 2));
      d[0] = v[u].x; d[1] = v[u].y;
    }
  }
}

// 1/sqrt(p) to full double precision: v_rsq_f64 seed + two Newton steps (no IEEE sqrt/div sequence on
// the 128-pivot critical path).  p <= 0 or NaN propagates inf/NaN and the caller raises the fail flag.
__device__ __forceinline__ double rsqrt_nr(double p) {
  double r = __builtin_amdgcn_rsq(p);
  const double h = 0.5 * p;
  r = r * __builtin_fma(-h, r * r, 1.5);
  r = r * __builtin_fma(-h, r * r, 1.5);
  return r;
}

// Publication mode of the diagonal-tile body.  false: plain stores + release fence per panel (the stream / event schedule:
// its TRSM workgroups acquire).  true: every datum another workgroup reads is stored write-through (sc1) and the panel's
// progress word follows the workgroup barrier as a relaxed store -- no cache-wide write-back on the serial chain (the fence
// cost 3-4 us per panel, 14 us per tile, next to 60 workgroups per XCD producing dirty lines); see chol_dataflow.hip.
__device__ __forceinline__ void st_pub(double* p, double v, bool wt) {
  if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}

// ---- cross-half helpers (gfx950 v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second)
// value of the same lane (mod 32) of half H (0: lanes 0-31, 1: lanes 32-63), delivered to both halves
template <int H>
__device__ __forceinline__ double half_bcast(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[H], (int)a[H]);
}
// v(lane) + v(lane ^ 32), identical in both halves
__device__ __forceinline__ double half_sum(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// MFMA-operand image of a 32x32 block M as k_trsm128 consumes it (B operand of X * M^T, k split as 8 lk + s):
// half tj, lane = 16 lk + lr keeps M[16 tj + lr][8 lk + 0..7] in 8 consecutive doubles, so a wavefront fetches its
// operands of a block with four fully coalesced 16-byte loads per lane.
__device__ __forceinline__ int opnd_off(int blk, int row, int col) {
  return ((blk * 2 + (row >> 4)) * 64 + 16 * (col >> 3) + (row & 15)) * 8 + (col & 7);
}
constexpr int kOpndBase = 4 * SB * SB;   // doubles: the operand images follow the four plain inverses in the Xinv slot

// 1/p to full double precision: v_rcp_f64 seed + two Newton steps (4 dependent FMAs: the shortest way from a pivot
// to the multiplier of its rank-1 update)
__device__ __forceinline__ double rcp_nr(double p) {
  double x = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, x, 1.0);
  x = __builtin_fma(x, e, x);
  e = __builtin_fma(-p, x, 1.0);
  x = __builtin_fma(x, e, x);
  return x;
}

// ---- the 32-column panel of the diagonal tile, streamed --------------------------------------------------------
// Register layout (every wavefront of the panel): lane l keeps row i = l & 31 of its 32x32 block, the columns of
// parity h = l >> 5 (a[cl] = A[i][2 cl + h]): a rank-1 step costs at most 16 FMAs per lane and all 64 lanes work.
//
// Wavefront 0 owns the diagonal block and walks the pivots (PotrfStep).  For pivot J it needs 1/piv (v_readlane +
// rcp_nr: the serial chain, ~100 cycles) to finish column J+1, whose entries it PUBLISHES unscaled as line J+1
// [stored in (parity, index) order pos(c) = 16 (c & 1) + c / 2 so that a half reads its 16 values as 16-byte
// broadcasts], then it applies the rest of the rank-1 update, computes r = 1/sqrt(piv) off the chain, scales column
// J, stores dinv[J] = r and bumps the progress counter.  All 32 lines stay in LDS.
//
// Every other 32-row block below (rows of L(ib,jb), i.e. the TRSM X L^T = A) and the rows of the identity (which
// turn into L^-T, i.e. the inverse needed by k_trsm128 / k_bwd_diag) are FOLLOWERS (FollowGroup): they replay the same
// elimination on their own rows from the published (line J, r_J), a few pivots behind wavefront 0, and are done a
// few hundred cycles after it.  They never feed back into the chain, so the chain wavefront never waits.
// The step index is a template parameter: every register index is static.  The chain wavefront's code is branch
// free (selects, trash address for the non-owner half) so that the scheduler can overlap the chain with the updates.
constexpr int kLineTrash = SB * SB;   // doubles: lines[32][32], then 64 trash slots
// LDS of the diagonal-tile body in doubles: the 10 packed sub-blocks, 1/pivot [T], 1/sqrt(pivot) [T], the published columns + trash, two
// progress words
constexpr int kPotrfSmemDoubles = 10 * SB * PB + 2 * T + SB * SB + 64 + 2;
// explicit LDS pointers for the volatile accesses (address-space inference leaves volatile accesses as flat_*)
typedef GT_LDS_VOLATILE(double) lds_vdouble_p;
typedef GT_LDS_VOLATILE(int) lds_vint_p;

template <int J>
struct PotrfStep {
  static __device__ __forceinline__ void run(double (&a)[16], const double (&cj)[16], double own, double* lines,
                                             lds_vdouble_p rinvs, lds_vint_p prog, int progbase, int lane, int i, int h) {
    constexpr int hJ = J & 1, cJ = J >> 1;
    // ---- the serial chain: pivot -> 1/pivot -> column J+1 finished in its owner half -> next pivot.  The pivot and
    // A[J+1][J] travel by v_readlane; the lane's own A[i][J] (`own`) was read back from line J one step ago, and that
    // LDS round trip (~130 cycles) is shorter than the ~250 cycles of instruction issue of a step (measured with
    // tools/potrf_chain_probe.hip: the bare readlane-rcp-fma chain is 90 cycles, a full step 250-300).
    const double piv = readlane_f64(a[cJ], J + 32 * hJ);
    const double rinv = rcp_nr(piv);
    double cn[16], ownN = 0.0;
    if constexpr (J + 1 < SB) {
      constexpr int hN = (J + 1) & 1, cN = (J + 1) >> 1;
      const double s1 = readlane_f64(a[cJ], J + 1 + 32 * hJ);   // A[J+1][J]
      a[cN] = __builtin_fma(-((h == hN) ? own * s1 : 0.0), rinv, a[cN]);
      // publish column J+1 (unscaled) and fetch it back for the rest of step J+1's update
      double* line = lines + (J + 1) * SB;
      const int pos = 16 * (i & 1) + (i >> 1);
      *(lds_vdouble_p)((h == hN) ? line + pos : lines + kLineTrash + lane) = a[cN];
      GT_WAVE_SYNC();
#pragma unroll
      for (int cl = cN; cl < 16; cl++) cn[cl] = line[16 * h + cl];
      ownN = line[pos];               // A[i][J+1], written by the lane of row i in the owner half
    }
    // software-pipeline cut: nothing moves across, so a scheduling region = [rest of update J] + [chain of J+1]:
    // the chain overlaps the updates and register pressure stays bounded
    __builtin_amdgcn_sched_barrier(0);
    // ---- rest of the rank-1 update, fed by line J (read one step ago): off the chain
    const double u = own * rinv;
    if constexpr ((J & 1) == 1 && J + 1 < SB)   // J odd: the odd half's a[cJ+1] is column J+2
      a[cJ + 1] = __builtin_fma(-((h == 1) ? u : 0.0), cj[cJ + 1], a[cJ + 1]);
    constexpr int c0 = (J & 1) ? cJ + 2 : cJ + 1;
#pragma unroll
    for (int cl = c0; cl < 16; cl++) a[cl] = __builtin_fma(-u, cj[cl], a[cl]);
    // pin the row: keeps hipcc from deferring these updates across many steps (every step's broadcast values alive)
#pragma unroll
    for (int cl = cJ + 1; cl < 16; cl++) GT_PIN(a[cl]);
    rinvs[J] = rinv;                // every lane stores the same value: no exec games on the chain wavefront
    if constexpr ((J & 3) == 3) *prog = progbase + J + 1;   // LDS operations of one wavefront complete in order
    PotrfStep<J + 1>::run(a, cn, ownN, lines, rinvs, prog, progbase, lane, i, h);
  }
};
template <>
struct PotrfStep<SB> {
  static __device__ __forceinline__ void run(double (&)[16], const double (&)[16], double, double*, lds_vdouble_p, lds_vint_p,
                                             int, int, int, int) {}
};

// (A WINDOWED pivot chain -- the chain wavefront keeps 8 columns, a helper wavefront on another SIMD applies every pivot to the columns
// beyond the window and hands the next window over through LDS -- was built at the end of round 4 (29.5 instead of 34 instructions per
// step, no spills, bit-identical) and measured in round 5: 5.24 - 5.27 ms against 5.12 - 5.14 on the L1723 factorisation; the three
// hand-overs per panel cost more than the shorter steps save.  Removed; profiles/r05a_variants_ab.txt.)

// The columns stay UNSCALED through the elimination (only 1/pivot is on the chain).  At the end of a panel the chain
// wavefront computes the 32 values r_c = 1/sqrt(pivot_c) = sqrt(1/pivot_c) in parallel (lane c), publishes them in
// (parity, index) order and raises rready; every wavefront then scales its columns.
__device__ __forceinline__ void scale_columns(double (&a)[16], const double* rs, int h) {
#pragma unroll
  for (int cl = 0; cl < 16; cl++) a[cl] *= rs[16 * h + cl];
}

// factor the diagonal sub-block jb in place (upper part zeroed)
// pk / prev_exp: the reference's rank test (gtsam/base/cholesky.cpp:144-157, underconstrainedExponentDifference = 12) at the end of
// every variable's block of pivots: pk[c] = 1 -- column c is the last pivot of a variable of dimension >= 2: fail when the binary
// exponent of R(c-1,c-1) exceeds that of R(c,c) by 12 or more; pk[c] = 2 -- a one-dimensional variable: fail unless the exponent
// of R(c,c) is > -12; 0 -- no test (inside a variable, padding).  prev_exp carries the exponent of the pivot before this panel.
__device__ __forceinline__ void stage_potrf(double* __restrict__ A, double* __restrict__ rinvs, double* __restrict__ rs,
                                            double* __restrict__ lines, int* prog, int jb, int lane, double* fail,
                                            const unsigned char* __restrict__ pk, int& prev_exp) {
  const int i = lane & 31, h = lane >> 5;
  double* row = A + boff(jb, jb) + i * PB;
  // the pivot kinds of this panel's columns (rank test below): requested NOW, so that the global load's latency lies under the 32 pivots
  // (left where it is used -- after them, in front of the barrier that releases the panel -- it cost a memory round trip per panel
  // on the wavefront everybody waits for; the scheduling barriers inside the steps keep the load up here)
  unsigned kind = 0;
  if (pk) kind = pk[SB * jb + i];
  double a[16], c0[16];
#pragma unroll
  for (int cl = 0; cl < 16; cl++) a[cl] = row[2 * cl + h];
  const int pos = 16 * (i & 1) + (i >> 1);
  *(lds_vdouble_p)((h == 0) ? lines + pos : lines + kLineTrash + lane) = a[0];
  GT_WAVE_SYNC();
#pragma unroll
  for (int cl = 0; cl < 16; cl++) c0[cl] = lines[16 * h + cl];
  const double own0 = lines[pos];
  PotrfStep<0>::run(a, c0, own0, lines, (lds_vdouble_p)(rinvs + SB * jb), (lds_vint_p)prog, SB * jb, lane, i, h);
  GT_WAVE_SYNC();
  const double rv = rinvs[SB * jb + i];
  // Eigen LLT: non-positive pivot -> NumericalIssue (NaN compares false too); checked once per panel, off the chain
  if (__builtin_amdgcn_ballot_w64(!(rv > 0.0 && rv < __builtin_inf())) != 0 && lane == 0) *fail = 1.0;
  *(lds_vdouble_p)(rs + SB * jb + pos) = rv * rsqrt_nr(rv);
  GT_WAVE_SYNC();
  *(lds_vint_p)(prog + 1) = jb + 1;
  scale_columns(a, rs + SB * jb, h);
#pragma unroll
  for (int cl = 0; cl < 16; cl++) row[2 * cl + h] = (2 * cl + h <= i) ? a[cl] : 0.0;
  if (pk) {   // off the pivot chain: the panel is out, the followers are running
    const double Rii = rsqrt_nr(rv);   // sqrt(pivot) = the diagonal entry of the factor
    const int ex = (int)((__double_as_longlong(Rii) >> 52) & 0x7ff) - 1022;   // frexp exponent
    int before = __builtin_amdgcn_ds_bpermute(((lane - 1) & 63) << 2, ex);
    if (i == 0) before = prev_exp;
    const bool bad = (kind == 1u && before - ex >= 12) || (kind == 2u && !(ex > -12));
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) *fail = 1.0;
    prev_exp = __builtin_amdgcn_readlane(ex, 31);
  }
}

#ifndef GT_KERNEL_EMU
#define GT_STAMP_CLOCK() wall_clock64()
#else
#define GT_STAMP_CLOCK() __builtin_amdgcn_s_memtime()
#endif
// The replay runs four pivots at a time (the chain wavefront publishes its progress every fourth pivot) with the LDS operands of a step
// -- its line and 1/pivot -- requested two steps ahead into two register buffers; scheduling barriers keep the requests where they are.
// (Rounds 1 - 5 replayed step by step and left the order to the compiler, which put the eight reads of a step's line right in front of its
// 16 FMAs: one LDS round trip exposed per pivot, ~350 cycles per pivot against the chain's ~265 -- the followers ended 1.2 - 1.7 us
// after the chain wavefront, the inverse last.  Now they end 0.1 - 0.4 us after it: profiles/r06q_potrf_followers_ab.txt.)
template <int J>
__device__ __forceinline__ void follow_load(double (&ln)[16], double& rv, const double* lines, const double* rinvs, int h) {
  constexpr int lo = (J >> 1) & ~1;   // first column pair the step reads, 16-byte aligned
  rv = rinvs[J];
#pragma unroll
  for (int cl = lo; cl < 16; cl++) ln[cl] = lines[J * SB + 16 * h + cl];
}
template <int J>
__device__ __forceinline__ void follow_apply(double (&a)[16], const double (&ln)[16], double rv, int h) {
  constexpr int hJ = J & 1, cJ = J >> 1;
  const double own = half_bcast<hJ>(a[cJ]);
  const double u = own * rv;
  if constexpr (J + 1 < SB && ((J + 1) & 1) == 1)
    a[cJ] = __builtin_fma(-((h == 1) ? u : 0.0), ln[cJ], a[cJ]);   // column J+1 lives in the odd half's a[cJ]
#pragma unroll
  for (int cl = cJ + 1; cl < 16; cl++) a[cl] = __builtin_fma(-u, ln[cl], a[cl]);
#pragma unroll
  for (int cl = cJ; cl < 16; cl++) GT_PIN(a[cl]);
}
template <int J0>
struct FollowGroup {
  static __device__ __forceinline__ void run(double (&a)[16], const double* lines, const double* rinvs,
                                             lds_vint_p prog, int progbase, int h) {
    while (*prog < progbase + J0 + 4) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double l0[16], l1[16], r0, r1;   // two line buffers: the operands of step s+2 are requested into the registers of step s as soon as it is applied
    follow_load<J0>(l0, r0, lines, rinvs, h);
    follow_load<J0 + 1>(l1, r1, lines, rinvs, h);
    __builtin_amdgcn_sched_barrier(0);
    follow_apply<J0>(a, l0, r0, h);
    __builtin_amdgcn_sched_barrier(0);
    follow_load<J0 + 2>(l0, r0, lines, rinvs, h);
    __builtin_amdgcn_sched_barrier(0);
    follow_apply<J0 + 1>(a, l1, r1, h);
    __builtin_amdgcn_sched_barrier(0);
    follow_load<J0 + 3>(l1, r1, lines, rinvs, h);
    __builtin_amdgcn_sched_barrier(0);
    follow_apply<J0 + 2>(a, l0, r0, h);
    follow_apply<J0 + 3>(a, l1, r1, h);
    __builtin_amdgcn_sched_barrier(0);
    FollowGroup<J0 + 4>::run(a, lines, rinvs, prog, progbase, h);
  }
};
template <>
struct FollowGroup<SB> {
  static __device__ __forceinline__ void run(double (&)[16], const double*, const double*, lds_vint_p, int, int) {}
};

// follower of the diagonal sub-block jb: ib > jb -> the rows of A(ib,jb) become L(ib,jb) (in LDS);
// ib < 0 -> the rows of the identity become L(jb,jb)^-T: lane i ends up with column i of the inverse, written
// plain (Xout, row-major) and as the MFMA operand image (Xop)
__device__ __forceinline__ void stage_follow(double* A, const double* lines, const double* rinvs, const double* rs,
                                             const int* prog, int jb, int ib, int lane, double* Xout, double* Xop, bool wt = false, long long* fdbg = nullptr) {
#define FSTAMP(n) do { if (fdbg && lane == 0) fdbg[n] = (long long)GT_STAMP_CLOCK(); } while (0)
  const int i = lane & 31, h = lane >> 5;
  const bool inv = ib < 0;
  double* R = A + boff(inv ? jb : ib, jb) + i * PB;
  double a[16];
  if (inv) {
    // (the row index goes through GT_PIN: left visible, the 16 constants are hoisted out of the panel loop of potrf_body, spilled, and
    // reloaded here one scratch round trip at a time)
    int iv = i;
    GT_PIN(iv);
#pragma unroll
    for (int cl = 0; cl < 16; cl++) a[cl] = (2 * cl + h == iv) ? 1.0 : 0.0;
  } else {
#pragma unroll
    for (int cl = 0; cl < 16; cl++) a[cl] = R[2 * cl + h];
  }
  FSTAMP(0);
  FollowGroup<0>::run(a, lines, rinvs + SB * jb, (lds_vint_p)prog, SB * jb, h);
  FSTAMP(1);
  while (*(lds_vint_p)(prog + 1) < jb + 1) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  FSTAMP(2);
  scale_columns(a, rs + SB * jb, h);
  if (inv) {
    // two base addresses + compile-time offsets (row 2 cl + h of the inverse: opnd_off(6 + jb, 2 cl + h, i) = its value at cl = 0 plus
    // 512 (cl >> 3) + 16 (cl & 7) doubles).  With the offsets left to the compiler it kept 16 address registers, spilled six of them
    // and reloaded each one behind an s_waitcnt vmcnt(0) -- which also waits for the write-through store before it: six store round
    // trips in a row on the wavefront that every panel's release waits for.
    double* xo = Xout + h * SB + i;
    double* xp = Xop + opnd_off(6 + jb, h, i);
#pragma unroll
    for (int cl = 0; cl < 16; cl++) {
      xo[cl * 2 * SB] = a[cl];   // plain copy of the inverse: read by later kernels only (backward solve)
      st_pub(xp + (cl >> 3) * 512 + (cl & 7) * 16, a[cl], wt);
    }
  } else {
#pragma unroll
    for (int cl = 0; cl < 16; cl++) R[2 * cl + h] = a[cl];
  }  FSTAMP(3);
#undef FSTAMP
}

// Two 16x16 MFMA tiles at a time:  C_i -= A_i B_i^T  (A_i, B_i: 16 x 32 patches of LDS at pitch PB; C_i 16 x 16; i = 0 and, if `two`, 1).
// Every LDS operand of both tiles is requested before the first MFMA and the two accumulation chains alternate on the matrix pipe: one
// tile at a time, with the reads left to the compiler (pairs of reads between pairs of dependent MFMAs), a task of 8 MFMAs took 1 000 -
// 1 300 cycles -- half of the pipe idle with two wavefronts per SIMD (profiles/r06v_slice_tasks.txt).  Per tile the order of the
// accumulation is what it was: bit-identical.
// An 8-byte LDS read that stays one: hipcc pairs neighbouring 8-byte reads into ds_read2_b64 / ds_read2st64_b64, which the LDS serves in its
// 32-bank mode at half the rate -- and the pitches here (PB, the swizzle of the bulk kernel's chunks) are laid out for the 64-bank mode of
// ds_read_b64: in pairs the operand reads of an MFMA tile were 2-way conflicts on top (16 LDS cycles per pair against 4 for two single
// reads; MI355X_MICROARCH.md, LDS table).  A volatile access is not merged.
__device__ __forceinline__ double lds_ld(const double* p) { return *(lds_vdouble_p)p; }
__device__ __forceinline__ void upd_tiles2(double* C0, const double* A0, const double* B0, double* C1, const double* A1, const double* B1,
                                           bool two, int lr, int lk) {
  v4f64 acc0, acc1 = {0.0, 0.0, 0.0, 0.0};
  double a0[8], b0[8], a1[8], b1[8];
#pragma unroll
  for (int r = 0; r < 4; r++) acc0[r] = lds_ld(C0 + (lk + 4 * r) * PB + lr);
#pragma unroll
  for (int s = 0; s < 8; s++) { a0[s] = -lds_ld(A0 + lr * PB + 4 * s + lk); b0[s] = lds_ld(B0 + lr * PB + 4 * s + lk); }
  if (two) {
#pragma unroll
    for (int r = 0; r < 4; r++) acc1[r] = lds_ld(C1 + (lk + 4 * r) * PB + lr);
#pragma unroll
    for (int s = 0; s < 8; s++) { a1[s] = -lds_ld(A1 + lr * PB + 4 * s + lk); b1[s] = lds_ld(B1 + lr * PB + 4 * s + lk); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; s++) { acc0 = MFMA(a0[s], b0[s], acc0); acc1 = MFMA(a1[s], b1[s], acc1); }
#pragma unroll
    for (int r = 0; r < 4; r++) { C0[(lk + 4 * r) * PB + lr] = acc0[r]; C1[(lk + 4 * r) * PB + lr] = acc1[r]; }
  } else {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; s++) acc0 = MFMA(a0[s], b0[s], acc0);
#pragma unroll
    for (int r = 0; r < 4; r++) C0[(lk + 4 * r) * PB + lr] = acc0[r];
  }
}
// A whole 32 x 32 block:  C -= A B^T  with A, B 32 x 32 patches (2 x 2 MFMA tiles, four accumulation chains, every operand row read once:
// 40 LDS reads for 32 MFMAs, against 40 for 16 with two unrelated tiles).  Per tile the order of the accumulation is unchanged.
__device__ __forceinline__ void upd_block4(double* C, const double* A, const double* B, int lr, int lk) {
  v4f64 acc[2][2];
  double a[2][8], b[2][8];
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[ti][tj][r] = lds_ld(C + (16 * ti + lk + 4 * r) * PB + 16 * tj + lr);
#pragma unroll
  for (int tt = 0; tt < 2; tt++)
#pragma unroll
    for (int s = 0; s < 8; s++) { a[tt][s] = -lds_ld(A + (16 * tt + lr) * PB + 4 * s + lk); b[tt][s] = lds_ld(B + (16 * tt + lr) * PB + 4 * s + lk); }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 8; s++)
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
      for (int tj = 0; tj < 2; tj++) acc[ti][tj] = MFMA(a[ti][s], b[tj][s], acc[ti][tj]);
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++)
#pragma unroll
      for (int r = 0; r < 4; r++) C[(16 * ti + lk + 4 * r) * PB + 16 * tj + lr] = acc[ti][tj][r];
}
// the (ti, tj) tile of the update A(ib,cb) -= L(ib,jb) L(cb,jb)^T inside the diagonal tile: its three patches
struct TilePatch { double* C; const double* A; const double* B; };
__device__ __forceinline__ TilePatch tile_patch(double* A, int jb, int ib, int cb, int ti, int tj) {
  return {A + boff(ib, cb) + (16 * ti) * PB + 16 * tj, A + boff(ib, jb) + (16 * ti) * PB, A + boff(cb, jb) + (16 * tj) * PB};
}
// ... and of  C(ib,cb) -= X(ib) X(cb)^T,  X = four 32 x 32 blocks
__device__ __forceinline__ TilePatch slice_patch(double* A, const double* X, int ib, int cb, int ti, int tj) {
  return {A + boff(ib, cb) + (16 * ti) * PB + 16 * tj, X + ib * SB * PB + (16 * ti) * PB, X + cb * SB * PB + (16 * tj) * PB};
}
// the lower 32 x 32 blocks of the packed tile in row-major order: block number (0..9) -> (ib, cb)
__device__ __forceinline__ void lower_block(int blk, int& ib, int& cb) {
  ib = 0; cb = blk;
  while (cb > ib) { cb -= ib + 1; ib++; }
}

// write the finished sub-blocks (ib, jb), ib = jb..3, back to the tile (diagonal one with its upper part zeroed; the
// strictly-upper sub-blocks of the tile are never read by anyone) and, for ib > jb, as operand images for k_trsm128
__device__ __forceinline__ void store_column(const double* A, double* tile, double* Xinv, int jb, int t, int nthreads, bool wt = false) {
  // Block by block (the block index is uniform: what depends on the lane is the 16-byte piece w = 16 r + c / 2 of a 32 x 32 block only),
  // up to three pieces per lane and block with their LDS reads in flight together.  (One flat loop over all pieces, decoding block, row
  // and column per piece, cost ~70 VALU instructions and an exposed LDS round trip per piece: 2.6 us for panel 0's write-back on three
  // wavefronts that share their SIMDs with the followers -- what panel 1's stage waited for.)
  for (int ib = jb; ib < 4; ib++) {
    const double* blk = A + boff(ib, jb);
    double* trow = tile + (SB * ib) * T + SB * jb;
    double* img = Xinv + kOpndBase + opnd_off(ib * (ib - 1) / 2 + jb, 0, 0);
    const bool diag = ib == jb;
    for (int w0 = t; w0 < 512; w0 += 3 * nthreads) {
      double2 v[3];
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int w = w0 + u * nthreads, r = w >> 4, c = 2 * (w & 15);
        v[u].x = 0.0; v[u].y = 0.0;
        if (w < 512) {
          if (!diag || c <= r) v[u].x = blk[r * PB + c];
          if (!diag || c + 1 <= r) v[u].y = blk[r * PB + c + 1];
        }
      }
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int w = w0 + u * nthreads, r = w >> 4, c = 2 * (w & 15);
        if (w < 512) {
          *reinterpret_cast<double2*>(trow + r * T + c) = v[u];   // (the tile itself is read by later kernels only)
          if (!diag) {
            double* o = img + opnd_off(0, r, c);
            if (wt) { st_pub(o, v[u].x, true); st_pub(o + 1, v[u].y, true); } else *reinterpret_cast<double2*>(o) = v[u];
          }
        }
      }
    }
  }
}

// ---- diagonal tile ------------------------------------------------------------------------------------------
// One workgroup of 8 wavefronts; per 32-column panel jb of the tile:
//   P1  wave 0: potrf32(jb), the pivot chain | waves 1..3-jb: followers of the row blocks below | wave 5: follower
//       producing the inverse | wave 4 idle (same SIMD as wave 0) | the remaining waves: everything of panel jb-1
//       that nobody is waiting for (the off-chain MFMA updates and the write-back of its finished blocks)
//   P3  the update of the NEXT panel (diagonal block + the blocks below it), all waves.
#ifndef GT_KERNEL_EMU
#define STAMP(n) do { if (dbg && threadIdx.x == 0) dbg[n] = (long long)wall_clock64(); } while (0)   // 100 MHz, like the other stamps of GTG_DF_TRACE
#else
#define STAMP(n) do { if (dbg && threadIdx.x == 0) dbg[n] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#endif
constexpr int kFlagOff = kOpndBase + 10 * 2 * 64 * 8;   // doubles: progress word of the tile, after the operand images
// the lower 32x32 sub-blocks of a diagonal tile -> the packed LDS image: 10 blocks x 512 16-byte pieces, 10 per lane, all
// loads in flight before the writes (512 threads)
// SC1: the tile was handed over by another workgroup of a running kernel (write-through stores + flag, chol_dataflow.hip): read it
// with agent-scope (sc1) loads, which are served past this CU's L1 -- the consumer half of the hand-off (8-byte accesses: the widest
// a relaxed agent-scope load lowers to)
template <bool SC1 = false>
__device__ __forceinline__ void diag_tile_to_lds(const double* tile, double* __restrict__ A, int tid) {
  double2 v[10];
#pragma unroll
  for (int u = 0; u < 10; u++) {
    const int e = u * 512 + tid, blk = e >> 9, w = e & 511;
    int ib = 0, rem = blk;
    while (rem > ib) { rem -= ib + 1; ib++; }
    const double* src = tile + (SB * ib + (w >> 4)) * T + SB * rem + 2 * (w & 15);
    if constexpr (SC1) {
      v[u].x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v[u].y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      v[u] = *reinterpret_cast<const double2*>(src);
    }
  }
#pragma unroll
  for (int u = 0; u < 10; u++) {
    const int e = u * 512 + tid, blk = e >> 9, w = e & 511;
    double* d = A + blk * SB * PB + (w >> 4) * PB + 2 * (w & 15);
    d[0] = v[u].x; d[1] = v[u].y;
  }
}

__device__ __forceinline__ void potrf_body(char* smem_raw, double* __restrict__ tile, int k, double* __restrict__ Xinv,
                                           double* __restrict__ fail, long long* __restrict__ dbg,
                                           long long epoch, long long* __restrict__ pflag, long long pflag_shadow, bool preloaded = false, bool wt = false,
                                           const unsigned char* __restrict__ pivot_kind = nullptr, double* __restrict__ tile_exp = nullptr,
                                           const double* Xdef = nullptr, int xmask_lo = -1, int xmask_hi = -1) {
  // (Xdef: the dataflow chain kernel only -- the last 32-column slice of the tile left of this one, in LDS, of which the blocks (ib, cb),
  // cb >= 1, are still to be applied; see chain_loop.  nullptr: nothing deferred)
  const long long flagbase = epoch * 8;   // progress words are monotonic over factorisations: no reset.  The epoch is a kernel ARGUMENT
  // (host-counted): a word in device memory that every factorisation rewrites was read one factorisation stale by one of two
  // co-operating kernels under multi-handle contention (rate ~1e-3; tools/df_contention_diag.py, profiles/r03_df_contention.txt)
  double* A = reinterpret_cast<double*>(smem_raw);   // 10 packed lower sub-blocks [SB][PB]
  double* rinvs = A + 10 * SB * PB;                   // [T]  1 / pivot
  double* rs = rinvs + T;                             // [T]  1 / sqrt(pivot), (parity, index) order inside a panel
  double* lines = rs + T;                             // [SB][SB] published columns of the current panel + 64 trash
  int* prog = reinterpret_cast<int*>(lines + kLineTrash + 64);   // [0] pivots published so far (monotonic over the tile), [1] panels whose rs are published
  // (the thread index is laundered per call: what is derived from it is recomputed for every tile instead of being hoisted out of the chain
  // kernel's tile loop, spilled -- the kernel uses all 256 registers -- and reloaded here one scratch round trip after the other)
  int tid_ = threadIdx.x;
  GT_PIN(tid_);
  const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: the roles below are scalar branches)
  const int wave_hw = wave; (void)wave_hw;
  const int lr = lane & 15, lk = lane >> 4;
  // critical-path kernel: win issue arbitration against co-resident k_syrk waves, and the chain wavefront against
  // its own followers
  if (wave == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
  if (tid == 0) { *(lds_vint_p)prog = 0; *(lds_vint_p)(prog + 1) = 0; }   // (volatile: as one 8-byte store of a hoisted zero pair the compiler spilled the pair and reloaded it here, a scratch round trip per tile)
  STAMP(0);
  if (!preloaded) diag_tile_to_lds(tile, A, tid);   // (preloaded: the caller filled the image and synchronises below)
  __syncthreads();
  STAMP(1);
  // rank test at the variables' block ends (see stage_potrf): pivot kinds of this tile's columns, exponent of the pivot before it
  const unsigned char* pk = pivot_kind ? pivot_kind + (size_t)k * T : nullptr;
  int prev_exp = 0;
  if (pk && tile_exp && k > 0 && wave == 0)
    prev_exp = (int)__double_as_longlong(__hip_atomic_load(tile_exp + k - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll 1
  for (int jb = 0; jb < 4; jb++) {
    const int nfol = 3 - jb;   // row blocks below
    // from panel 1 on the inverse follower runs on wavefront 7 (SIMD 3, whose other wavefront -- 3 -- is no follower any more) instead of
    // wavefront 5, which shares SIMD 1 with the follower wavefront 1
    const int wave = (jb >= 1 && wave_hw == 5) ? 7 : (jb >= 1 && wave_hw == 7) ? 5 : wave_hw;
    if (wave == 0) {
      stage_potrf(A, rinvs, rs, lines, prog, jb, lane, fail, pk, prev_exp);
      if (jb == 3 && tile_exp && lane == 0) st_pub(tile_exp + k, __longlong_as_double((long long)prev_exp), true);
    } else if (wave == 4) {
      // wavefront 4 shares SIMD 0 with the chain wavefront (wave id mod 4), which is issue bound: no arithmetic here (a third of panel 0's
      // deferred slice tasks, MFMAs, was tried in round 6: the chain wavefront's 32 pivots then take 6.2 us instead of 3.7), only its share of
      // the previous panel's write-back
      if (jb > 0) { __builtin_amdgcn_s_setprio(1); store_column(A, tile, Xinv, jb - 1, (2 + jb) * 64 + lane, (3 + jb) * 64, wt); }
    } else if (wave <= nfol || wave == 5) {
      stage_follow(A, lines, rinvs, rs, prog, jb, wave == 5 ? -1 : jb + wave, lane, Xinv + (int64_t)jb * SB * SB, Xinv + kOpndBase, wt,
                   dbg && jb == 2 && (wave_hw == 1 || wave_hw == 5 || wave_hw == 7) ? dbg + 48 + 4 * (wave_hw == 1 ? 0 : wave_hw == 5 ? 1 : 2) : nullptr);
    } else if (jb == 0) {
      __builtin_amdgcn_s_setprio(1);   // deferred work yields issue slots to the followers it shares a SIMD with
      // (the wavefronts 6 and 7 have nothing of their own to do in panel 0: the part of the last slice's update that neither panel 0 nor the
      // update of panel 1 reads -- the blocks (2,2), (3,1), (3,2), (3,3); the chain kernel passes the slice as Xdef and has applied it to the
      // blocks (ib, 0), (1,1), (2,1) itself, chol_dataflow.hip::chain_loop -- runs here, under panel 0's pivots)
      if (Xdef)
        for (int t = 8 + (wave - 6); t < 24; t += 4) {   // tile t: block t / 4 of (1,1) (2,1) (2,2) (3,1) (3,2) (3,3), MFMA tile t % 4; two at a time
          const int b2 = (t >> 2) * 2, u = t + 2, c2 = (u >> 2) * 2;
          const int ib0 = (0xFE9 >> b2) & 3, cb0 = (0xE65 >> b2) & 3, ib1 = (0xFE9 >> c2) & 3, cb1 = (0xE65 >> c2) & 3;
          const TilePatch p0 = slice_patch(A, Xdef, ib0, cb0, (t >> 1) & 1, t & 1);
          const TilePatch p1 = slice_patch(A, Xdef, ib1, cb1, (u >> 1) & 1, u & 1);
          // (xmask: the sub-tile mask of the tile the slice belongs to, chol_dataflow.hip::chain_loop -- a tile of the update whose two
          // operand patches share no live 16-column strip of the last slice, strips 6 and 7, is left out)
          const unsigned long long xm = (unsigned)xmask_lo | ((unsigned long long)(unsigned)xmask_hi << 32);
          const bool l0 = ((xm >> (8 * (2 * ib0 + ((t >> 1) & 1)) + 6)) & (xm >> (8 * (2 * cb0 + (t & 1)) + 6)) & 3ull) != 0;
          const bool l1 = ((xm >> (8 * (2 * ib1 + ((u >> 1) & 1)) + 6)) & (xm >> (8 * (2 * cb1 + (u & 1)) + 6)) & 3ull) != 0;
          if (l0 && l1) upd_tiles2(p0.C, p0.A, p0.B, p1.C, p1.A, p1.B, true, lr, lk);
          else if (l0) upd_tiles2(p0.C, p0.A, p0.B, p0.C, p0.A, p0.B, false, lr, lk);
          else if (l1) upd_tiles2(p1.C, p1.A, p1.B, p1.C, p1.A, p1.B, false, lr, lk);
        }
    } else if (jb > 0) {
      __builtin_amdgcn_s_setprio(1);   // (see above)
      const int pj = jb - 1;                            // deferred work of panel pj
      const int nh = 2 + jb, hw = wave >= 6 ? wave - 6 : 2 + (wave - nfol - 1);
      // updates of the blocks right of panel pj+1 (those of panel pj+1 itself were done in P3, before its followers
      // started): blocks (ib, cb), pj+2 <= cb <= ib
      const int nb = 3 - pj, ntask = (nb * (nb - 1) / 2) * 4;
      if (ntask == 4 * nh) {                             // (panel 1: three blocks for three wavefronts) a whole 32 x 32 block each
        int bi, rem;
        lower_block(hw, bi, rem);
        upd_block4(A + boff(pj + 2 + bi, pj + 2 + rem), A + boff(pj + 2 + bi, pj), A + boff(pj + 2 + rem, pj), lr, lk);
      } else
      for (int t = hw; t < ntask; t += 2 * nh) {         // two tiles at a time
        const int u = t + nh < ntask ? t + nh : t;
        int bi, rem, bi1, rem1;                          // (bi, rem): 0 <= rem <= bi < nb - 1
        lower_block(t >> 2, bi, rem); lower_block(u >> 2, bi1, rem1);
        const TilePatch p0 = tile_patch(A, pj, pj + 2 + bi, pj + 2 + rem, (t >> 1) & 1, t & 1);
        const TilePatch p1 = tile_patch(A, pj, pj + 2 + bi1, pj + 2 + rem1, (u >> 1) & 1, u & 1);
        upd_tiles2(p0.C, p0.A, p0.B, p1.C, p1.A, p1.B, u != t, lr, lk);
      }
      store_column(A, tile, Xinv, pj, hw * 64 + lane, (nh + 1) * 64, wt);   // (wavefront 4 is the last share)
    }
    // Panel jb is released to the workgroups waiting for it (the TRSM workgroups of this launch / the substitutions of the dataflow
    // schedule) once its inverse and every L(jb, q<jb) operand image are in memory.
    //   plain stores (wt = false): workgroup barrier, then ONE lane's agent-scope release store (L2 write-back) of the progress word;
    //   write-through stores (wt = true): EVERY storing wavefront waits for its own stores to be acknowledged (s_waitcnt vmcnt(0)),
    //   then the workgroup barrier, then one lane's relaxed store of the word.  The barrier alone is NOT enough: a workgroup-scope
    //   release does not wait for vmcnt on this target, and the word -- stored by wavefront 0 -- overtook the images of wavefront 5
    //   and of the deferred wavefronts under uneven memory load: a substitution then read the previous factorisation's inverse
    //   (rounds 1-3: rare last-digit differences with several handles on one device; cdna_hip_programming.md G16 pitfall 14).
    //   For the panels 0..2 the wait is taken AFTER the updates of the next panel (P3: LDS only), so the acknowledgements arrive
    //   under them and the pivot chain does not stall; the substitution step q only has to be done before panel q+1 is.  The last
    //   panel's word follows its drain directly (it is the one on the serial chain of the factorisation).
    if (wave_hw != 0) __builtin_amdgcn_s_setprio(2);
    const bool late = wt && jb < 3;
    if (dbg && lane == 0) dbg[16 + 8 * jb + wave_hw] = (long long)GT_STAMP_CLOCK();   // trace: when each wavefront was done with its part of the stage
    if (wt && !late) GT_DRAIN_STORES();
    __syncthreads();
    if (tid == 0 && !late)
    {
      if (wt) (void)__hip_atomic_exchange(pflag, flagbase + jb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (chol_dataflow.hip::st_flag)
      else __hip_atomic_store(pflag, flagbase + jb + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (pflag_shadow) __hip_atomic_store(pflag + pflag_shadow, flagbase + jb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // chol_dataflow.hip::st_flag
    }
    STAMP(2 + 3 * jb);
    STAMP(3 + 3 * jb);
    // P3: panel jb+1 (its diagonal block and the blocks below it) must be complete before its chain / followers start
    if (wave < 4 * (3 - jb)) {   // at most 12 tiles for 8 wavefronts: the first four take two
      const int t = wave, u = t + 8 < 4 * (3 - jb) ? t + 8 : t;
      const TilePatch p0 = tile_patch(A, jb, jb + 1 + (t >> 2), jb + 1, (t >> 1) & 1, t & 1);
      const TilePatch p1 = tile_patch(A, jb, jb + 1 + (u >> 2), jb + 1, (u >> 1) & 1, u & 1);
      upd_tiles2(p0.C, p0.A, p0.B, p1.C, p1.A, p1.B, u != t, lr, lk);
    }
    if (late) GT_DRAIN_STORES();
    __syncthreads();
    if (tid == 0 && late)
    {
      (void)__hip_atomic_exchange(pflag, flagbase + jb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (chol_dataflow.hip::st_flag)
      if (pflag_shadow) __hip_atomic_store(pflag + pflag_shadow, flagbase + jb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    STAMP(4 + 3 * jb);
  }
  // (the thread index is laundered: left visible, this block's LDS address is computed once per kernel, spilled, and reloaded here behind an
  // s_waitcnt vmcnt(0) that also waits for every store of the tile that is still in flight -- 0.4 us at the end of the serial chain's link)
  int tl = tid;
  GT_PIN(tl);
  store_column(A, tile, Xinv, 3, tl, 512, wt);
  STAMP(14);
}

}  // namespace gt
