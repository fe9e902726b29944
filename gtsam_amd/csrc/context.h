// context.h -- device-resident state of one LM problem (one handle = one GPU = one shard).
//
// HBM layout (all FP64 unless noted; see DESIGN.md "Data layout"):
//   values / trial          packed gtsam::Values, variable-id order (12 | 17 | 3 doubles per variable)
//   *_J                     per-factor whitened records [A1 | A2 | b], AoS, written once per linearize
//   Hd, gred0               reduced variables: d x d diagonal blocks (stride 81) and gradient (stride 9)
//   V, gp                   landmarks: 3x3 blocks (stride 9) and gradient (stride 3)
//   Hoff                    off-diagonal reduced blocks from BetweenFactors (stride 81)
//   E                       per landmark-observation  E = Jc^T Jp L^-T  (stride 27), rebuilt per lambda
//   S                       reduced system BY TILES: one contiguous 128 x 128 slot per stored tile of the lower triangle
//                           (plan.slot: tile -> slot) plus the tiles of one extra tile row whose first row carries the rhs (g^T -> y^T)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <exception>
#include <thread>
#include <vector>

#include "../../include/gtsam_amd.h"

namespace gt {

constexpr int kTile = 128;  // dense tile / panel width of the reduced-system Cholesky
constexpr int kTileDoubles = kTile * kTile;
constexpr int kSub = 16;    // granularity of the structural masks of a tile (one MFMA tile: 8 x 8 bits per 128 x 128 tile; DfPlan, analysis.hip)

// ---- storage of the reduced system: BY TILES -----------------------------------------------------------------------------
// Every 128 x 128 tile of the lower triangle that the symbolic factorisation says can become non-zero (fill included), and every
// tile of the rhs row (tile row nt: row 0 of those tiles carries the right-hand side through the factorisation), owns one
// contiguous slot of 128 x 128 doubles, row-major with pitch 128; slot[I * nt + J] is its index, -1 for a tile that is never
// touched (what the reference keeps per clique as SymmetricBlockMatrix blocks, base/SymmetricBlockMatrix.h:195-237).  Rounds 1-3
// kept the whole (NP + 128) x NP array: 1.9 GB for the 15 507 columns of the L1723 shape with 45 % of the tiles ever touched,
// 31 GB for a 20 000-pose graph with a few per cent.
struct SMat {
  double* p; const int32_t* slot; int nt;
  __host__ __device__ __forceinline__ double* tile(int I, int J) const { return p + (int64_t)slot[(int64_t)I * nt + J] * kTileDoubles; }
  // entry (r, c) of the matrix in scalar coordinates (r = 128 nt: the rhs row)
  __host__ __device__ __forceinline__ double* at(int64_t r, int64_t c) const { return tile((int)(r >> 7), (int)(c >> 7)) + ((r & 127) << 7) + (c & 127); }
  // the same for writers of whole variable blocks: a d x d block that straddles a tile boundary next to the diagonal has entries in a
  // tile of the UPPER triangle, which has no slot (and which nobody reads): nullptr there
  __host__ __device__ __forceinline__ double* at_stored(int64_t r, int64_t c) const {
    const int32_t q = slot[(r >> 7) * nt + (c >> 7)];
    return q < 0 ? nullptr : p + (int64_t)q * kTileDoubles + ((r & 127) << 7) + (c & 127);
  }
};


// scalar slots reduced on the device (doubles)
// SC_TIMEOUT directly follows SC_FAIL: the factorisation gets `scalars + SC_FAIL` and raises [0] for a non-positive pivot, [1] for a
// dependency wait that ran into its bound (a scheduling problem, reported as an error -- never as "not positive definite")
// SC_UNSUPPORTED: a smart factor met a case in which the reference throws out of linearize() / error(): 1 = Cal3Bundler::calibrate did
// not converge, kUnsupportedCheirality = CheiralityException (a failed track's point at infinity behind one of its cameras); summed over
// the shards, so the second code lies above any count of the first
constexpr double kUnsupportedCheirality = 1024.0;
enum { SC_ERROR = 0, SC_LIN0 = 1, SC_LIN1 = 2, SC_TRIAL_ERROR = 3, SC_DELTA_SQ = 4, SC_FAIL = 5, SC_TIMEOUT = 6, SC_UNSUPPORTED = 7, SC_COUNT = 8 };

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t cap = 0;                               // bytes of the allocation behind p (>= n elements: api.hip keeps and re-issues big blocks)
  void alloc(size_t count);
  void upload(const T* host, size_t count, hipStream_t s);
  void free();
};

// Tile-sparse schedule of the reduced-system Cholesky (built once per graph by the host symbolic analysis):
// which 128x128 tiles exist after fill-in, at the granularity of 256-wide column pairs.
constexpr int kEStride = 32;                    // doubles per E slot (one per observation): 9x3 block padded to 256 bytes = two 128-byte lines

struct CholPlan {
  int nt = 0;                                   // 128-tiles of the square part (the rhs tile is index nt)
  DevBuf<int32_t> rows, pairs, bcols;           // concatenated lists: TRSM row tiles, SYRK (I,J) pairs, backward column tiles
  DevBuf<int32_t> bwd_col_off, bwd_col_rows;    // the backward tiles by column: offsets (nt + 1), row tiles in descending order
  DevBuf<int32_t> stored;                       // every stored tile (I,J) incl. the rhs row, in SLOT order: tile q of this list lives in slot q of S
  int64_t n_stored = 0;
  DevBuf<int32_t> slot;                         // (nt + 1) x nt: tile (I, J) -> its slot in S, -1 = never touched (chol_device.h::SMat)
  std::vector<int32_t> h_slot;
  DevBuf<int32_t> exch;                         // the stored tiles that can be non-zero BEFORE the factorisation (no fill): what a multi-GPU
  int64_t n_exch = 0;                           // exchange of the partial reduced systems has to carry (L1723: 19 % fewer than `stored`)
  std::vector<int64_t> trsm_off, trsm_cnt;      // per column tile
  std::vector<int64_t> s1_off, s1_cnt, nar_off, nar_cnt, rest_off, rest_cnt;   // per pair: thin update, look-ahead part, rest
  std::vector<int64_t> bwd_off, bwd_cnt;        // per row tile
  // nested dissection: part of every 256-column pair (parts numbered in elimination order, children before parents),
  // parent part (-1: root); updates of a pair whose target columns lie in an ancestor part
  std::vector<int32_t> pair_part, part_parent;
  std::vector<int64_t> anc_off, anc_cnt;
  std::vector<uint8_t> h_fill;                 // the fill structure over column pairs ((np x np) bytes): what ensure_stream_lists builds the SYRK pair lists from
  bool stream_lists = false;                    // `pairs` and the per-pair offsets are built (on first use of the stream schedule: cholesky.hip::ensure_stream_lists)
  int critical_pairs = 0;                       // pairs on the longest leaf-to-root path (= all pairs without parts)
  double flops = 0.0;                           // algorithmic flops of one factorisation over the stored tiles
  double dense_fraction = 1.0;                  // stored lower tiles / all lower tiles
};


// Dataflow schedule of the same factorisation (chol_dataflow.hip): one task per stored 128x128 tile, left-looking, executed by
// persistent workgroups that take tasks in a fixed topological order; dependencies are epoch-stamped flags in HBM.
struct DfPlan {
  int nt = 0;
  int64_t n_tasks = 0;                          // bulk tasks: per block column J the diagonal accumulation PD(J), then the tiles (I, J) below, the rhs tile last
  DevBuf<int32_t> tasks;                        // 12 per task: I, J, offset / count into klist, piece r of R (the last piece finishes the tile), slots, accumulator lanes
  int64_t n_scratch = 0;                        // scratch slots behind the stored tiles of S: the accumulator lanes 1.. of the tiles with very long contraction lists
  DevBuf<long long> part_flag;                  // (nt + 1) x nt: kPieceBase epoch + pieces of the tile's contraction that are in
  DevBuf<int32_t> has_sub;                      // per diagonal tile J: tile (J, J-1) is stored (its update is streamed by the chain kernel)
  std::vector<int32_t> h_has_sub;
  DevBuf<int32_t> klist;                        // contraction lists: the column tiles k < J with both (I, k) and (J, k) stored; 6 words per step: the two slots, then the
                                                // 64-bit sub-tile masks of (I, k) and (J, k) (bit 8 r + q: rows 16 r.., columns 16 q.. can be non-zero)
  DevBuf<long long> tile_flag;                  // (nt + 1) x nt: epoch in which the tile became final
  DevBuf<long long> pd_flag;                    // nt: epoch in which the diagonal tile received all its updates
  int64_t shadow = 0;                           // every flag is also stored `shadow` words behind its word (chol_dataflow.hip::st_flag)
  DevBuf<int32_t> ctrl;                         // [0] ticket counter of the bulk queue; [8..15] record of the first wait that gave up
  DevBuf<long long> trace;                      // GTG_DF_TRACE=1: 4 stamps per task + 2 per diagonal tile (gtg_debug_df_trace)
  std::vector<int32_t> h_tasks, h_klist;        // host copies (debug getters, CPU tests)
  // the diagonal tiles by chain workgroup: workgroup w of k_df_chain factors chain_tiles[chain_off[w] .. chain_off[w + 1]) in that order.
  // One slot = two workgroups that alternate; several slots when a nested-dissection ordering gave the factorisation independent parts
  int n_chain = 0;
  DevBuf<int32_t> chain_off, chain_tiles;
  std::vector<int32_t> h_chain_off, h_chain_tiles, h_seq;   // h_seq: the order in which the block columns are taken (ticket groups)
  int critical_tiles = 0;                       // diagonal tiles of the longest slot
  double flops = 0.0, dense_fraction = 1.0;
  double flops_executed = 0.0;                  // flops minus the contraction MFMAs skipped on structurally empty 16 x 16 sub-tiles (upload_df_plan)
};

struct FactorTables {
  // SFM
  int64_t n_sfm = 0;
  DevBuf<int32_t> sfm_cam, sfm_point, sfm_noise;
  DevBuf<int32_t> sfm_cam_at, sfm_point_at;   // where the factor's camera / point start in the packed values (val_off of the two ids: one dependent gather less per recomputed record, fused.h)
  DevBuf<double> sfm_z, sfm_J;
  // projection
  int64_t n_proj = 0;
  DevBuf<int32_t> proj_pose, proj_point, proj_noise, proj_calib, proj_sensor;
  DevBuf<double> proj_z, proj_J, calib, sensor;
  // between
  int64_t n_between = 0;
  DevBuf<int32_t> between_v1, between_v2, between_noise;
  DevBuf<double> between_z, between_J;
  // prior
  int64_t n_prior = 0;
  DevBuf<int32_t> prior_var, prior_noise;
  DevBuf<int64_t> prior_off;
  DevBuf<double> prior_data, prior_J;
};

}  // namespace gt

namespace gt {
// Streams and events of the factorisation schedule (cholesky.hip) -- owned by the handle: streams and events belong to
// the device they were created on, and two handles driven from two host threads must not share them.
struct CholStreams {
  hipStream_t panel = nullptr;    // high-priority stream of the serial panel chain (the bulk updates run on the handle's stream)
  hipEvent_t start = nullptr;
  std::vector<hipEvent_t> P, N;   // per column pair: rest(p) finished / chain of pair p finished
};
}  // namespace gt

struct gtg_context {
  int device = 0;
  hipStream_t stream = nullptr;
  gt::CholStreams cs;
  std::vector<hipEvent_t> phase_events;   // 2 per phase (gtg_enable_timing)
  hipStream_t copy_stream = nullptr;      // non-blocking: the uploads that run beside the symbolic analysis (gtg_upload_problem)
  int shard = 0, n_shards = 1;
  bool uploaded = false, linearized = false, have_trial = false;

  // ---- variables -------------------------------------------------------------------------------
  int32_t n_vars = 0;
  std::vector<int32_t> h_var_type;
  std::vector<int64_t> h_val_off, h_dim_off;   // size n_vars+1
  int64_t val_size = 0, dim_size = 0;
  // Smart factors (SmartProjectionFactor): every factor owns a HIDDEN landmark variable (ids n_user_vars .. n_vars-1, appended
  // behind the caller's variables, so the caller's values / tangent vectors are prefixes of the device's) that is re-triangulated
  // from the cameras instead of being optimised; its measurements are GeneralSFM observations smart_obs0 .. of the sfm tables.
  int32_t n_user_vars = 0;
  int64_t user_val_size = 0, user_dim_size = 0;
  int64_t n_smart = 0, smart_obs0 = 0;
  gt::DevBuf<int64_t> smart_ptr;            // [n_smart + 1] measurements of a factor, relative to smart_obs0
  gt::DevBuf<double> smart_params;          // 8 per factor (include/gtsam_amd.h)
  // per factor: 0 VALID, 1 DEGENERATE, 2 BEHIND_CAMERA, 3 OUTLIER, 4 FAR_POINT (triangulation.h:611-612), + 16 (kTriAtInfinity) when this
  // use of a failed track replaces the landmark by a point at infinity.  TWO arrays: smart_lin_status is
  // the outcome at the LINEARISATION point (written by gtg_linearize only; read by everything that builds the linear system, which
  // stays fixed over the lambda retries of an iteration as the reference's linearised Hessian factor does); smart_status the
  // outcome of the most recent ERROR evaluation (gtg_error, the trial point of a lambda try), read by k_error only
  gt::DevBuf<int32_t> smart_status, smart_lin_status;
  gt::DevBuf<int32_t> smart_cache_state;    // per factor: -1 nothing cached, else the cached status
  gt::DevBuf<double> smart_cache_pose;      // 12 per measurement: the camera poses of the cached triangulation
  gt::DevBuf<double> smart_cache_point;     // 3 per factor
  gt::DevBuf<int32_t> sfm_smart, lm_smart;  // per sfm observation / per landmark: its smart factor or -1
  gt::DevBuf<int32_t> var_type;
  gt::DevBuf<int64_t> val_off, dim_off;
  gt::DevBuf<double> values, trial, delta;

  // classification: landmark (POINT3 eliminated first) or reduced variable
  int32_t n_lm = 0, n_red_vars = 0;
  int64_t n_red = 0;                            // scalar dimension of the reduced system
  uint64_t structure_hash = 0;                  // identity of the layout below (ordering, offsets, padding, tile structure): equal on every shard of a job
  int32_t NP = 0;                               // padded dimension of S: n_red + alignment gaps (parts start on 256-column boundaries), multiple of kTile
  std::vector<int64_t> h_pad_index;             // the padded (identity) rows/columns of S
  gt::DevBuf<int64_t> pad_index;
  std::vector<int32_t> h_lm_index, h_red_index; // per variable: index among landmarks / reduced (or -1)
  std::vector<int32_t> h_lm_var, h_red_var;     // inverse maps
  std::vector<int32_t> h_red_pos;               // reduced index -> position in the ordering
  std::vector<int64_t> h_red_off;               // per reduced index: scalar offset in S
  std::vector<int32_t> h_red_dim;
  gt::DevBuf<int32_t> lm_var, red_var, red_dim, lm_index, red_index, lm_owned;
  gt::DevBuf<int64_t> red_off;

  // ---- noise table (inverse sigmas precomputed like the reference constructors) ------------------
  gt::DevBuf<int32_t> noise_kind, noise_rkind;
  gt::DevBuf<double> noise_rk;
  gt::DevBuf<int64_t> noise_off;
  gt::DevBuf<double> noise_data;

  gt::FactorTables f;

  // ---- incidence (host-built CSR, device copies) ---------------------------------------------------
  // observations: obs id o < n_sfm -> sfm factor o; else projection factor o - n_sfm
  int64_t n_obs = 0;
  gt::DevBuf<int32_t> obs_red, obs_lm;          // reduced index / landmark index of each observation
  gt::DevBuf<int64_t> lm_obs_ptr;  gt::DevBuf<int32_t> lm_obs;      // landmark -> observations
  gt::DevBuf<int64_t> lm_pri_ptr;  gt::DevBuf<int32_t> lm_pri;      // landmark -> prior factors
  gt::DevBuf<int64_t> red_inc_ptr; gt::DevBuf<int32_t> red_inc_kind, red_inc_idx;  // reduced var -> contributions
  int64_t n_hoff = 0;                                                // off-diagonal blocks from between factors
  gt::DevBuf<int32_t> hoff_row, hoff_col;                            // reduced indices (row pos > col pos)
  gt::DevBuf<int64_t> hoff_ptr;  gt::DevBuf<int32_t> hoff_fac;       // block -> between factors (sign bit = transposed)
  int64_t n_pairs = 0, n_pair_terms = 0;                             // Schur block pairs
  bool device_terms = false;                                         // their term lists were built on the device (device_analysis.hip)
  gt::DevBuf<int32_t> pair_row, pair_col;
  gt::DevBuf<int64_t> pair_ptr;  gt::DevBuf<int32_t> pair_oa, pair_ob;

  // ---- numeric buffers ---------------------------------------------------------------------------
  gt::DevBuf<double> Hd, gred0, hdiag_red;      // reduced diag blocks (81), gradient (9), diagonal (n_red)
  gt::DevBuf<double> V, gp;                     // landmark blocks (9) and gradient (3)
  gt::DevBuf<double> Hoff;                      // (81 per block)
  gt::DevBuf<double> Linv, ylm, E, delta_lm;    // per try: landmark L^-1 (9), y (3), E (32/obs), delta (3)
  gt::DevBuf<double> pcg_vec, pcg_bj, pcg_y;    // PCG solver: r, p, q1, q2, b (5 x NP); block-Jacobi factors (81 / reduced variable); y_l (3 / landmark)
  bool fused_sfm = false;                       // the GeneralSFM records are recomputed where they are needed instead of stored (fused.h): graphs without smart factors
  gt::DevBuf<double> wobs;                      // per try: E_o y_l (9 doubles) per entry of the cameras' contribution lists: the summands of the reduced right-hand side
  gt::DevBuf<int32_t> obs_wpos;                 // observation -> its entry in red_inc_* (where k_obs_E stores its summand)
  gt::DevBuf<double> cam_part;                  // k_cam_fused with several workgroups per camera: their partial sums
  gt::DevBuf<double> cam_pack;                  // fused.h::CamPack per entry of the camera-sorted contribution lists (4 doubles each), once per graph
  gt::DevBuf<double> vobs;                      // per try: Jp^T (Jc x_cam) of every observation (3), for the back-substitution
  gt::DevBuf<double> S;                         // plan.n_stored slots of 128 x 128 doubles: the stored tiles of the reduced system + the rhs row's tiles
  gt::DevBuf<double> yred;                      // NP: y = L^-1 g gathered from the rhs tiles for the backward solve
  gt::DevBuf<double> Dinv;                      // per diagonal tile (128x128 doubles): the four 32x32 diagonal inverses, the MFMA operand
                                                // images of the tile's sub-blocks for the TRSM, and the tile's progress word (zeroed at allocation)
  gt::DevBuf<unsigned char> pivot_kind;         // per scalar column of S: 1 / 2 = last pivot of a variable (dim >= 2 / dim 1): the rank test of
  gt::DevBuf<double> tile_exp;                  // base/cholesky.cpp:144-157 applies there; tile_exp: exponent of every tile's last pivot (carry)
  gt::DevBuf<long long> chol_epoch_dev;         // factorisations launched so far (base of the progress words): a device copy of chol_epoch (debug)
  long long chol_epoch = 0;                     // counted on the host, passed to the kernels by value
  gt::CholPlan plan;
  gt::DfPlan df;                                // the dataflow schedule (default); `plan` keeps the lists for zeroing / exchange / backward solve
  bool use_df = true;
  int64_t df_fallbacks = 0;                    // lambda tries repeated with the stream schedule after a time-out of the dataflow pass
  gt::DevBuf<double> xbuf;                      // multi-GPU: the exchanged part of S packed contiguously for the all-reduce
  // multi-GPU exchange at block granularity: every structurally non-zero d x d block of the reduced system (diagonal blocks +
  // the off-diagonal blocks of the WHOLE graph, identical on every shard), 81 doubles per block, then the rhs row and the
  // padding diagonal.  3x less than whole 128x128 tiles on the L1723 shape (0.13 GB against 0.38 GB).
  int64_t n_xb = 0;
  gt::DevBuf<int64_t> xb_row_off, xb_col_off;   // scalar offsets of the block in S (row >= col)
  gt::DevBuf<int32_t> xb_dim;                   // rows | cols << 8
  gt::DevBuf<double> xred;                      // NP solution of the reduced system
  gt::DevBuf<double> partials;                  // block partial sums for reductions
  gt::DevBuf<double> scalars;                   // SC_COUNT
  double h_scalars[gt::SC_COUNT] = {0};

  // ---- multi-GPU exchange --------------------------------------------------------------------------
  gt::DevBuf<double> layout_probe;              // 4 doubles: pieces of structure_hash summed across the shards (verify_layout)
  bool layout_verified = false;
  gtg_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;

  // ---- timing ----------------------------------------------------------------------------------------
  bool timing = false;
  double phase_ms[GTG_PH_COUNT] = {0};
  int64_t phase_calls[GTG_PH_COUNT] = {0};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double chol_flops = 0, chol_flops_block = 0, lin_bytes = 0;
  std::thread block_level_thread;               // computes chol_flops_block beside / after analyze() (analysis.hip::join_block_level)
  std::exception_ptr block_level_err;
};

namespace gt {
inline SMat smat(const gtg_context& c) { return SMat{c.S.p, c.plan.slot.p, c.plan.nt}; }   // the handle's reduced system (context.h::SMat)
}  // namespace gt
