// kernels.h -- host-callable launchers of the HIP kernels (all stream-ordered on ctx.stream).
#pragma once
#include <initializer_list>
#include <vector>

#include "context.h"

namespace gt {

// factors.hip -------------------------------------------------------------------------------------
void launch_linearize(gtg_context& c);                                  // fills *_J from c.values (not the GeneralSFM records of a fused graph: fused.h)
void launch_sfm_value_offsets(gtg_context& c);                            // once per graph: f.sfm_cam_at / sfm_point_at = val_off[f.sfm_cam / sfm_point]
void launch_sfm_records(gtg_context& c, double* dst);                    // debug: the GeneralSFM records of the current values into dst
void launch_error(gtg_context& c, const double* values, int scalar_slot, const double* gate = nullptr);  // nonlinear error -> scalars[slot] (gate: as launch_smart_triangulate)
void launch_linear_error(gtg_context& c);                               // scalars[SC_LIN0], [SC_LIN1] from J, delta
void launch_retract(gtg_context& c);                                    // trial = values (+) delta ; scalars[SC_DELTA_SQ]

// assemble.hip ------------------------------------------------------------------------------------
void launch_obs_wpos(gtg_context& c, int64_t n_inc);   // once per graph: observation -> place in its camera's contribution list (where k_obs_E puts w_o)
void launch_assemble(gtg_context& c);            // Hd, gred0, V, gp, Hoff, hdiag_red (lambda-invariant)
void launch_point_eliminate(gtg_context& c, double lambda, int diag, double dmin, double dmax);  // Linv, ylm, E
void launch_build_reduced(gtg_context& c, double lambda, int diag, double dmin, double dmax);    // S and rhs row
void launch_back_substitute(gtg_context& c);     // delta_lm from xred, ylm, E, Linv
void launch_scatter_delta(gtg_context& c);       // delta (variable id order) from xred + delta_lm

// cholesky.hip ------------------------------------------------------------------------------------
// Host: build the tile schedule.  pair_struct = lower-triangular boolean structure over 256-wide column pairs
// ((np x np) row-major bytes, np = ceil(nt/2); nullptr = dense); symbolic fill-in is computed here.
void build_chol_plan(CholPlan& plan, int nt, const std::vector<uint8_t>* pair_struct, hipStream_t s,
                     const std::vector<int32_t>* pair_part = nullptr, const std::vector<int32_t>* part_parent = nullptr);
void ensure_stream_lists(CholPlan& plan, hipStream_t s);   // the stream schedule's SYRK pair lists, built on its first use
// In-place tile-sparse blocked Cholesky of the NP x NP lower triangle of S (ld = NP) carrying one extra 128-row
// tile (the rhs: forward solve for free).  Non-positive pivots set *fail_flag (device double) to nonzero.
void launch_zero_tiles(gtg_context& c, SMat S, const CholPlan& plan);
void launch_cholesky(gtg_context& c, SMat S, int NP, const CholPlan& plan, double* Xinv, double* fail_flag,
                     const unsigned char* pivot_kind = nullptr, double* tile_exp = nullptr);
// x = L^-T y  with L the factor in S, y = row NP of S. Result in x[0..NP).
void launch_pack_tiles(gtg_context& c, SMat S, const CholPlan& plan, double* buf, bool unpack);
int64_t exchange_block_doubles(const gtg_context& c);                    // size of the block-granular exchange buffer
void launch_pack_blocks(gtg_context& c, SMat S, int NP, double* buf, bool unpack);
// device_analysis.hip: the Schur term lists built on the device (single shard, real runtime)
void device_incidence_lists(gtg_context& c, const std::vector<int32_t>& red_pos, gt::DevBuf<int32_t>& d_pos);
void device_schur_terms(gtg_context& c, gt::DevBuf<int32_t>& d_pos, int nrv, const std::vector<int32_t>& pos_to_red, std::vector<int32_t>& block_row,
                        std::vector<int32_t>& block_col, std::vector<int64_t>& block_ptr);   // blocks stay in c.pair_row / c.pair_col; host copies out
void device_orient_blocks(gtg_context& c, int64_t n_blocks, const std::vector<int32_t>& red_pos);   // after the ordering: row = the later position
// after the ordering: the marks of the off-diagonal Schur blocks (c.pair_row / c.pair_col, already on the device) in the tile structure T1
// (nt x nt bytes), the strip structure M16 (n16 columns of w16 words) and the commutative sum over the block set (structure_hash)
struct RedLayout { int64_t off, dim; };      // scalar offset and dimension of a reduced variable in S
void device_tile_marks(gtg_context& c, int64_t n_blocks, const std::vector<RedLayout>& layout, int nt, int n16, int w16,
                       std::vector<uint8_t>& T1, std::vector<uint64_t>& M16, uint64_t* block_sum);
// device_ordering.hip: reverse Cuthill-McKee of the reduced variables' block graph on the device (false: outside what the kernel handles)
// (edges on the host: uploaded; d_edge_a / d_edge_b non-null: m edges already on the device, pairs a == b among them are skipped)
bool device_rcm(gtg_context& c, int n, const std::vector<int32_t>& edge_a, const std::vector<int32_t>& edge_b, std::vector<int32_t>& order,
                const int32_t* d_edge_a = nullptr, const int32_t* d_edge_b = nullptr, int64_t d_edges = 0);
// smart factors (SmartProjectionFactor): triangulation of the hidden landmarks from the cameras in `values` (gated: only when the
// linear cost change of the current try is >= 0), Schur-complement correction of the Hessian diagonal, constant of linear.error
void launch_smart_triangulate(gtg_context& c, double* values, const double* gate, bool for_linearize);   // gate: scalars with the linear errors (trial point) or null
void launch_smart_hdiag(gtg_context& c);
void launch_smart_lin1(gtg_context& c);
void exchange_sum(gtg_context& c, double* ptr, int64_t n);   // api.hip: all-reduce (sum) over the shards on the handle's stream; no-op on one shard
void launch_backward_solve(gtg_context& c, SMat S, int NP, const CholPlan& plan, const double* Xinv, double* x, double* fail);
void destroy_chol_streams(gtg_context& c);

// chol_dataflow.hip -----------------------------------------------------------------------------------
// The same factorisation as one dataflow pass of two persistent kernels (default schedule).  tile_struct = lower-triangular
// boolean structure over 128x128 tiles before the factorisation ((nt x nt) row-major bytes; nullptr = dense).
void build_df_plan(DfPlan& df, int nt, const std::vector<uint8_t>* tile_struct, hipStream_t s,
                   const std::vector<int32_t>& slot, int64_t n_slots,     // tile -> slot table of the stored tiles (CholPlan::h_slot)
                   const std::vector<int32_t>* tile_part = nullptr, const std::vector<int32_t>* part_parent = nullptr);
void build_df_plan_host(DfPlan& df, int nt, const std::vector<uint8_t>* tile_struct,             // the host half (no runtime call)
                        const std::vector<int32_t>* tile_part = nullptr, const std::vector<int32_t>* part_parent = nullptr);
void upload_df_plan(DfPlan& df, hipStream_t s, const std::vector<int32_t>& slot, int64_t n_slots, const std::vector<uint64_t>* sub16 = nullptr,
                    const int32_t* d_slot = nullptr);   // the device half (d_slot: the device copy of `slot` -- the tables are resolved by a kernel then)
void free_df_plan(DfPlan& df);
void df_prepare_streams_async(int device, int n_chain);   // the masked stream pair of a plan with n_chain chain workgroups, created on a helper thread ahead of the first factorisation
void df_join_prepared();                                       // (joined by the first factorisation and by gtg_destroy)
bool dataflow_schedule_selected();   // false: GTG_CHOL=streams (the stream / event schedule of cholesky.hip, the A/B of the dataflow pass)
void launch_cholesky_df(gtg_context& c, SMat S, int NP, DfPlan& df, double* Xinv, double* fail_flags,
                        const unsigned char* pivot_kind = nullptr, double* tile_exp = nullptr);

// pcg.hip -----------------------------------------------------------------------------------------
// Block-Jacobi PCG on the implicit Schur complement (needs launch_point_eliminate first): c.xred = S^-1 b.
int launch_pcg(gtg_context& c, double lambda, int diag, double dmin, double dmax, int max_iterations, int min_iterations,
               double epsilon_rel, double epsilon_abs, double* gamma0, double* gamma_end);

void check_hip(hipError_t e, const char* what);

// gtg_prewarm (api.hip): every translation unit with kernels registers a function that makes the runtime load the unit's code object and
// create the function objects of its kernels (hipFuncGetAttributes does both, without a launch) -- the work a kernel's FIRST launch
// would otherwise do (measured on the L1723 shape, tools/cpp/cold_start_probe.cpp: ~37 ms inside the first gtg_upload_problem and ~20 ms
// inside the first gtg_try_lambda of a process).
struct PrewarmUnit { explicit PrewarmUnit(void (*fn)(int device)); };
inline void prewarm_kernels(std::initializer_list<const void*> kernels) {
  hipFuncAttributes a;
  for (const void* k : kernels) (void)hipFuncGetAttributes(&a, k);
}

}  // namespace gt
