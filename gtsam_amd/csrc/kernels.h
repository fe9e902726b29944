// kernels.h -- host-callable launchers of the HIP kernels (all stream-ordered on ctx.stream).
#pragma once
#include "context.h"

namespace gt {

// factors.hip -------------------------------------------------------------------------------------
void launch_linearize(gtg_context& c);                                  // fills *_J from c.values
void launch_error(gtg_context& c, const double* values, int scalar_slot);  // nonlinear error -> scalars[slot]
void launch_linear_error(gtg_context& c);                               // scalars[SC_LIN0], [SC_LIN1] from J, delta
void launch_retract(gtg_context& c);                                    // trial = values (+) delta ; scalars[SC_DELTA_SQ]

// assemble.hip ------------------------------------------------------------------------------------
void launch_assemble(gtg_context& c);            // Hd, gred0, V, gp, Hoff, hdiag_red (lambda-invariant)
void launch_point_eliminate(gtg_context& c, double lambda, int diag, double dmin, double dmax);  // Linv, ylm, E
void launch_build_reduced(gtg_context& c, double lambda, int diag, double dmin, double dmax);    // S and rhs row
void launch_back_substitute(gtg_context& c);     // delta_lm from xred, ylm, E, Linv
void launch_scatter_delta(gtg_context& c);       // delta (variable id order) from xred + delta_lm

// cholesky.hip ------------------------------------------------------------------------------------
// In-place blocked Cholesky of the NP x NP lower triangle of S (ld = NP) carrying `extra_rows`
// additional rows (multiple of kTile) through TRSM + trailing updates (the rhs row: forward solve for
// free).  Non-positive pivots set *fail_flag (device double) to nonzero.
void launch_cholesky(gtg_context& c, double* S, int NP, int extra_rows, double* Dinv, double* fail_flag);
// x = L^-T y  with L the factor in S, y = row NP of S (first n entries). Result in xred[0..NP).
void launch_backward_solve(gtg_context& c, double* S, int NP, double* x);

void check_hip(hipError_t e, const char* what);

}  // namespace gt
