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
#include <stdexcept>

#include "kernels.h"

namespace gt {

namespace {

#include "schur_groups_kernel.h"

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

}  // namespace gt
