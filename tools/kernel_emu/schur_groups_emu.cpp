// tools/kernel_emu/schur_groups_emu.cpp -- csrc/schur_groups_kernel.h on host threads (emu_hip.h), behind one C entry point for
// tests/test_schur_groups_emulated.py.  S is a dense NP x NP row-major matrix here (SMat::at_stored = plain indexing): what is under
// test is the kernel's walk -- chunk cut, staging, who accumulates what, the write-out -- not the tile storage.
#include "emu_hip.h"

constexpr int kSchurGroup = 8;
constexpr int kSchurChunkSlots = 128;
constexpr int kEStride = 32;
struct SMat {
  double* p; int64_t NP;
  double* at_stored(int64_t r, int64_t c) const { return (r >= 0 && c >= 0 && r < NP && c < NP) ? p + r * NP + c : nullptr; }
};

namespace gt { namespace {
#include "../../gtsam_amd/csrc/schur_groups_kernel.h"
} }

extern "C" int emu_schur_groups(int n_pairs, int NG, int nrv, const int32_t* order, const int32_t* pair_key, const int64_t* pair_ptr,
                                const int32_t* cell_a0, const int32_t* cell_b0, const int32_t* cell_pq, const int32_t* gobs,
                                const int32_t* pos_red, const int32_t* red_dim, const int64_t* red_off,
                                const double* E, double* S, int64_t NP, int pipelined) {
  SMat sm{S, NP};
  for (int b = 0; b < n_pairs; b++)
    emu::run_workgroup(gt::kThreads, (unsigned)b, [&] {
      if (pipelined) gt::k_schur_groups_pipe(n_pairs, NG, nrv, order, pair_key, pair_ptr, cell_a0, cell_b0, cell_pq, gobs, pos_red, red_dim, red_off, E, sm);
      else gt::k_schur_groups(n_pairs, NG, nrv, order, pair_key, pair_ptr, cell_a0, cell_b0, cell_pq, gobs, pos_red, red_dim, red_off, E, sm);
    });
  return 0;
}

// The pair-major sums of assemble.hip::k_schur_pairs in plain loops, with the emulated MFMA's arithmetic (fused multiply-add chain over
// k = 0..3, k = 3 a zero lane group): block (row variable, column variable), its terms (oa, ob) in list order, then S -= the sum.
extern "C" int emu_schur_pairs_reference(int64_t n_pairs, const int32_t* prow, const int32_t* pcol, const int64_t* pptr, const int32_t* oa,
                                         const int32_t* ob, const int32_t* red_dim, const int64_t* red_off, const double* E, double* S,
                                         int64_t NP) {
  for (int64_t p = 0; p < n_pairs; p++) {
    const int ra = prow[p], rb = pcol[p], da = red_dim[ra], db = red_dim[rb];
    double acc[16][16];
    for (auto& r : acc) for (double& v : r) v = 0.0;
    for (int64_t t = pptr[p]; t < pptr[p + 1]; t++) {
      const double* Ea = E + (int64_t)kEStride * oa[t];
      const double* Eb = E + (int64_t)kEStride * ob[t];
      for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
          double a = acc[i][j];
          for (int k = 0; k < 4; k++) {
            const double av = (i < da && k < 3) ? Ea[3 * i + k] : 0.0, bv = (j < db && k < 3) ? Eb[3 * j + k] : 0.0;
            a = std::fma(av, bv, a);
          }
          acc[i][j] = a;
        }
    }
    for (int i = 0; i < da; i++)
      for (int j = 0; j < db; j++) S[(red_off[ra] + i) * NP + red_off[rb] + j] -= acc[i][j];
  }
  return 0;
}

// ---- the per-landmark kernels of the device-side list builder (csrc/schur_groups_lists_kernel.h), one workgroup of 256 host threads at a time
namespace gt { namespace {
#include "../../gtsam_amd/csrc/schur_groups_lists_kernel.h"
} }

extern "C" int emu_sg_sort_count(int n_lm, const int64_t* ptr, const int32_t* lm_obs, const int32_t* obs_red, const int32_t* red_pos,
                                 int32_t* gobs, int32_t* gpos, int64_t* cnt) {
  for (int b = 0; b < (n_lm + 1 + 255) / 256; b++)
    emu::run_workgroup(256, (unsigned)b, [&] { gt::k_sg_sort_count(n_lm, ptr, lm_obs, obs_red, red_pos, gobs, gpos, cnt); });
  return 0;
}
extern "C" int emu_sg_emit(int n_lm, int NG, const int64_t* ptr, const int32_t* gpos, const int64_t* off, uint32_t* key, uint32_t* idx,
                           int32_t* a0, int32_t* b0, int32_t* pq, int32_t* bad) {
  for (int b = 0; b < (n_lm + 255) / 256; b++)
    emu::run_workgroup(256, (unsigned)b, [&] { gt::k_sg_emit(n_lm, NG, ptr, gpos, off, key, idx, a0, b0, pq, bad); });
  return 0;
}
