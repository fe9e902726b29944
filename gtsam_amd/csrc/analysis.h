// analysis.h -- the one-time host-side symbolic analysis of a graph (analysis.hip) and the few helpers it shares with the C ABI.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "context.h"

namespace gt {

static inline int storage_size(int t) { return t == GTG_VAR_POSE3 ? 12 : t == GTG_VAR_SFM_CAMERA ? 17 : 3; }   // POINT3, POSE2: 3
static inline int tangent_dim(int t) { return t == GTG_VAR_POSE3 ? 6 : t == GTG_VAR_SFM_CAMERA ? 9 : 3; }

// host copy of the (shard-filtered) factor index arrays needed by the symbolic analysis
struct HostIndex {
  std::vector<int32_t> sfm_cam, sfm_point, proj_pose, proj_point, between_v1, between_v2, prior_var;
  std::vector<int32_t> user_order;  // optional reduced ordering (variable ids)
  // n_shards > 1: the keys of EVERY observation and between factor of the whole graph (all shards).  The structure of
  // the reduced system -- ordering, offsets, tile schedule, exchange list -- must be identical on every shard, so it is
  // derived from the whole graph; only the numeric lists (terms, incidence) are the shard's own.
  std::vector<int32_t> all_obs_red_var, all_obs_point, all_between_v1, all_between_v2;
};
template <class T> static void up(DevBuf<T>& b, const std::vector<T>& v, hipStream_t s) {
  b.upload(v.data(), v.size(), s);
  if (v.empty()) b.alloc(1);  // keep kernels' pointer arguments non-null
}

struct StageClock {   // GTG_DEBUG_TIMING=1 prints the host-side setup breakdown
  bool on = std::getenv("GTG_DEBUG_TIMING") != nullptr;
  std::chrono::high_resolution_clock::time_point t = std::chrono::high_resolution_clock::now();
  void lap(const char* what) {
    if (!on) return;
    auto n = std::chrono::high_resolution_clock::now();
    std::fprintf(stderr, "[gtsam_amd setup] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

void join_block_level(gtg_context& c);       // waits for the handle's block-level flop count (analysis.hip: it runs on a thread of the handle)
HostIndex& host_index(gtg_context* c);     // the handle's host-side index arrays (created on first use)
void drop_index(gtg_context* c);
// landmark classification, CSR incidence lists, Schur block / term lists, ordering of the reduced variables, tile schedule
// of the Cholesky, device buffers; sharded handles derive the layout from the whole graph and verify it across the shards
void analyze(gtg_context& c);
void verify_layout(gtg_context& c);

}  // namespace gt
