// analysis.hip -- the one-time host-side symbolic analysis of a graph.
//
// Host work here is O(#factors) bookkeeping done ONCE per graph: landmark classification, CSR incidence lists, the block
// pattern of the Schur complement and its term lists (the summation order of the device), the fill-reducing order of the
// reduced variables and the tile schedule of their Cholesky.  It replaces the VariableIndex / EliminationTree /
// JunctionTree / Scatter construction that the reference repeats on every lambda try (inference/VariableIndex-inl.h:27-49,
// EliminationTree-inst.h:77-155, JunctionTree-inst.h:63-151, linear/Scatter.cpp:39-73).  Runs on host threads; the
// result does not depend on their number.  No arithmetic of the hot path lives here.
#include <algorithm>
#include <atomic>
#include <exception>
#include <functional>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <stdexcept>
#include <cstring>
#include <sys/mman.h>
#include <thread>

#include "analysis.h"
#include "factors.h"
#include "kernels.h"

namespace gt {

static std::vector<std::pair<gtg_context*, HostIndex*>> g_index;  // tiny registry (handles are few)
static std::mutex g_index_mutex;                                   // handles may be created / destroyed from several host threads
HostIndex& host_index(gtg_context* c) {
  std::lock_guard<std::mutex> lock(g_index_mutex);
  for (auto& kv : g_index) if (kv.first == c) return *kv.second;
  g_index.emplace_back(c, new HostIndex);
  return *g_index.back().second;
}
void drop_index(gtg_context* c) {
  std::lock_guard<std::mutex> lock(g_index_mutex);
  for (size_t i = 0; i < g_index.size(); i++)
    if (g_index[i].first == c) { delete g_index[i].second; g_index.erase(g_index.begin() + i); return; }
}

// Large scratch arrays of the analysis (tens of MB, first touched by many threads at once): 2 MB aligned and advised to
// transparent huge pages, so that the first touch is a few dozen page faults instead of tens of thousands serialised on
// the process' address-space lock.  Not value-initialised.
template <class T> struct HugeBuf {
  T* p = nullptr;
  explicit HugeBuf(size_t n) {
    const size_t bytes = std::max<size_t>((n * sizeof(T) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1), (size_t)2 << 20);
    p = static_cast<T*>(std::aligned_alloc((size_t)2 << 20, bytes));
    if (!p) throw std::bad_alloc();
    (void)madvise(p, bytes, MADV_HUGEPAGE);
  }
  ~HugeBuf() { std::free(p); }
  HugeBuf(const HugeBuf&) = delete;
  HugeBuf& operator=(const HugeBuf&) = delete;
  HugeBuf(HugeBuf&& o) noexcept : p(o.p) { o.p = nullptr; }
  HugeBuf& operator=(HugeBuf&& o) noexcept { if (this != &o) { std::free(p); p = o.p; o.p = nullptr; } return *this; }
  T& operator[](size_t i) { return p[i]; }
  T* get() { return p; }
  void reset() { std::free(p); p = nullptr; }
};

// host threads of the symbolic analysis: GTG_HOST_THREADS, else the hardware concurrency capped at 32
static int host_threads() {
  static const int n = [] {
    const char* e = std::getenv("GTG_HOST_THREADS");
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    return e ? std::max(1, std::atoi(e)) : std::min(hw, 32);
  }();
  return n;
}
template <class F> static void run_threads(int nt, F f) {   // f(thread index) on nt threads (the caller is thread 0)
  if (nt <= 1) { f(0); return; }
  std::vector<std::thread> th;
  std::exception_ptr err = nullptr; std::mutex m;
  auto guarded = [&](int t) { try { f(t); } catch (...) { std::lock_guard<std::mutex> g(m); if (!err) err = std::current_exception(); } };
  for (int t = 1; t < nt; t++) th.emplace_back(guarded, t);
  guarded(0);
  for (auto& x : th) x.join();
  if (err) std::rethrow_exception(err);
}

// Three 20-bit pieces of the layout hash and a 1 go through the all-reduce: the sums must be n_shards times this shard's
// own values (every shard derived the same layout AND the communicator spans n_shards ranks).
void verify_layout(gtg_context& c) {
  c.layout_verified = true;
  double mine[4] = {(double)(c.structure_hash & 0xFFFFF), (double)((c.structure_hash >> 20) & 0xFFFFF), (double)((c.structure_hash >> 40) & 0xFFFFF), 1.0};
  double sum[4] = {0, 0, 0, 0};
  check_hip(hipMemcpyAsync(c.layout_probe.p, mine, sizeof(mine), hipMemcpyHostToDevice, c.stream), "H2D");
  if (c.allreduce(c.layout_probe.p, 4, (void*)c.stream, c.allreduce_user) != 0) throw std::runtime_error("allreduce callback failed");
  check_hip(hipMemcpyAsync(sum, c.layout_probe.p, sizeof(sum), hipMemcpyDeviceToHost, c.stream), "D2H");
  check_hip(hipStreamSynchronize(c.stream), "sync");
  for (int i = 0; i < 4; i++)
    if (sum[i] != mine[i] * c.n_shards)
      throw std::runtime_error("sharded upload: the shards disagree on the layout of the reduced system (or the all-reduce spans a "
                               "different number of ranks than n_shards)");
}

// ---- symbolic analysis ------------------------------------------------------------------------------
void analyze(gtg_context& c) {
  StageClock clk;
  HostIndex& hi = host_index(&c);
  const int nv = c.n_vars;
  const int64_t n_sfm = c.f.n_sfm, n_proj = c.f.n_proj, n_btw = c.f.n_between, n_pri = c.f.n_prior;
  hipStream_t s = c.stream;

  // landmarks = POINT3 variables (eliminated first, timing/timeSFMBAL.h:74-83); the rest is reduced
  c.h_lm_index.assign(nv, -1); c.h_red_index.assign(nv, -1);
  c.h_lm_var.clear(); c.h_red_var.clear();
  for (int v = 0; v < nv; v++) {
    if (c.h_var_type[v] == GTG_VAR_POINT3) { c.h_lm_index[v] = (int)c.h_lm_var.size(); c.h_lm_var.push_back(v); }
    else { c.h_red_index[v] = (int)c.h_red_var.size(); c.h_red_var.push_back(v); }
  }
  c.n_lm = (int)c.h_lm_var.size(); c.n_red_vars = (int)c.h_red_var.size();
  // The incidence and term lists come from the device (device_analysis.hip) on a single shard with a real runtime; from the host threads
  // below for a shard (which needs the blocks of the WHOLE graph but only its own terms), under the dry-run runtime of the CPU
  // tests (no kernels run there) and with GTG_HOST_ANALYSIS=1 (the A/B: both give bit-identical lists).
  // (kernels_can_run: the library's code objects are gfx950 only; a runtime that reports another architecture -- the dry-run
  // runtime of the CPU tests reports none -- cannot run the device pass)
  bool kernels_can_run = false;
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c.device) == hipSuccess) kernels_can_run = std::strncmp(prop.gcnArchName, "gfx", 3) == 0; }
  const bool device_terms = c.n_shards == 1 && c.n_lm > 0 && !std::getenv("GTG_HOST_ANALYSIS") && kernels_can_run;
  c.device_terms = device_terms;
  // validate factor roles (the observation factors' in the device pass when it runs: device_incidence_lists)
  if (!device_terms) {
    for (int64_t i = 0; i < n_sfm; i++)
      if (c.h_var_type[hi.sfm_cam[i]] != GTG_VAR_SFM_CAMERA || c.h_var_type[hi.sfm_point[i]] != GTG_VAR_POINT3)
        throw std::invalid_argument("GeneralSFMFactor keys must be (SFM_CAMERA, POINT3)");
    for (int64_t i = 0; i < n_proj; i++)
      if (c.h_var_type[hi.proj_pose[i]] != GTG_VAR_POSE3 || c.h_var_type[hi.proj_point[i]] != GTG_VAR_POINT3)
        throw std::invalid_argument("GenericProjectionFactor keys must be (POSE3, POINT3)");
  }
  for (int64_t i = 0; i < n_btw; i++)
    if (c.h_var_type[hi.between_v1[i]] != c.h_var_type[hi.between_v2[i]] || hi.between_v1[i] == hi.between_v2[i] ||
        (c.h_var_type[hi.between_v1[i]] != GTG_VAR_POSE3 && c.h_var_type[hi.between_v1[i]] != GTG_VAR_POSE2))
      throw std::invalid_argument("BetweenFactor keys must be two distinct POSE3 (or two distinct POSE2) variables");

  // ordering of the reduced variables
  c.h_red_pos.assign(c.n_red_vars, -1);
  if (!hi.user_order.empty()) {
    if ((int)hi.user_order.size() != c.n_red_vars) throw std::invalid_argument("reduced ordering has wrong length");
    for (int i = 0; i < c.n_red_vars; i++) {
      const int v = hi.user_order[i];
      if (v < 0 || v >= nv || c.h_red_index[v] < 0 || c.h_red_pos[c.h_red_index[v]] >= 0)
        throw std::invalid_argument("reduced ordering is not a permutation of the non-landmark variables");
      c.h_red_pos[c.h_red_index[v]] = i;
    }
  } else {
    for (int r = 0; r < c.n_red_vars; r++) c.h_red_pos[r] = r;
  }
  std::vector<int32_t> pos_to_red(c.n_red_vars);
  for (int r = 0; r < c.n_red_vars; r++) pos_to_red[c.h_red_pos[r]] = r;
  c.h_red_dim.assign(c.n_red_vars, 0); c.h_red_off.assign(c.n_red_vars, 0);
  int64_t off = 0;
  for (int p = 0; p < c.n_red_vars; p++) {
    const int r = pos_to_red[p];
    c.h_red_dim[r] = tangent_dim(c.h_var_type[c.h_red_var[r]]);
    c.h_red_off[r] = off; off += c.h_red_dim[r];
  }
  c.n_red = off;
  c.NP = (int)((std::max<int64_t>(off, 1) + kTile - 1) / kTile * kTile);

  // observations
  c.n_obs = n_sfm + n_proj;
  std::vector<int32_t> obs_red, obs_lm, lm_obs, inc_kind, inc_idx;
  std::vector<int64_t> lm_obs_ptr, inc_ptr;
  DevBuf<int32_t> d_obs_pos;                    // (device pass: observation -> position of its camera, consumed by device_schur_terms)
  if (device_terms) {
    clk.lap("variable roles, layout of the reduced variables (host)");
    device_incidence_lists(c, c.h_red_pos, d_obs_pos);
    clk.lap("incidence lists (device)");
  } else {
  obs_red.resize(c.n_obs); obs_lm.resize(c.n_obs);
  for (int64_t i = 0; i < n_sfm; i++) { obs_red[i] = c.h_red_index[hi.sfm_cam[i]]; obs_lm[i] = c.h_lm_index[hi.sfm_point[i]]; }
  for (int64_t i = 0; i < n_proj; i++) { obs_red[n_sfm + i] = c.h_red_index[hi.proj_pose[i]]; obs_lm[n_sfm + i] = c.h_lm_index[hi.proj_point[i]]; }

  // landmark -> observations (CSR, factor order)
  lm_obs_ptr.assign(c.n_lm + 1, 0);
  for (int64_t o = 0; o < c.n_obs; o++) lm_obs_ptr[obs_lm[o] + 1]++;
  for (int l = 0; l < c.n_lm; l++) lm_obs_ptr[l + 1] += lm_obs_ptr[l];
  lm_obs.resize(c.n_obs);
  { std::vector<int64_t> w(lm_obs_ptr.begin(), lm_obs_ptr.end() - 1);
    for (int64_t o = 0; o < c.n_obs; o++) lm_obs[w[obs_lm[o]]++] = (int32_t)o; }

  // reduced variable -> contributions
  inc_ptr.assign(c.n_red_vars + 1, 0);
  auto count = [&](int v) { const int r = c.h_red_index[v]; if (r >= 0) inc_ptr[r + 1]++; };
  for (int64_t i = 0; i < n_sfm; i++) count(hi.sfm_cam[i]);
  for (int64_t i = 0; i < n_proj; i++) count(hi.proj_pose[i]);
  for (int64_t i = 0; i < n_btw; i++) { count(hi.between_v1[i]); count(hi.between_v2[i]); }
  for (int64_t i = 0; i < n_pri; i++) count(hi.prior_var[i]);
  for (int r = 0; r < c.n_red_vars; r++) inc_ptr[r + 1] += inc_ptr[r];
  inc_kind.resize(inc_ptr[c.n_red_vars]); inc_idx.resize(inc_ptr[c.n_red_vars]);
  { std::vector<int64_t> w(inc_ptr.begin(), inc_ptr.end() - 1);
    auto put = [&](int v, int kind, int64_t idx) { const int r = c.h_red_index[v]; if (r >= 0) { inc_kind[w[r]] = kind; inc_idx[w[r]++] = (int32_t)idx; } };
    for (int64_t i = 0; i < n_sfm; i++) put(hi.sfm_cam[i], 0, i);
    for (int64_t i = 0; i < n_proj; i++) put(hi.proj_pose[i], 1, i);
    for (int64_t i = 0; i < n_btw; i++) { put(hi.between_v1[i], 2, i); put(hi.between_v2[i], 3, i); }
    for (int64_t i = 0; i < n_pri; i++) put(hi.prior_var[i], 4, i); }
  }
  // landmark -> priors (CSR, factor order; a handful)
  std::vector<int64_t> lm_pri_ptr(c.n_lm + 1, 0);
  int64_t n_lm_pri = 0;
  for (int64_t i = 0; i < n_pri; i++) { const int l = c.h_lm_index[hi.prior_var[i]]; if (l >= 0) { lm_pri_ptr[l + 1]++; n_lm_pri++; } }
  if (n_lm_pri) for (int l = 0; l < c.n_lm; l++) lm_pri_ptr[l + 1] += lm_pri_ptr[l];     // (no prior on a landmark: the offsets are all zero already)
  std::vector<int32_t> lm_pri(lm_pri_ptr[c.n_lm]);
  if (n_lm_pri) { std::vector<int64_t> w(lm_pri_ptr.begin(), lm_pri_ptr.end() - 1);
    for (int64_t i = 0; i < n_pri; i++) { const int l = c.h_lm_index[hi.prior_var[i]]; if (l >= 0) lm_pri[w[l]++] = (int32_t)i; } }
  std::vector<int32_t> lm_owned(std::max(c.n_lm, 1), c.n_shards == 1 ? 1 : 0);
  if (c.n_lm == 0) lm_owned[0] = 0;
  if (c.n_shards > 1) for (int l = 0; l < c.n_lm; l++) lm_owned[l] = (l % c.n_shards) == c.shard;

  // off-diagonal pose-pose blocks from BetweenFactors
  struct HB { int64_t key; int32_t code; };
  std::vector<HB> hb(n_btw);
  for (int64_t i = 0; i < n_btw; i++) {
    const int r1 = c.h_red_index[hi.between_v1[i]], r2 = c.h_red_index[hi.between_v2[i]];
    const bool swap = c.h_red_pos[r2] > c.h_red_pos[r1];  // row variable (later position) is key2
    const int rr = swap ? r2 : r1, rc = swap ? r1 : r2;
    hb[i].key = (int64_t)c.h_red_pos[rr] * c.n_red_vars + c.h_red_pos[rc];
    hb[i].code = (int32_t)i | (swap ? (1 << 30) : 0);
  }
  std::stable_sort(hb.begin(), hb.end(), [](const HB& a, const HB& b) { return a.key < b.key; });
  std::vector<int32_t> hoff_row, hoff_col, hoff_fac(n_btw);
  std::vector<int64_t> hoff_ptr;
  for (int64_t i = 0; i < n_btw; i++) {
    if (i == 0 || hb[i].key != hb[i - 1].key) {
      hoff_ptr.push_back(i);
      hoff_row.push_back(pos_to_red[hb[i].key / c.n_red_vars]);
      hoff_col.push_back(pos_to_red[hb[i].key % c.n_red_vars]);
    }
    hoff_fac[i] = hb[i].code;
  }
  hoff_ptr.push_back(n_btw);
  c.n_hoff = (int64_t)hoff_row.size();

  clk.lap(device_terms ? "landmark priors, between blocks (host)" : "incidence lists");
  // Schur block pairs: for every landmark, every pair of its observations is one term E_a E_b^T of the block
  // (row = the later position, column = the earlier one).  Terms are bucketed by the row position of their block
  // (counting sort), then every row bucket is sorted by column position (stable: the generation order = landmark
  // order is kept inside a block, so the summation order is reproducible).  All passes run on host threads and the
  // result does not depend on their number: a thread owns a contiguous range of landmarks and writes behind the
  // terms of the threads before it in every row bucket; rows are sorted independently.
  struct PT { int32_t pb, oa, ob; };
  const int nrv = c.n_red_vars;
  const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), c.n_obs / 16384));
  std::vector<int32_t> obs_pos;
  if (!device_terms) { obs_pos.resize(c.n_obs); for (int64_t o = 0; o < c.n_obs; o++) obs_pos[o] = c.h_red_pos[obs_red[o]]; }
  auto for_terms = [&](int l0, int l1, auto&& emit) {
    for (int l = l0; l < l1; l++)
      for (int64_t a = lm_obs_ptr[l]; a < lm_obs_ptr[l + 1]; a++) {
        const int32_t oa0 = lm_obs[a]; const int pa0 = obs_pos[oa0];
        for (int64_t b = lm_obs_ptr[l]; b <= a; b++) {
          int32_t oa = oa0, ob = lm_obs[b];
          int pa = pa0, pb = obs_pos[ob];
          if (pa < pb) { std::swap(oa, ob); std::swap(pa, pb); }
          emit(pa, pb, oa, ob);
          if (pa == pb && oa != ob) emit(pa, pb, ob, oa);   // same camera twice
        }
      }
  };
  int64_t n_terms = 0;
  HugeBuf<int32_t> pair_oa(1), pair_ob(1);
  std::vector<int32_t> pair_row, pair_col;
  std::vector<int64_t> pair_ptr;
  if (device_terms) {
    device_schur_terms(c, d_obs_pos, nrv, pos_to_red, pair_row, pair_col, pair_ptr);   // (the block list itself stays in c.pair_row / c.pair_col)
    n_terms = c.n_pair_terms;
    clk.lap("incidence lists, schur terms + block list (device)");
  } else {
  std::vector<int> lm_cut(nth + 1, c.n_lm);      // landmark ranges with equal numbers of terms
  {
    std::vector<int64_t> cum(c.n_lm + 1, 0);
    for (int l = 0; l < c.n_lm; l++) { const int64_t k = lm_obs_ptr[l + 1] - lm_obs_ptr[l]; cum[l + 1] = cum[l] + k * (k + 1) / 2; }
    lm_cut[0] = 0;
    for (int t = 1; t < nth; t++) lm_cut[t] = (int)(std::lower_bound(cum.begin(), cum.end(), cum[c.n_lm] / nth * t) - cum.begin());
    for (int t = 1; t <= nth; t++) lm_cut[t] = std::min(c.n_lm, std::max(lm_cut[t], lm_cut[t - 1]));
    lm_cut[nth] = c.n_lm;
  }
  std::vector<std::vector<int64_t>> cursor(nth, std::vector<int64_t>(nrv + 1, 0));
  run_threads(nth, [&](int t) { auto& cnt = cursor[t]; for_terms(lm_cut[t], lm_cut[t + 1], [&](int pa, int, int32_t, int32_t) { cnt[pa]++; }); });
  std::vector<int64_t> row_ptr(nrv + 1, 0);
  for (int r = 0; r < nrv; r++) {
    int64_t at = row_ptr[r];
    for (int t = 0; t < nth; t++) { const int64_t k = cursor[t][r]; cursor[t][r] = at; at += k; }   // count -> write cursor
    row_ptr[r + 1] = at;
  }
  n_terms = row_ptr[nrv];
  HugeBuf<PT> pt((size_t)std::max<int64_t>(n_terms, 1));                 // first touched by the writers
  run_threads(nth, [&](int t) { auto& w = cursor[t]; for_terms(lm_cut[t], lm_cut[t + 1], [&](int pa, int pb, int32_t oa, int32_t ob) { pt[w[pa]++] = PT{pb, oa, ob}; }); });
  clk.lap("schur terms bucketed");
  // per row bucket: stable counting sort by column position straight into the final term lists + the row's blocks
  pair_oa = HugeBuf<int32_t>((size_t)std::max<int64_t>(n_terms, 1)); pair_ob = HugeBuf<int32_t>((size_t)std::max<int64_t>(n_terms, 1));
  struct RowBlocks { std::vector<int32_t> col; std::vector<int64_t> start; };
  std::vector<RowBlocks> row_blocks(nrv);
  {
    std::atomic<int> next{0};
    run_threads(nth, [&](int) {
      std::vector<int64_t> cnt(nrv + 2, 0);
      for (;;) {
        const int r0 = next.fetch_add(4), r1 = std::min(nrv, r0 + 4);
        if (r0 >= nrv) break;
        for (int r = r0; r < r1; r++) {
          const int64_t b = row_ptr[r], e = row_ptr[r + 1];
          if (e == b) continue;
          for (int64_t i = b; i < e; i++) cnt[pt[i].pb + 1]++;
          RowBlocks& rb = row_blocks[r];
          for (int q = 0; q <= r; q++) {                       // columns of row r are <= r
            if (cnt[q + 1]) { rb.col.push_back(q); rb.start.push_back(b + cnt[q]); }
            cnt[q + 1] += cnt[q];
          }
          for (int64_t i = b; i < e; i++) { const int64_t d = b + cnt[pt[i].pb]++; pair_oa[d] = pt[i].oa; pair_ob[d] = pt[i].ob; }
          std::fill(cnt.begin(), cnt.begin() + r + 2, 0);
        }
      }
    });
  }
  pt.reset();
  clk.lap("schur terms sorted");
  { size_t nb = 0;
    for (int r = 0; r < nrv; r++) nb += row_blocks[r].col.size();
    pair_row.reserve(nb); pair_col.reserve(nb); pair_ptr.reserve(nb + 1);
    for (int r = 0; r < nrv; r++)
      for (size_t k = 0; k < row_blocks[r].col.size(); k++) {
        pair_row.push_back(pos_to_red[r]); pair_col.push_back(pos_to_red[row_blocks[r].col[k]]); pair_ptr.push_back(row_blocks[r].start[k]);
      }
    pair_ptr.push_back(n_terms); }
  }
  c.n_pairs = (int64_t)pair_row.size(); c.n_pair_terms = n_terms;
  clk.lap("schur block list");

  // ---- block structure of the reduced system: this handle's own blocks, or -- sharded -- those of the WHOLE graph ----
  // (a shard only has the Schur blocks of its own landmarks; an ordering or a tile list derived from them would differ
  // from shard to shard and the exchanged buffers would not line up)
  std::vector<int32_t> sb_row, sb_col;
  if (c.n_shards > 1) {
    const size_t words = ((size_t)nrv + 63) / 64;
    if ((double)nrv * (double)words * 8.0 > 2e9) throw std::invalid_argument("sharded analysis: too many reduced variables for the block bitmap");
    std::vector<uint64_t> bits((size_t)nrv * words, 0);
    auto set_block = [&](int ra, int rb) {
      if (ra == rb) return;
      const int hi_ = std::max(ra, rb), lo_ = std::min(ra, rb);
      __atomic_fetch_or(&bits[(size_t)hi_ * words + (size_t)(lo_ >> 6)], (uint64_t)1 << (lo_ & 63), __ATOMIC_RELAXED);
    };
    const int64_t n_all = (int64_t)hi.all_obs_point.size();
    std::vector<int64_t> aptr(c.n_lm + 1, 0);
    for (int64_t o = 0; o < n_all; o++) {
      const int l = c.h_lm_index[hi.all_obs_point[o]], r = c.h_red_index[hi.all_obs_red_var[o]];
      if (l < 0 || r < 0) throw std::invalid_argument("observation factor keys must be (camera / pose, POINT3)");
      aptr[l + 1]++;
    }
    for (int l = 0; l < c.n_lm; l++) aptr[l + 1] += aptr[l];
    std::vector<int32_t> acam(n_all);
    { std::vector<int64_t> w(aptr.begin(), aptr.end() - 1);
      for (int64_t o = 0; o < n_all; o++) acam[w[c.h_lm_index[hi.all_obs_point[o]]]++] = c.h_red_index[hi.all_obs_red_var[o]]; }
    const int nt_s = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n_all / 16384));
    run_threads(nt_s, [&](int t) {
      const int l0 = (int)((int64_t)c.n_lm * t / nt_s), l1 = (int)((int64_t)c.n_lm * (t + 1) / nt_s);
      for (int l = l0; l < l1; l++)
        for (int64_t a = aptr[l]; a < aptr[l + 1]; a++)
          for (int64_t b = aptr[l]; b < a; b++) set_block(acam[a], acam[b]);
    });
    for (size_t i = 0; i < hi.all_between_v1.size(); i++) {
      const int r1 = c.h_red_index[hi.all_between_v1[i]], r2 = c.h_red_index[hi.all_between_v2[i]];
      if (r1 >= 0 && r2 >= 0) set_block(r1, r2);
    }
    for (int r = 0; r < nrv; r++)
      for (size_t w = 0; w < words; w++) {
        uint64_t m = bits[(size_t)r * words + w];
        while (m) { const int b = __builtin_ctzll(m); m &= m - 1; sb_row.push_back(r); sb_col.push_back((int)(w * 64 + b)); }
      }
    clk.lap("whole-graph block structure (sharded)");
  }
  auto for_each_block = [&](auto&& f) {   // every off-diagonal block of the reduced system's structure (reduced indices)
    if (c.n_shards > 1) { for (size_t i = 0; i < sb_row.size(); i++) f(sb_row[i], sb_col[i]); return; }
    for (size_t i = 0; i < pair_row.size(); i++) f(pair_row[i], pair_col[i]);
    for (size_t i = 0; i < hoff_row.size(); i++) f(hoff_row[i], hoff_col[i]);
  };

  // ---- fill-reducing ordering of the reduced variables (reverse Cuthill-McKee on the block graph) -------------
  // The reference gets its elimination order from COLAMD (inference/Ordering.cpp:42-124) unless the user passes
  // one; here the order only decides where each camera/pose block sits in S.  A banded / loop-closing block
  // pattern then leaves most 128x128 tiles of the factor empty, and the tile schedule skips them.
  // ---- ordering of the reduced variables + Cholesky schedule.  RCM gives ONE serial chain of diagonal tiles; nested dissection
  // (GTG_ND_DEPTH=n levels; default: 2 levels where the criterion below holds, 0 switches it off) gives the elimination tree independent
  // subtrees, which the dataflow kernels run as several chains side by side (chol_dataflow.hip::build_df_plan) -- the reference's
  // parallel elimination of independent cliques (inference/ClusterTree-inst.h:218-317).  It pays in the latency-bound regime only:
  // the separators cost fill (L1723 shape: +60 % flops for a 30 % shorter chain), so it is TRIED where the block structure is very
  // sparse (pose graphs: < 1 % of the reduced system's entries) and KEPT where the longest chain gets >= 30 % shorter. ----
  const char* nd_env = std::getenv("GTG_ND_DEPTH");
  const bool nd_forced = nd_env != nullptr;
  int nd_auto = 0;
  const char* ord_req = std::getenv("GTG_ORDERING");   // an explicitly requested ordering method (rcm, mindegree, auto; natural = the caller's order) is one chain
  const bool keep_order = ord_req && std::string(ord_req) == "natural";
  if (!nd_forced && hi.user_order.empty() && c.n_red >= 24 * (int64_t)kTile && !ord_req) {
    double nnz = 0.0;
    for_each_block([&](int ra, int rb) { if (ra != rb) nnz += 2.0 * c.h_red_dim[ra] * c.h_red_dim[rb]; });
    // 16 leaves on 8 chain slots (round 4, with the accumulator lanes of chol_dataflow.hip): sphere2500 1.07 ms, w20000 2.70 ms
    // (3 levels on 4 slots, round 3: 1.38 / 5.05; 2 levels: 1.78; one chain: 4.6 / 18.4; 5 levels: no better)
    if (nnz <= 0.01 * (double)c.n_red * (double)c.n_red) nd_auto = 4;
  }
  join_block_level(c);      // the flop count of a previous analysis of this handle (below) may still be running
  for (int attempt = 0; attempt < 2; attempt++) {
  join_block_level(c);
  const int nd_depth_try = attempt == 0 ? (nd_env ? std::atoi(nd_env) : nd_auto) : 0;
  // (no nested dissection = one chain = two chain workgroups: that plan's masked stream pair can be on its way before the ordering)
  if (nd_depth_try == 0 && kernels_can_run && dataflow_schedule_selected()) df_prepare_streams_async(c.device, 2);
  bool retry_rcm = false;
  std::vector<int32_t> part_of_pos;          // nested-dissection part of every position (empty: one part)
  std::vector<int32_t> part_parent;          // parent part (-1: root) of every part, parts numbered in elimination order
  if (hi.user_order.empty() && c.n_red_vars >= 16 && !keep_order) {
    const int nrv2 = c.n_red_vars;
    std::vector<std::vector<int32_t>> adj(nrv2);
    bool have_adj = false;
    auto ensure_adj = [&] {   // the host's adjacency lists (sorted, unique): not built when the ordering runs on the device
      if (have_adj) return;
      auto edge = [&](int a, int b) { if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } };
      for_each_block(edge);
      for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
      have_adj = true;
    };
    std::vector<int32_t> level(nrv2, -1);
    std::vector<char> active(nrv2, 0);     // node belongs to the subgraph being processed and is not ordered yet
    auto bfs_levels = [&](int start, std::vector<int32_t>& q) {   // BFS over active nodes, fills level[], returns order
      q.assign(1, start);
      for (int32_t v = 0; v < nrv2; v++) level[v] = -1;
      level[start] = 0;
      for (size_t h = 0; h < q.size(); h++)
        for (int32_t w : adj[q[h]]) if (active[w] && level[w] < 0) { level[w] = level[q[h]] + 1; q.push_back(w); }
    };
    auto far_node = [&](int start) {
      std::vector<int32_t> q; bfs_levels(start, q);
      int best = q.back();
      for (int32_t v : q) if (level[v] == level[q.back()] && adj[v].size() < adj[best].size()) best = v;
      return best;
    };
    // reverse Cuthill-McKee of a node set (all its components)
    auto rcm = [&](const std::vector<int32_t>& nodes, std::vector<int32_t>& out) {
      for (int32_t v : nodes) active[v] = 1;
      std::vector<int32_t> ord; ord.reserve(nodes.size());
      for (int32_t seed : nodes) {
        if (!active[seed]) continue;
        const int start = far_node(far_node(seed));   // two sweeps towards a pseudo-peripheral node of the component
        std::vector<int32_t> q{start}; active[start] = 0;
        for (size_t h = 0; h < q.size(); h++) {
          std::vector<int32_t> nb;
          for (int32_t w : adj[q[h]]) if (active[w]) { active[w] = 0; nb.push_back(w); }
          std::sort(nb.begin(), nb.end(), [&](int32_t a, int32_t b) { return adj[a].size() < adj[b].size() || (adj[a].size() == adj[b].size() && a < b); });
          q.insert(q.end(), nb.begin(), nb.end());
        }
        ord.insert(ord.end(), q.begin(), q.end());
      }
      std::reverse(ord.begin(), ord.end());
      out.insert(out.end(), ord.begin(), ord.end());
    };
    // Nested dissection by level-set separators: the elimination tree gets independent subtrees, i.e. the tile
    // Cholesky gets several serial chains that run side by side instead of one (cholesky.hip).  A separator is the
    // smallest BFS level (from a pseudo-peripheral node) that leaves at least a quarter of the nodes on each side.
    struct PartRec { std::vector<int32_t> nodes; int parent; };
    std::vector<PartRec> parts;
    std::function<int(const std::vector<int32_t>&, int)> dissect = [&](const std::vector<int32_t>& nodes, int depth) -> int {
      // returns the index of the part that roots this subtree (its last part in elimination order)
      if (depth > 0 && nodes.size() >= 256) {
        for (int32_t v : nodes) active[v] = 1;
        std::vector<int32_t> q;
        const int start = far_node(far_node(nodes[0]));
        bfs_levels(start, q);
        const int L = level[q.back()];
        std::vector<int64_t> cnt(L + 2, 0);
        for (int32_t v : q) cnt[level[v]]++;
        const int64_t n = (int64_t)nodes.size();
        int best = -1; int64_t below = 0;
        std::vector<int64_t> pre(L + 2, 0);
        for (int l = 0; l <= L; l++) pre[l + 1] = pre[l] + cnt[l];
        // balance first (the parts become chains of diagonal tiles that run side by side: the longest one is the critical path), then
        // the smallest separator among the balanced cuts; a cut that leaves a quarter on each side is the fallback
        for (int64_t share : {10, 8, 5})
          if (best < 0)
            for (int l = 1; l < L; l++) {
              below = pre[l];
              const int64_t above = n - pre[l + 1];   // nodes not reached by the BFS count as "above"
              if (std::min(below, above) * 20 < n * share) continue;
              if (best < 0 || cnt[l] < cnt[best]) best = l;
            }
        for (int32_t v : nodes) active[v] = 0;
        if (best > 0 && cnt[best] * 3 < n) {
          std::vector<int32_t> A, Bn, Sn;
          for (int32_t v : nodes) {
            if (level[v] >= 0 && level[v] < best) A.push_back(v);
            else if (level[v] == best) Sn.push_back(v);
            else Bn.push_back(v);
          }
          const int ra = dissect(A, depth - 1);
          const int rb = dissect(Bn, depth - 1);
          PartRec sp; sp.parent = -1;
          rcm(Sn, sp.nodes);
          parts.push_back(std::move(sp));
          const int me = (int)parts.size() - 1;
          parts[ra].parent = me; parts[rb].parent = me;
          return me;
        }
      }
      PartRec leaf; leaf.parent = -1;
      rcm(nodes, leaf.nodes);
      parts.push_back(std::move(leaf));
      return (int)parts.size() - 1;
    };
    const int nd_depth = nd_depth_try;
    std::vector<int32_t> all(nrv2);
    for (int i = 0; i < nrv2; i++) all[i] = i;
    // One part (no nested dissection) on a single shard with a real runtime: reverse Cuthill-McKee runs on the DEVICE
    // (device_ordering.hip: the same ordering position for position; GTG_HOST_ORDERING=1 keeps the host queue as the A/B).
    bool ordered_on_device = false;
    {
      const bool plain_rcm = !ord_req || std::string(ord_req) == "rcm";
      if (nd_depth == 0 && plain_rcm && kernels_can_run && c.n_shards == 1 && !std::getenv("GTG_HOST_ORDERING")) {
        StageClock sub;
        std::vector<int32_t> ea, eb;
        const bool edges_on_device = device_terms && hoff_row.empty();    // the device's own block list is the edge list
        if (!edges_on_device) { for_each_block([&](int a, int b) { if (a != b) { ea.push_back(a); eb.push_back(b); } }); sub.lap("  (ordering: edge list, host)"); }
        PartRec leaf; leaf.parent = -1;
        if (device_rcm(c, nrv2, ea, eb, leaf.nodes, edges_on_device ? c.pair_row.p : nullptr, edges_on_device ? c.pair_col.p : nullptr, (int64_t)pair_row.size())) {
          parts.push_back(std::move(leaf)); ordered_on_device = true;
        }
        sub.lap("  (ordering: device RCM)");
      }
    }
    if (!ordered_on_device) { ensure_adj(); dissect(all, nd_depth); }
    if (clk.on) std::fprintf(stderr, "[gtsam_amd setup] ordering computed on the %s\n", ordered_on_device ? "device" : "host");
    std::vector<int32_t> order; order.reserve(nrv2);
    for (size_t pi = 0; pi < parts.size(); pi++) {
      for (int32_t v : parts[pi].nodes) { order.push_back(v); part_of_pos.push_back((int32_t)pi); }
      part_parent.push_back(parts[pi].parent);
    }
    if (parts.size() == 1) { part_of_pos.clear(); part_parent.clear(); }
    // ---- minimum-degree alternative (GTG_ORDERING=mindegree | auto; default rcm).  The reference orders with COLAMD
    // (inference/Ordering.cpp:42-124), a minimum-degree method; on a reduced camera system it competes with the band ordering
    // above.  Both are scored by the block-level flops of the symbolic factorisation; "auto" keeps the cheaper one.  Measured on
    // the L1723 shape (cameras on a closed path: a cyclic band): RCM 136 GFLOP, minimum degree 145 GFLOP, natural order 782 --
    // the band wins there and its tiles are dense, so RCM stays the default and the 0.2 s of this search are opt-in.
    const std::string ord_mode = ord_req ? ord_req : "rcm";
    if (parts.size() == 1 && (ord_mode == "mindegree" || ord_mode == "auto")) {
      ensure_adj();
      auto block_flops = [&](const std::vector<int32_t>& ord) {
        std::vector<int32_t> pos(nrv2);
        for (int i = 0; i < nrv2; i++) pos[ord[i]] = i;
        std::vector<std::vector<int32_t>> below(nrv2), kids(nrv2);
        for (int v = 0; v < nrv2; v++) for (int32_t w : adj[v]) if (pos[w] > pos[v]) below[pos[v]].push_back(pos[w]);
        std::vector<int32_t> merged;
        double fl = 0.0;
        for (int j = 0; j < nrv2; j++) {
          auto& sj = below[j];
          std::sort(sj.begin(), sj.end()); sj.erase(std::unique(sj.begin(), sj.end()), sj.end());
          for (int32_t ch : kids[j]) {
            merged.clear();
            std::set_union(sj.begin(), sj.end(), below[ch].begin(), below[ch].end(), std::back_inserter(merged));
            merged.erase(std::remove(merged.begin(), merged.end(), (int32_t)j), merged.end());
            sj.swap(merged); std::vector<int32_t>().swap(below[ch]);
          }
          double sdim = 0.0;
          for (int32_t q : sj) sdim += c.h_red_dim[ord[q]];
          const double f = c.h_red_dim[ord[j]];
          fl += f * f * f / 3.0 + f * f * sdim + f * sdim * sdim;
          if (!sj.empty()) kids[sj.front()].push_back(j);
        }
        return fl;
      };
      auto min_degree = [&]() {   // plain minimum degree on the elimination graph (ties: lowest index): what COLAMD approximates
        std::vector<std::set<int32_t>> g(nrv2);
        for (int v = 0; v < nrv2; v++) g[v].insert(adj[v].begin(), adj[v].end());
        std::set<std::pair<int32_t, int32_t>> heap;
        for (int v = 0; v < nrv2; v++) heap.emplace((int32_t)g[v].size(), v);
        std::vector<int32_t> ord; ord.reserve(nrv2);
        while (!heap.empty()) {
          const int v = heap.begin()->second; heap.erase(heap.begin());
          ord.push_back(v);
          const std::vector<int32_t> nb(g[v].begin(), g[v].end());
          for (int32_t w : nb) { heap.erase({(int32_t)g[w].size(), w}); g[w].erase(v); }
          for (int32_t w : nb) for (int32_t u : nb) if (u != w) g[w].insert(u);
          for (int32_t w : nb) heap.emplace((int32_t)g[w].size(), w);
          std::set<int32_t>().swap(g[v]);
        }
        return ord;
      };
      // What the kernels execute is decided at TILE granularity: whole 128 x 128 tiles, fill included.  "auto" therefore scores the two
      // orderings by the flops over the tiles they would store (the count of chol_dataflow.hip::build_df_plan), not by the block-level
      // count: on the street-network shape (datasets.py::synthetic_bal_streets, random long-range loop closures) minimum degree needs
      // FEWER flops at block level (42 against 57 GFLOP) and 12 x MORE at tile level (1 170 against 101: its fill is scattered, 96 % of
      // the tiles are touched) -- the first version of "auto" compared block-level counts and picked it.
      auto tile_flops = [&](const std::vector<int32_t>& ord) {
        std::vector<int64_t> off(nrv2 + 1, 0);
        std::vector<int32_t> pos(nrv2);
        for (int i = 0; i < nrv2; i++) { pos[ord[i]] = i; off[i + 1] = off[i] + c.h_red_dim[ord[i]]; }
        const int ntl = (int)((off[nrv2] + kTile - 1) / kTile);
        std::vector<uint8_t> Bt((size_t)ntl * ntl, 0);
        auto mark = [&](int pa, int pb) {
          for (int64_t a = off[pa] / kTile; a <= (off[pa + 1] - 1) / kTile; a++)
            for (int64_t b = off[pb] / kTile; b <= (off[pb + 1] - 1) / kTile; b++) Bt[(size_t)std::max(a, b) * ntl + std::min(a, b)] = 1;
        };
        for (int v = 0; v < nrv2; v++) { mark(pos[v], pos[v]); for (int32_t w : adj[v]) if (pos[w] < pos[v]) mark(pos[v], pos[w]); }
        for (int k = 0; k < ntl; k++) {
          std::vector<int> r;
          for (int i = k + 1; i < ntl; i++) if (Bt[(size_t)i * ntl + k]) r.push_back(i);
          for (size_t a = 0; a < r.size(); a++) for (size_t b = 0; b <= a; b++) Bt[(size_t)r[a] * ntl + r[b]] = 1;
        }
        const double t3 = (double)kTile * kTile * kTile;
        double fl = 0.0;
        std::vector<int> cnt(ntl, 0);
        for (int i = 0; i < ntl; i++) for (int k = 0; k < i; k++) cnt[i] += Bt[(size_t)i * ntl + k];
        for (int J = 0; J < ntl; J++) {
          fl += t3 / 3.0 + cnt[J] * t3;
          for (int I = J + 1; I < ntl; I++) {
            if (!Bt[(size_t)I * ntl + J]) continue;
            int both = 0;
            for (int k = 0; k < J; k++) both += Bt[(size_t)I * ntl + k] & Bt[(size_t)J * ntl + k];
            fl += t3 + 2.0 * both * t3;
          }
        }
        return fl;
      };
      const std::vector<int32_t> md = min_degree();
      const double f_rcm = block_flops(order), f_md = block_flops(md);
      const double t_rcm = tile_flops(order), t_md = tile_flops(md);
      if (clk.on) std::fprintf(stderr, "[gtsam_amd setup] ordering: GFLOP at block level rcm %.1f, minimum degree %.1f; over 128 x 128 tiles rcm %.1f, minimum degree %.1f (%s)\n",
                               f_rcm / 1e9, f_md / 1e9, t_rcm / 1e9, t_md / 1e9, ord_mode.c_str());
      if (ord_mode == "mindegree" || t_md < t_rcm) order = md;
    }
    for (int i = 0; i < nrv2; i++) { c.h_red_pos[order[i]] = i; pos_to_red[i] = order[i]; }
    // offsets: every part starts on a 256-column pair boundary, so that a pair of block columns belongs to one part
    int64_t o2 = 0;
    c.h_pad_index.clear();
    for (int pp = 0; pp < nrv2; pp++) {
      if (!part_of_pos.empty() && (pp == 0 || part_of_pos[pp] != part_of_pos[pp - 1]))
        while (o2 % (2 * kTile)) c.h_pad_index.push_back(o2++);
      const int r = pos_to_red[pp]; c.h_red_off[r] = o2; o2 += c.h_red_dim[r];
    }
    const int64_t align = part_of_pos.empty() ? kTile : 2 * kTile;
    while (o2 % align) c.h_pad_index.push_back(o2++);
    c.NP = (int)o2;
    // re-orient the blocks: the row variable is the one placed later
    // (device-built lists: c.pair_row / c.pair_col and the terms are re-oriented in place by one kernel; the host's copy of the block list
    // keeps the orientation it has -- every host pass that still reads it looks at the positions itself)
    if (device_terms) device_orient_blocks(c, (int64_t)pair_row.size(), c.h_red_pos);
    else
      for (size_t i = 0; i < pair_row.size(); i++)
        if (c.h_red_pos[pair_row[i]] < c.h_red_pos[pair_col[i]]) {
          std::swap(pair_row[i], pair_col[i]);
          for (int64_t t = pair_ptr[i]; t < pair_ptr[i + 1]; t++) std::swap(pair_oa[t], pair_ob[t]);
        }
    for (size_t i = 0; i < hoff_row.size(); i++)
      if (c.h_red_pos[hoff_row[i]] < c.h_red_pos[hoff_col[i]]) {
        std::swap(hoff_row[i], hoff_col[i]);
        for (int64_t t = hoff_ptr[i]; t < hoff_ptr[i + 1]; t++) hoff_fac[t] ^= (1 << 30);
      }
    if (clk.on) {
      std::fprintf(stderr, "[gtsam_amd setup] nested dissection: %zu parts:", parts.size());
      for (size_t pi = 0; pi < parts.size(); pi++) std::fprintf(stderr, " %zu(^%d)", parts[pi].nodes.size(), parts[pi].parent);
      std::fprintf(stderr, "\n");
    }
    clk.lap("ordering (nested dissection + RCM)");
  } else {
    c.h_pad_index.clear();
    for (int64_t i = c.n_red; i < c.NP; i++) c.h_pad_index.push_back(i);
  }
  up(c.pad_index, c.h_pad_index, s);

  // ---- block-level symbolic factorisation: the flops the elimination needs at the granularity of the variables (d x d
  // blocks, fill included) -- sum over block columns of f^3/3 + f^2 s + f s^2 (f = the variable's dimension, s = the dimension
  // of its below-diagonal structure), the count SURVEY section 8(d) asks for next to the stored-tile count the kernels execute.
  // It only feeds a getter (gtg_cholesky_flops_block_level), so it runs on a host thread of the HANDLE, on its own copies of the block
  // list and the ordering, and is joined by whoever needs it next: the getter, the next analysis of the handle, gtg_destroy (round 5:
  // analyze() used to wait 1.3 ms for it on the L1723 shape).
  {
  // (its own copies of the block list: plain vector copies -- the loop that turns them into positions runs on the thread as well)
  std::vector<int32_t> bl_a, bl_b;                      // the off-diagonal blocks (reduced indices)
  if (c.n_shards > 1) { bl_a = sb_row; bl_b = sb_col; }
  else { bl_a = pair_row; bl_b = pair_col; bl_a.insert(bl_a.end(), hoff_row.begin(), hoff_row.end()); bl_b.insert(bl_b.end(), hoff_col.begin(), hoff_col.end()); }
  std::vector<int32_t> dim_at_pos(c.n_red_vars);
  for (int r = 0; r < c.n_red_vars; r++) dim_at_pos[c.h_red_pos[r]] = c.h_red_dim[r];
  gtg_context* cp = &c;
  c.block_level_err = nullptr;
  c.chol_flops_block = 0.0;      // (never the previous analysis's number while the new count runs)
  c.block_level_thread = std::thread([cp, bl_a = std::move(bl_a), bl_b = std::move(bl_b), dim_at = std::move(dim_at_pos), red_pos = c.h_red_pos] { try {
    const int n = (int)dim_at.size();
    std::vector<std::vector<int32_t>> below(n);          // positions > own position
    for (size_t i = 0; i < bl_a.size(); i++) {
      if (bl_a[i] == bl_b[i]) continue;
      const int pa = red_pos[bl_a[i]], pb = red_pos[bl_b[i]];
      below[std::min(pa, pb)].push_back(std::max(pa, pb));
    }
    std::vector<std::vector<int32_t>> children(n);
    std::vector<int32_t> merged;
    double fl = 0.0;
    for (int j = 0; j < n; j++) {
      std::vector<int32_t>& sj = below[j];
      std::sort(sj.begin(), sj.end()); sj.erase(std::unique(sj.begin(), sj.end()), sj.end());
      for (int32_t ch : children[j]) {                   // struct(j) |= struct(child) \ {j}
        merged.clear();
        std::set_union(sj.begin(), sj.end(), below[ch].begin(), below[ch].end(), std::back_inserter(merged));
        merged.erase(std::remove(merged.begin(), merged.end(), (int32_t)j), merged.end());
        sj.swap(merged);
        std::vector<int32_t>().swap(below[ch]);
      }
      double sdim = 0.0;
      for (int32_t q : sj) sdim += dim_at[q];
      const double f = dim_at[j];
      fl += f * f * f / 3.0 + f * f * sdim + f * sdim * sdim;
      if (!sj.empty()) children[sj.front()].push_back(j);
    }
    cp->chol_flops_block = fl;
  } catch (...) { cp->block_level_err = std::current_exception(); } });
  }

  // ---- tile structure of the reduced system -> Cholesky schedule ------------------------------------------------
  {
    const int nt = c.NP / kTile, np2 = (nt + 1) / 2;
    std::vector<uint8_t> B2((size_t)np2 * np2, 0);      // the same structure over 256-wide column pairs (filled from T1 below)
    std::vector<int32_t> pair_part;
    if (!part_of_pos.empty()) {
      pair_part.assign(np2, -1);
      for (int pp = 0; pp < c.n_red_vars; pp++) pair_part[c.h_red_off[pos_to_red[pp]] / (2 * kTile)] = part_of_pos[pp];
      for (int q = 0; q < np2; q++) if (pair_part[q] < 0) throw std::runtime_error("nested dissection: a column pair without variables");
    }
    {  // tiles that hold something before the factorisation: diagonal blocks, pose-pose blocks, Schur pairs, rhs row
      std::vector<uint8_t> T1((size_t)nt * nt, 0);
      std::vector<uint8_t> rhs((size_t)nt, 0);
      // the same structure at the granularity of 16 rows / columns (kSub): column c of the strip matrix = a bit set over the strips below it.
      // Its symbolic factorisation (below) tells the bulk kernel which 16 x 16 sub-tiles of an operand tile are structurally zero.
      const int n16 = c.NP / kSub, w16 = (n16 + 63) / 64;
      std::vector<uint64_t> M16((size_t)n16 * w16, 0);
      auto mark16 = [&](int64_t ra0, int64_t ra1, int64_t rb0, int64_t rb1) {
        for (int64_t a = ra0; a <= ra1; a++)
          for (int64_t b = rb0; b <= rb1; b++) { const int64_t hi = std::max(a, b), lo = std::min(a, b); M16[(size_t)lo * w16 + (hi >> 6)] |= 1ull << (hi & 63); }
      };
      auto mark1 = [&](int ra, int rb) {
        mark16(c.h_red_off[ra] / kSub, (c.h_red_off[ra] + c.h_red_dim[ra] - 1) / kSub, c.h_red_off[rb] / kSub, (c.h_red_off[rb] + c.h_red_dim[rb] - 1) / kSub);
        const int64_t a0 = c.h_red_off[ra] / kTile, a1 = (c.h_red_off[ra] + c.h_red_dim[ra] - 1) / kTile;
        const int64_t b0 = c.h_red_off[rb] / kTile, b1 = (c.h_red_off[rb] + c.h_red_dim[rb] - 1) / kTile;
        for (int64_t a = a0; a <= a1; a++)
          for (int64_t b = b0; b <= b1; b++) T1[(size_t)std::max(a, b) * nt + std::min(a, b)] = 1;
        for (int64_t a = a0; a <= a1; a++) rhs[(size_t)a] = 1;
      };
      // ONE pass over the blocks (203 k on the L1723 shape) marks the tiles and feeds the identity of the block set (below)
      // the block SET identifies the layout (a commutative sum over the blocks: a sharded handle lists them in bitmap order, a single
      // one in term order); only a sharded handle, or one whose pose pairs can carry two kinds of blocks, needs the list itself
      uint64_t hb = 0;
      auto mix_block = [&](int64_t ro, int64_t co, int32_t dd) {
        uint64_t z = (uint64_t)ro * 0x9E3779B97F4A7C15ull ^ ((uint64_t)co + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)dd;
        z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        hb += z;
      };
      std::vector<int64_t> xro, xco; std::vector<int32_t> xd;
      const bool need_list = c.n_shards > 1 || (!pair_row.empty() && !hoff_row.empty());
      auto add_block = [&](int64_t ro, int64_t co, int32_t dd) { if (need_list) { xro.push_back(ro); xco.push_back(co); xd.push_back(dd); } else mix_block(ro, co, dd); };
      // The marks of the Schur blocks come from the DEVICE where the block list is its own (single shard, real runtime: one lane per
      // block, device_analysis.hip::k_da_tile_marks, over the device's own re-oriented block list; 3 ms of this loop on the L1723 shape).
      // GTG_HOST_SYMBOLIC=1 keeps the host loop as the A/B (tests/test_gpu_device_analysis.py: same hash, same masks).
      const bool marks_on_device = device_terms && !need_list && pair_row.size() >= 4096 && !std::getenv("GTG_HOST_SYMBOLIC");
      if (marks_on_device) {
        std::vector<RedLayout> lay(c.n_red_vars);
        for (int r = 0; r < c.n_red_vars; r++) lay[r] = RedLayout{c.h_red_off[r], c.h_red_dim[r]};
        device_tile_marks(c, (int64_t)pair_row.size(), lay, nt, n16, w16, T1, M16, &hb);
      }
      for (int r = 0; r < c.n_red_vars; r++) { mark1(r, r); add_block(c.h_red_off[r], c.h_red_off[r], c.h_red_dim[r] | (c.h_red_dim[r] << 8)); }
      if (!marks_on_device) for_each_block([&](int ra, int rb) {
        mark1(ra, rb);
        if (ra == rb) return;                                  // a camera's Schur terms with itself: the diagonal block above
        const bool a_later = c.h_red_pos[ra] > c.h_red_pos[rb];
        const int rr = a_later ? ra : rb, rc = a_later ? rb : ra;
        add_block(c.h_red_off[rr], c.h_red_off[rc], c.h_red_dim[rr] | (c.h_red_dim[rc] << 8));
      });
      for (int a = 0; a < nt; a++)
        for (int b2 = 0; b2 <= a; b2++) if (T1[(size_t)a * nt + b2]) B2[(size_t)(a / 2) * np2 + (size_t)(b2 / 2)] = 1;
      for (int64_t i : c.h_pad_index) T1[(size_t)(i / kTile) * nt + (size_t)(i / kTile)] = 1;
      clk.lap("tile structure (marks of every block)");
      // Symbolic factorisation of the strip matrix: the structure of column c of L is that of S plus the structures of its children in the
      // elimination tree (parent = first strip below the diagonal), so every column is merged into ONE other column.  sub16[I nt + J], I >= J:
      // bit 8 r + q = the 16 x 16 sub-tile (r, q) of tile (I, J) of L can be non-zero.
      std::vector<uint64_t> sub16((size_t)nt * nt, 0);
      {
        for (int q = 0; q < n16; q++) M16[(size_t)q * w16 + (q >> 6)] |= 1ull << (q & 63);
        for (int q = 0; q < n16; q++) {
          const uint64_t* col = M16.data() + (size_t)q * w16;
          int parent = -1;
          for (int w = (q + 1) >> 6; w < w16 && parent < 0; w++) {
            uint64_t bits = col[w];
            if (w == ((q + 1) >> 6)) bits &= ~0ull << ((q + 1) & 63);
            if (bits) parent = w * 64 + __builtin_ctzll(bits);
          }
          if (parent < 0) continue;
          uint64_t* pc = M16.data() + (size_t)parent * w16;
          for (int w = parent >> 6; w < w16; w++) { uint64_t bits = col[w]; if (w == (parent >> 6)) bits &= ~0ull << (parent & 63); pc[w] |= bits; }
        }
        constexpr int per = kTile / kSub;
        for (int q = 0; q < n16; q++) {
          const uint64_t* col = M16.data() + (size_t)q * w16;
          const int J = q / per, cq = q % per;
          for (int w = q >> 6; w < w16; w++) {
            uint64_t bits = col[w];
            while (bits) {
              const int r16 = w * 64 + __builtin_ctzll(bits); bits &= bits - 1;
              if (r16 < q) continue;
              sub16[(size_t)(r16 / per) * nt + J] |= 1ull << (8 * (r16 % per) + cq);
            }
          }
        }
      }
      clk.lap("strip-level symbolic factorisation (sub-tile masks)");
      std::vector<int32_t> ex;
      const bool dense = std::getenv("GTG_DENSE_PLAN") != nullptr;
      // default schedule: the dataflow pass (chol_dataflow.hip), symbolic fill at 128-tile granularity.  GTG_CHOL=streams
      // selects the per-column launch sequence of cholesky.hip; the elimination-tree schedule only exists there.
      c.use_df = dataflow_schedule_selected();
      free_df_plan(c.df);
      std::vector<int32_t> tile_part;                      // nested dissection: the part of every block column (parts are aligned to column pairs)
      if (!pair_part.empty()) { tile_part.resize(nt); for (int t = 0; t < nt; t++) tile_part[t] = pair_part[t / 2]; }
      // The two tile schedules are built side by side: the task lists of the dataflow pass on a thread of their own (pure host code),
      // the stream schedule's lists -- which also number the stored tiles (slots) -- here; the dataflow plan is resolved to slots
      // and uploaded when both are done.
      {
        std::thread dfh; std::exception_ptr dferr;
        // (the dataflow lists -- the longer of the two, thousands of small allocations -- on THIS thread, whose allocator arena is warm: on a
        // fresh thread the same code took 0.35 ms longer for its page faults; the stream schedule's slots and backward lists, 0.1 ms since
        // its pair lists are built on demand, and their uploads on the second one)
        double sp_ms = 0.0;
        const int dev = c.device;
        dfh = std::thread([&] { try { StageClock k; check_hip(hipSetDevice(dev), "hipSetDevice"); build_chol_plan(c.plan, nt, dense ? nullptr : &B2, s, &pair_part, &part_parent);
                                      sp_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - k.t).count(); } catch (...) { dferr = std::current_exception(); } });
        try { StageClock k; if (c.use_df) build_df_plan_host(c.df, nt, dense ? nullptr : &T1, &tile_part, &part_parent); k.lap("  (dataflow task lists, this thread)"); }
        catch (...) { if (dfh.joinable()) dfh.join(); throw; }
        if (dfh.joinable()) dfh.join();
        if (dferr) std::rethrow_exception(dferr);
        if (clk.on) std::fprintf(stderr, "[gtsam_amd setup]   (stream schedule's slots + backward lists, second thread) %8.2f ms\n", sp_ms);
        clk.lap("tile schedules (task lists of both passes, two threads)");
        if (c.use_df) upload_df_plan(c.df, s, c.plan.h_slot, c.plan.n_stored, dense ? nullptr : &sub16, kernels_can_run ? c.plan.slot.p : nullptr);
        clk.lap("dataflow plan resolved to slots + uploaded");
      }
      for (int a = 0; a < nt; a++)
        for (int b = 0; b <= a; b++) if (dense || T1[(size_t)a * nt + b]) { ex.push_back(a); ex.push_back(b); }
      for (int b = 0; b < nt; b++) if (dense || rhs[(size_t)b]) { ex.push_back(nt); ex.push_back(b); }
      c.plan.n_exch = (int64_t)ex.size() / 2;
      if (ex.empty()) { ex.push_back(0); ex.push_back(0); }
      up(c.plan.exch, ex, s);
      // identity of the layout of the reduced system (every shard of a job must arrive at the same one)
      uint64_t h = 1469598103934665603ull;
      auto mix = [&](const void* ptr, size_t bytes) { const unsigned char* q = (const unsigned char*)ptr; for (size_t i = 0; i < bytes; i++) h = (h ^ q[i]) * 1099511628211ull; };
      const int64_t head[3] = {c.NP, c.n_red, (int64_t)c.n_red_vars};
      mix(head, sizeof(head)); mix(c.h_red_off.data(), c.h_red_off.size() * sizeof(int64_t));
      mix(c.h_pad_index.data(), c.h_pad_index.size() * sizeof(int64_t)); mix(B2.data(), B2.size());
      mix(ex.data(), ex.size() * sizeof(int32_t)); mix(pair_part.data(), pair_part.size() * sizeof(int32_t));
      // sharded: the block-granular exchange list (diagonal blocks, then the off-diagonal blocks of the whole graph)
      c.n_xb = 0;
      {
        if (c.n_shards > 1) { c.n_xb = (int64_t)xd.size(); up(c.xb_row_off, xro, s); up(c.xb_col_off, xco, s); up(c.xb_dim, xd, s); }
        else if (!pair_row.empty() && !hoff_row.empty()) {   // a pose pair can carry a Schur block AND a between block: one entry in the set
          std::vector<size_t> idx(xd.size());
          for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
          std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return xro[a] != xro[b] ? xro[a] < xro[b] : xco[a] < xco[b]; });
          std::vector<int64_t> r2, c2; std::vector<int32_t> d2;
          for (size_t k = 0; k < idx.size(); k++)
            if (k == 0 || xro[idx[k]] != xro[idx[k - 1]] || xco[idx[k]] != xco[idx[k - 1]]) { r2.push_back(xro[idx[k]]); c2.push_back(xco[idx[k]]); d2.push_back(xd[idx[k]]); }
          xro.swap(r2); xco.swap(c2); xd.swap(d2);
        }
        for (size_t i = 0; i < xd.size(); i++) mix_block(xro[i], xco[i], xd[i]);
        mix(&hb, sizeof(hb));
      }
      c.structure_hash = h;
    }
    clk.lap("exchange list + layout hash");
    if (clk.on) std::fprintf(stderr, "[gtsam_amd setup] reduced system n = %lld, %d tiles, stored tile fraction %.3f, %.3f GFLOP per factorisation, critical path %d of %d column pairs\n",
                             (long long)c.n_red, nt, c.plan.dense_fraction, c.plan.flops * 1e-9, c.plan.critical_pairs, np2);
    if (clk.on && c.use_df) std::fprintf(stderr, "[gtsam_amd setup] dataflow plan: %lld tasks, %.3f GFLOP over the stored tiles (fraction %.3f; %.3f GFLOP after skipping structurally empty 16 x 16 sub-tiles), %d chain workgroups, longest chain slot %d of %d diagonal tiles\n",
                                         (long long)c.df.n_tasks, c.df.flops * 1e-9, c.df.dense_fraction, c.df.flops_executed * 1e-9, c.df.n_chain, c.df.critical_tiles, nt);
    // keep the nested-dissection ordering only where it pays: the chains must get clearly shorter and the problem must be
    // in the latency-bound regime (separators cost fill: on the L1723 shape +60 % flops for a 30 % shorter path)
    if (!part_of_pos.empty() && !nd_forced) {
      const bool shorter = c.use_df ? c.df.critical_tiles * 10 <= nt * 7 : c.plan.critical_pairs * 10 <= np2 * 7;
      if (!(shorter && (c.use_df ? c.df.flops : c.plan.flops) <= 6e10)) retry_rcm = true;
    }
  }

  if (!retry_rcm) {
    // the plan stands: its masked stream pair is created on a helper thread from here on (chol_dataflow.hip::df_prepare_streams_async)
    if (c.use_df && kernels_can_run) df_prepare_streams_async(c.device, c.df.n_chain);
    break;
  }
  }

  // ---- upload -----------------------------------------------------------------------------------
  up(c.lm_var, c.h_lm_var, s); up(c.red_var, c.h_red_var, s); up(c.red_dim, c.h_red_dim, s);
  up(c.lm_index, c.h_lm_index, s); up(c.red_index, c.h_red_index, s); up(c.red_off, c.h_red_off, s);
  up(c.lm_owned, lm_owned, s);
  if (!device_terms) {   // (else built in place by device_incidence_lists)
    up(c.obs_red, obs_red, s); up(c.obs_lm, obs_lm, s);
    up(c.lm_obs_ptr, lm_obs_ptr, s); up(c.lm_obs, lm_obs, s);
    up(c.red_inc_ptr, inc_ptr, s); up(c.red_inc_kind, inc_kind, s); up(c.red_inc_idx, inc_idx, s);
  }
  up(c.lm_pri_ptr, lm_pri_ptr, s); up(c.lm_pri, lm_pri, s);
  up(c.hoff_row, hoff_row, s); up(c.hoff_col, hoff_col, s); up(c.hoff_ptr, hoff_ptr, s); up(c.hoff_fac, hoff_fac, s);
  if (!device_terms) { up(c.pair_row, pair_row, s); up(c.pair_col, pair_col, s); }   // (else: built and re-oriented in place by the device passes)
  if (!device_terms) {   // (the device built pair_ptr / pair_oa / pair_ob in place)
    up(c.pair_ptr, pair_ptr, s);
    c.pair_oa.upload(pair_oa.get(), (size_t)c.n_pair_terms, s); c.pair_ob.upload(pair_ob.get(), (size_t)c.n_pair_terms, s);
    if (c.n_pair_terms == 0) { c.pair_oa.alloc(1); c.pair_ob.alloc(1); }
  }

  // ---- numeric buffers --------------------------------------------------------------------------
  const size_t NP = c.NP;
  c.Hd.alloc(std::max<size_t>(81 * (size_t)c.n_red_vars, 81)); c.gred0.alloc(std::max<size_t>(9 * (size_t)c.n_red_vars, 9));
  c.hdiag_red.alloc(NP);
  c.V.alloc(std::max<size_t>(9 * (size_t)c.n_lm, 9)); c.gp.alloc(std::max<size_t>(3 * (size_t)c.n_lm, 3));
  c.Linv.alloc(std::max<size_t>(9 * (size_t)c.n_lm, 1)); c.ylm.alloc(std::max<size_t>(3 * (size_t)c.n_lm, 1));
  c.delta_lm.alloc(std::max<size_t>(3 * (size_t)c.n_lm, 1));
  c.E.alloc(std::max<size_t>(kEStride * (size_t)c.n_obs, 1));
  {
    const int64_t n_inc = (int64_t)c.red_inc_kind.n;
    if (n_inc >= ((int64_t)1 << 31)) throw std::runtime_error("analysis: more than 2^31 entries in the contribution lists");
    c.wobs.alloc(std::max<size_t>(9 * (size_t)n_inc, 1));
    c.obs_wpos.alloc(std::max<size_t>((size_t)c.n_obs, 1));
    launch_obs_wpos(c, n_inc);
  }
  c.vobs.alloc(std::max<size_t>(3 * (size_t)c.n_obs, 1));
  c.Hoff.alloc(std::max<size_t>(81 * (size_t)c.n_hoff, 1));
  c.S.alloc((size_t)(c.plan.n_stored + (c.use_df ? c.df.n_scratch : 0)) * kTileDoubles);   // the stored tiles only (context.h::SMat) + the dataflow plan's scratch slots
  c.Dinv.alloc((NP / kTile) * (size_t)kTile * kTile);
  check_hip(hipMemsetAsync(c.Dinv.p, 0, sizeof(double) * c.Dinv.n, c.stream), "memset");
  c.chol_epoch_dev.alloc(1);
  check_hip(hipMemsetAsync(c.chol_epoch_dev.p, 0, sizeof(long long), c.stream), "memset");
  c.chol_epoch = 0;   // (Dinv and the flag arrays of the dataflow plan are freshly zeroed as well)
  {  // where the reference's rank test applies in the reduced system: the last pivot of every variable
    std::vector<unsigned char> pk(std::max<size_t>(NP, 1), 0);
    for (int r = 0; r < c.n_red_vars; r++) pk[c.h_red_off[r] + c.h_red_dim[r] - 1] = c.h_red_dim[r] >= 2 ? 1 : 2;
    c.pivot_kind.upload(pk.data(), pk.size(), c.stream);
    c.tile_exp.alloc(NP / kTile + 1);
  }
  c.xred.alloc(2 * NP);   // + the shadow copies the backward sweep polls when an entry seems stuck (cholesky.hip::sweep_wait)
  c.partials.alloc(2 * 2048);
  c.scalars.alloc(3 * SC_COUNT);   // [SC_COUNT, 2 SC_COUNT): the copy the sharded exchange sums (read_scalars); [2 SC_COUNT, 3 SC_COUNT): the
                                   // summed linear errors in front of the gated kernels of a sharded smart graph (smart_gate)
  check_hip(hipMemsetAsync(c.scalars.p, 0, sizeof(double) * 2 * SC_COUNT, s), "memset");
  check_hip(hipMemsetAsync(c.hdiag_red.p, 0, sizeof(double) * NP, s), "memset");
  check_hip(hipMemsetAsync(c.xred.p, 0, sizeof(double) * NP, s), "memset");
  check_hip(hipMemsetAsync(c.delta_lm.p, 0, sizeof(double) * c.delta_lm.n, s), "memset");
  check_hip(hipStreamSynchronize(s), "sync");
  clk.lap("upload + device buffers");

  // sharded: the buffers the shards exchange only line up if every shard derived the same layout -- checked once, through
  // the exchange itself: now if the callback is already registered, else in front of the first exchange
  c.layout_probe.alloc(4);
  c.layout_verified = false;
  if (c.n_shards > 1 && c.allreduce) verify_layout(c);

  c.chol_flops = c.use_df ? c.df.flops : c.plan.flops;
  // algorithmic HBM bytes of one linearize+assemble pass: factor indices + measurements + variable blocks read, blocks
  // written; stored records: + the Jacobian records written and read once by the assembly.  GeneralSFM factors of a fused graph
  // (fused.h) have no stored records: SURVEY.md section 8(d)'s fused figure -- indices, measurement and noise row read, the
  // off-diagonal block W = Jc^T Jp written (216 B; here in its lambda-dependent form E, by the first point elimination).
  c.lin_bytes = (double)n_sfm * (2 * 4 + 16 + 4 + (c.fused_sfm ? 216.0 : 2.0 * kSfmRec * 8)) + (double)n_proj * (2 * 4 + 16 + 12 + 2.0 * kProjRec * 8) +
                (double)n_btw * (2 * 4 + 96 + 4 + 2.0 * kBetweenRec * 8) + (double)c.val_size * 8 +
                (double)c.n_red_vars * 90 * 8 + (double)c.n_lm * 12 * 8 + (double)c.n_hoff * 36 * 8;
}

// (the count is a diagnostic: a failure of its thread -- bad_alloc -- must not fail the NEXT, unrelated upload or ordering call that joins
// it; the stored exception is dropped and the count reads 0)
void join_block_level(gtg_context& c) {
  if (c.block_level_thread.joinable()) c.block_level_thread.join();
  if (c.block_level_err) { c.block_level_err = nullptr; c.chol_flops_block = 0.0; }
}

}  // namespace gt
