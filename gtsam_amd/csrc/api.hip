// api.hip -- the C ABI (include/gtsam_amd.h) and the one-time host-side symbolic analysis.
//
// Host work here is O(#factors) bookkeeping done ONCE per graph: landmark classification, CSR
// incidence lists, the block pattern of the Schur complement.  It replaces the VariableIndex /
// EliminationTree / JunctionTree / Scatter construction that the reference repeats on every lambda
// try (inference/VariableIndex-inl.h:27-49, EliminationTree-inst.h:77-155, JunctionTree-inst.h:63-151,
// linear/Scatter.cpp:39-73).  All arithmetic of the hot path runs in the HIP kernels.
#include <algorithm>
#include <atomic>
#include <exception>
#include <functional>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <sys/mman.h>
#include <mutex>
#include <stdexcept>

#include "factors.h"
#include "kernels.h"

namespace gt {

static thread_local std::string g_last_error;

void check_hip(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

template <class T> void DevBuf<T>::alloc(size_t count) {
  free();
  n = count;
  if (count) check_hip(hipMalloc(&p, sizeof(T) * count), "hipMalloc");
}
template <class T> void DevBuf<T>::upload(const T* host, size_t count, hipStream_t s) {
  if (count != n || (count && !p)) alloc(count);
  if (count) check_hip(hipMemcpyAsync(p, host, sizeof(T) * count, hipMemcpyHostToDevice, s), "H2D");
}
template <class T> void DevBuf<T>::free() {
  if (p) (void)hipFree(p);
  p = nullptr; n = 0;
}
template struct DevBuf<double>;
template struct DevBuf<int32_t>;
template struct DevBuf<int64_t>;
template struct DevBuf<long long>;

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
static std::vector<std::pair<gtg_context*, HostIndex*>> g_index;  // tiny registry (handles are few)
static std::mutex g_index_mutex;                                   // handles may be created / destroyed from several host threads
static HostIndex& host_index(gtg_context* c) {
  std::lock_guard<std::mutex> lock(g_index_mutex);
  for (auto& kv : g_index) if (kv.first == c) return *kv.second;
  g_index.emplace_back(c, new HostIndex);
  return *g_index.back().second;
}
static void drop_index(gtg_context* c) {
  std::lock_guard<std::mutex> lock(g_index_mutex);
  for (size_t i = 0; i < g_index.size(); i++)
    if (g_index[i].first == c) { delete g_index[i].second; g_index.erase(g_index.begin() + i); return; }
}

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

static void exchange(gtg_context& c, double* ptr, int64_t n);

// Three 20-bit pieces of the layout hash and a 1 go through the all-reduce: the sums must be n_shards times this shard's
// own values (every shard derived the same layout AND the communicator spans n_shards ranks).
static void verify_layout(gtg_context& c) {
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
static void analyze(gtg_context& c) {
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
  // validate factor roles
  for (int64_t i = 0; i < n_sfm; i++)
    if (c.h_var_type[hi.sfm_cam[i]] != GTG_VAR_SFM_CAMERA || c.h_var_type[hi.sfm_point[i]] != GTG_VAR_POINT3)
      throw std::invalid_argument("GeneralSFMFactor keys must be (SFM_CAMERA, POINT3)");
  for (int64_t i = 0; i < n_proj; i++)
    if (c.h_var_type[hi.proj_pose[i]] != GTG_VAR_POSE3 || c.h_var_type[hi.proj_point[i]] != GTG_VAR_POINT3)
      throw std::invalid_argument("GenericProjectionFactor keys must be (POSE3, POINT3)");
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
  std::vector<int32_t> obs_red(c.n_obs), obs_lm(c.n_obs);
  for (int64_t i = 0; i < n_sfm; i++) { obs_red[i] = c.h_red_index[hi.sfm_cam[i]]; obs_lm[i] = c.h_lm_index[hi.sfm_point[i]]; }
  for (int64_t i = 0; i < n_proj; i++) { obs_red[n_sfm + i] = c.h_red_index[hi.proj_pose[i]]; obs_lm[n_sfm + i] = c.h_lm_index[hi.proj_point[i]]; }

  // landmark -> observations / priors (CSR, factor order)
  std::vector<int64_t> lm_obs_ptr(c.n_lm + 1, 0), lm_pri_ptr(c.n_lm + 1, 0);
  for (int64_t o = 0; o < c.n_obs; o++) lm_obs_ptr[obs_lm[o] + 1]++;
  for (int64_t i = 0; i < n_pri; i++) { const int l = c.h_lm_index[hi.prior_var[i]]; if (l >= 0) lm_pri_ptr[l + 1]++; }
  for (int l = 0; l < c.n_lm; l++) { lm_obs_ptr[l + 1] += lm_obs_ptr[l]; lm_pri_ptr[l + 1] += lm_pri_ptr[l]; }
  std::vector<int32_t> lm_obs(c.n_obs), lm_pri(lm_pri_ptr[c.n_lm]);
  { std::vector<int64_t> w(lm_obs_ptr.begin(), lm_obs_ptr.end() - 1);
    for (int64_t o = 0; o < c.n_obs; o++) lm_obs[w[obs_lm[o]]++] = (int32_t)o; }
  { std::vector<int64_t> w(lm_pri_ptr.begin(), lm_pri_ptr.end() - 1);
    for (int64_t i = 0; i < n_pri; i++) { const int l = c.h_lm_index[hi.prior_var[i]]; if (l >= 0) lm_pri[w[l]++] = (int32_t)i; } }
  std::vector<int32_t> lm_owned(std::max(c.n_lm, 1), 0);
  for (int l = 0; l < c.n_lm; l++) lm_owned[l] = (l % c.n_shards) == c.shard;

  // reduced variable -> contributions
  std::vector<int64_t> inc_ptr(c.n_red_vars + 1, 0);
  auto count = [&](int v) { const int r = c.h_red_index[v]; if (r >= 0) inc_ptr[r + 1]++; };
  for (int64_t i = 0; i < n_sfm; i++) count(hi.sfm_cam[i]);
  for (int64_t i = 0; i < n_proj; i++) count(hi.proj_pose[i]);
  for (int64_t i = 0; i < n_btw; i++) { count(hi.between_v1[i]); count(hi.between_v2[i]); }
  for (int64_t i = 0; i < n_pri; i++) count(hi.prior_var[i]);
  for (int r = 0; r < c.n_red_vars; r++) inc_ptr[r + 1] += inc_ptr[r];
  std::vector<int32_t> inc_kind(inc_ptr[c.n_red_vars]), inc_idx(inc_ptr[c.n_red_vars]);
  { std::vector<int64_t> w(inc_ptr.begin(), inc_ptr.end() - 1);
    auto put = [&](int v, int kind, int64_t idx) { const int r = c.h_red_index[v]; if (r >= 0) { inc_kind[w[r]] = kind; inc_idx[w[r]++] = (int32_t)idx; } };
    for (int64_t i = 0; i < n_sfm; i++) put(hi.sfm_cam[i], 0, i);
    for (int64_t i = 0; i < n_proj; i++) put(hi.proj_pose[i], 1, i);
    for (int64_t i = 0; i < n_btw; i++) { put(hi.between_v1[i], 2, i); put(hi.between_v2[i], 3, i); }
    for (int64_t i = 0; i < n_pri; i++) put(hi.prior_var[i], 4, i); }

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

  clk.lap("incidence lists");
  // Schur block pairs: for every landmark, every pair of its observations is one term E_a E_b^T of the block
  // (row = the later position, column = the earlier one).  Terms are bucketed by the row position of their block
  // (counting sort), then every row bucket is sorted by column position (stable: the generation order = landmark
  // order is kept inside a block, so the summation order is reproducible).  All passes run on host threads and the
  // result does not depend on their number: a thread owns a contiguous range of landmarks and writes behind the
  // terms of the threads before it in every row bucket; rows are sorted independently.
  struct PT { int32_t pb, oa, ob; };
  const int nrv = c.n_red_vars;
  const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), c.n_obs / 16384));
  std::vector<int32_t> obs_pos(c.n_obs);
  for (int64_t o = 0; o < c.n_obs; o++) obs_pos[o] = c.h_red_pos[obs_red[o]];
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
  const int64_t n_terms = row_ptr[nrv];
  HugeBuf<PT> pt((size_t)std::max<int64_t>(n_terms, 1));                 // first touched by the writers
  run_threads(nth, [&](int t) { auto& w = cursor[t]; for_terms(lm_cut[t], lm_cut[t + 1], [&](int pa, int pb, int32_t oa, int32_t ob) { pt[w[pa]++] = PT{pb, oa, ob}; }); });
  clk.lap("schur terms bucketed");
  // per row bucket: stable counting sort by column position straight into the final term lists + the row's blocks
  HugeBuf<int32_t> pair_oa((size_t)std::max<int64_t>(n_terms, 1)), pair_ob((size_t)std::max<int64_t>(n_terms, 1));
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
  std::vector<int32_t> pair_row, pair_col;
  std::vector<int64_t> pair_ptr;
  { size_t nb = 0;
    for (int r = 0; r < nrv; r++) nb += row_blocks[r].col.size();
    pair_row.reserve(nb); pair_col.reserve(nb); pair_ptr.reserve(nb + 1);
    for (int r = 0; r < nrv; r++)
      for (size_t k = 0; k < row_blocks[r].col.size(); k++) {
        pair_row.push_back(pos_to_red[r]); pair_col.push_back(pos_to_red[row_blocks[r].col[k]]); pair_ptr.push_back(row_blocks[r].start[k]);
      }
    pair_ptr.push_back(n_terms); }
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
  // ---- ordering of the reduced variables + Cholesky schedule.  Default: RCM (one serial chain).  GTG_ND_DEPTH=n asks for
  // n levels of nested dissection (independent chains, tree schedule in cholesky.hip): correct, but measured slower or
  // equal on every workload of this round (sphere2500 5.3 -> 5.3..8.9 ms, w20000 21.8 -> 20.4..31.6 ms, L1723 +60 % flops),
  // because a chain is issued at ~45 us of host time per column pair and the separators cost fill. ----
  const char* nd_env = std::getenv("GTG_ND_DEPTH");
  const bool nd_forced = nd_env != nullptr;
  for (int attempt = 0; attempt < 2; attempt++) {
  const int nd_depth_try = attempt == 0 ? (nd_env ? std::atoi(nd_env) : 0) : 0;   // opt-in (GTG_ND_DEPTH=levels), see DESIGN.md
  bool retry_rcm = false;
  std::vector<int32_t> part_of_pos;          // nested-dissection part of every position (empty: one part)
  std::vector<int32_t> part_parent;          // parent part (-1: root) of every part, parts numbered in elimination order
  if (hi.user_order.empty() && c.n_red_vars >= 16 && !std::getenv("GTG_NO_REORDER")) {
    const int nrv2 = c.n_red_vars;
    std::vector<std::vector<int32_t>> adj(nrv2);
    auto edge = [&](int a, int b) { if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } };
    for_each_block(edge);
    for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
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
        for (int l = 1; l < L; l++) {
          below = pre[l];
          const int64_t above = n - pre[l + 1];   // nodes not reached by the BFS count as "above"
          if (std::min(below, above) * 4 < n) continue;
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
    dissect(all, nd_depth);
    std::vector<int32_t> order; order.reserve(nrv2);
    for (size_t pi = 0; pi < parts.size(); pi++) {
      for (int32_t v : parts[pi].nodes) { order.push_back(v); part_of_pos.push_back((int32_t)pi); }
      part_parent.push_back(parts[pi].parent);
    }
    if (parts.size() == 1) { part_of_pos.clear(); part_parent.clear(); }
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

  // ---- tile structure of the reduced system -> Cholesky schedule ------------------------------------------------
  {
    const int nt = c.NP / kTile, np2 = (nt + 1) / 2;
    std::vector<uint8_t> B2((size_t)np2 * np2, 0);
    auto mark = [&](int ra, int rb) {
      const int64_t a0 = c.h_red_off[ra] / (2 * kTile), a1 = (c.h_red_off[ra] + c.h_red_dim[ra] - 1) / (2 * kTile);
      const int64_t b0 = c.h_red_off[rb] / (2 * kTile), b1 = (c.h_red_off[rb] + c.h_red_dim[rb] - 1) / (2 * kTile);
      for (int64_t a = a0; a <= a1; a++)
        for (int64_t b = b0; b <= b1; b++) B2[(size_t)std::max(a, b) * np2 + std::min(a, b)] = 1;
    };
    for (int r = 0; r < c.n_red_vars; r++) mark(r, r);
    for_each_block(mark);
    std::vector<int32_t> pair_part;
    if (!part_of_pos.empty()) {
      pair_part.assign(np2, -1);
      for (int pp = 0; pp < c.n_red_vars; pp++) pair_part[c.h_red_off[pos_to_red[pp]] / (2 * kTile)] = part_of_pos[pp];
      for (int q = 0; q < np2; q++) if (pair_part[q] < 0) throw std::runtime_error("nested dissection: a column pair without variables");
    }
    build_chol_plan(c.plan, nt, std::getenv("GTG_DENSE_PLAN") ? nullptr : &B2, s, &pair_part, &part_parent);
    {  // tiles that hold something before the factorisation: diagonal blocks, pose-pose blocks, Schur pairs, rhs row
      std::vector<uint8_t> T1((size_t)nt * nt, 0);
      std::vector<uint8_t> rhs((size_t)nt, 0);
      auto mark1 = [&](int ra, int rb) {
        const int64_t a0 = c.h_red_off[ra] / kTile, a1 = (c.h_red_off[ra] + c.h_red_dim[ra] - 1) / kTile;
        const int64_t b0 = c.h_red_off[rb] / kTile, b1 = (c.h_red_off[rb] + c.h_red_dim[rb] - 1) / kTile;
        for (int64_t a = a0; a <= a1; a++)
          for (int64_t b = b0; b <= b1; b++) T1[(size_t)std::max(a, b) * nt + std::min(a, b)] = 1;
        for (int64_t a = a0; a <= a1; a++) rhs[(size_t)a] = 1;
      };
      for (int r = 0; r < c.n_red_vars; r++) mark1(r, r);
      for_each_block(mark1);
      for (int64_t i : c.h_pad_index) T1[(size_t)(i / kTile) * nt + (size_t)(i / kTile)] = 1;
      std::vector<int32_t> ex;
      const bool dense = std::getenv("GTG_DENSE_PLAN") != nullptr;
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
        std::vector<int64_t> xro, xco; std::vector<int32_t> xd;
        for (int r = 0; r < c.n_red_vars; r++) { xro.push_back(c.h_red_off[r]); xco.push_back(c.h_red_off[r]); xd.push_back(c.h_red_dim[r] | (c.h_red_dim[r] << 8)); }
        for_each_block([&](int ra, int rb) {
          if (ra == rb) return;                                  // a camera's Schur terms with itself: the diagonal block above
          const bool a_later = c.h_red_pos[ra] > c.h_red_pos[rb];
          const int rr = a_later ? ra : rb, rc = a_later ? rb : ra;
          xro.push_back(c.h_red_off[rr]); xco.push_back(c.h_red_off[rc]); xd.push_back(c.h_red_dim[rr] | (c.h_red_dim[rc] << 8));
        });
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
        // the block SET identifies the layout; a sharded handle lists it in bitmap order, a single one in term order
        uint64_t hb = 0;
        for (size_t i = 0; i < xd.size(); i++) {
          uint64_t z = (uint64_t)xro[i] * 0x9E3779B97F4A7C15ull ^ ((uint64_t)xco[i] + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)xd[i];
          z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
          hb += z;
        }
        mix(&hb, sizeof(hb));
      }
      c.structure_hash = h;
    }
    clk.lap("cholesky tile schedule");
    if (clk.on) std::fprintf(stderr, "[gtsam_amd setup] reduced system n = %lld, %d tiles, stored tile fraction %.3f, %.3f GFLOP per factorisation, critical path %d of %d column pairs\n",
                             (long long)c.n_red, nt, c.plan.dense_fraction, c.plan.flops * 1e-9, c.plan.critical_pairs, np2);
    // keep the nested-dissection ordering only where it pays: the chains must get clearly shorter and the problem must be
    // in the latency-bound regime (separators cost fill: on the L1723 shape +60 % flops for a 30 % shorter path)
    if (!part_of_pos.empty() && !nd_forced &&
        !(c.plan.critical_pairs * 10 <= np2 * 8 && c.plan.flops <= 6e10)) { retry_rcm = true; }
  }

  if (!retry_rcm) break;
  }

  // ---- upload -----------------------------------------------------------------------------------
  up(c.lm_var, c.h_lm_var, s); up(c.red_var, c.h_red_var, s); up(c.red_dim, c.h_red_dim, s);
  up(c.lm_index, c.h_lm_index, s); up(c.red_index, c.h_red_index, s); up(c.red_off, c.h_red_off, s);
  up(c.lm_owned, lm_owned, s);
  up(c.obs_red, obs_red, s); up(c.obs_lm, obs_lm, s);
  up(c.lm_obs_ptr, lm_obs_ptr, s); up(c.lm_obs, lm_obs, s); up(c.lm_pri_ptr, lm_pri_ptr, s); up(c.lm_pri, lm_pri, s);
  up(c.red_inc_ptr, inc_ptr, s); up(c.red_inc_kind, inc_kind, s); up(c.red_inc_idx, inc_idx, s);
  up(c.hoff_row, hoff_row, s); up(c.hoff_col, hoff_col, s); up(c.hoff_ptr, hoff_ptr, s); up(c.hoff_fac, hoff_fac, s);
  up(c.pair_row, pair_row, s); up(c.pair_col, pair_col, s); up(c.pair_ptr, pair_ptr, s);
  c.pair_oa.upload(pair_oa.get(), (size_t)c.n_pair_terms, s); c.pair_ob.upload(pair_ob.get(), (size_t)c.n_pair_terms, s);
  if (c.n_pair_terms == 0) { c.pair_oa.alloc(1); c.pair_ob.alloc(1); }

  // ---- numeric buffers --------------------------------------------------------------------------
  const size_t NP = c.NP;
  c.Hd.alloc(std::max<size_t>(81 * (size_t)c.n_red_vars, 81)); c.gred0.alloc(std::max<size_t>(9 * (size_t)c.n_red_vars, 9));
  c.hdiag_red.alloc(NP);
  c.V.alloc(std::max<size_t>(9 * (size_t)c.n_lm, 9)); c.gp.alloc(std::max<size_t>(3 * (size_t)c.n_lm, 3));
  c.Linv.alloc(std::max<size_t>(9 * (size_t)c.n_lm, 1)); c.ylm.alloc(std::max<size_t>(3 * (size_t)c.n_lm, 1));
  c.delta_lm.alloc(std::max<size_t>(3 * (size_t)c.n_lm, 1));
  c.E.alloc(std::max<size_t>(kEStride * (size_t)c.n_obs, 1));
  c.vobs.alloc(std::max<size_t>(3 * (size_t)c.n_obs, 1));
  c.Hoff.alloc(std::max<size_t>(81 * (size_t)c.n_hoff, 1));
  c.S.alloc((NP + kTile) * NP);
  c.Dinv.alloc((NP / kTile) * (size_t)kTile * kTile);
  check_hip(hipMemsetAsync(c.Dinv.p, 0, sizeof(double) * c.Dinv.n, c.stream), "memset");
  c.chol_epoch_dev.alloc(1);
  check_hip(hipMemsetAsync(c.chol_epoch_dev.p, 0, sizeof(long long), c.stream), "memset");
  c.xred.alloc(NP);
  c.partials.alloc(2 * 2048);
  c.scalars.alloc(SC_COUNT);
  check_hip(hipMemsetAsync(c.scalars.p, 0, sizeof(double) * SC_COUNT, s), "memset");
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

  c.chol_flops = c.plan.flops;
  // algorithmic HBM bytes of one linearize+assemble pass: factor indices + measurements +
  // variable blocks read, Jacobian records written and read once by the assembly, blocks written
  c.lin_bytes = (double)n_sfm * (2 * 4 + 16 + 4 + 2.0 * kSfmRec * 8) + (double)n_proj * (2 * 4 + 16 + 12 + 2.0 * kProjRec * 8) +
                (double)n_btw * (2 * 4 + 96 + 4 + 2.0 * kBetweenRec * 8) + (double)c.val_size * 8 +
                (double)c.n_red_vars * 90 * 8 + (double)c.n_lm * 12 * 8 + (double)c.n_hoff * 36 * 8;
}

static void exchange(gtg_context& c, double* ptr, int64_t n) {
  if (c.n_shards > 1) {
    if (!c.allreduce) throw std::runtime_error("n_shards > 1 but no allreduce callback was set (gtg_set_allreduce)");
    if (!c.layout_verified) verify_layout(c);   // the callback was registered after the upload
    const int rc = c.allreduce(ptr, n, (void*)c.stream, c.allreduce_user);
    if (rc != 0) throw std::runtime_error("allreduce callback failed");
  }
}

static void read_scalars(gtg_context& c) {
  exchange(c, c.scalars.p, SC_COUNT);
  check_hip(hipMemcpyAsync(c.h_scalars, c.scalars.p, sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost, c.stream), "D2H");
  check_hip(hipStreamSynchronize(c.stream), "sync");
  c.h_scalars[SC_DELTA_SQ] /= c.n_shards;  // identical on every shard, summed by the exchange
}

struct PhaseTimer {
  gtg_context& c; int ph; hipEvent_t a, b;
  PhaseTimer(gtg_context& c_, int ph_, hipEvent_t* evs) : c(c_), ph(ph_), a(evs[2 * ph_]), b(evs[2 * ph_ + 1]) {
    if (c.timing) (void)hipEventRecord(a, c.stream);
  }
  ~PhaseTimer() { if (c.timing) (void)hipEventRecord(b, c.stream); }
};
static void ensure_events(gtg_context& c) {   // the handle's own events, on its device
  if (!c.phase_events.empty()) return;
  c.phase_events.resize(2 * GTG_PH_COUNT, nullptr);
  for (auto& e : c.phase_events) check_hip(hipEventCreate(&e), "event");
}
static void collect(gtg_context& c, std::initializer_list<int> phases) {
  if (!c.timing) return;
  for (int ph : phases) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c.phase_events[2 * ph], c.phase_events[2 * ph + 1]) == hipSuccess) { c.phase_ms[ph] += ms; c.phase_calls[ph]++; }
  }
}

}  // namespace gt

namespace gt { long long* g_potrf_dbg_set(long long*); float debug_time_syrk(gtg_context&, double*, int, int, int, int); }
using namespace gt;

#define GTG_TRY try {
#define GTG_CATCH                                                                     \
  } catch (const std::invalid_argument& e) { g_last_error = e.what(); return GTG_ERR_USAGE; } \
  catch (const std::exception& e) { g_last_error = e.what(); return GTG_ERR_HIP; }

extern "C" {

const char* gtg_last_error(void) { return g_last_error.c_str(); }
const char* gtg_version(void) { return "gtsam_amd 0.1 (gfx950, FP64)"; }
const char* gtg_phase_name(int ph) {
  static const char* names[GTG_PH_COUNT] = {"linearize", "assemble", "point_eliminate", "schur", "cholesky",
                                            "solve", "linear_error", "retract", "error"};
  return (ph >= 0 && ph < GTG_PH_COUNT) ? names[ph] : "?";
}

int gtg_create(gtg_handle* out, int device_id) {
  GTG_TRY
  if (!out) throw std::invalid_argument("null out");
  int ndev = 0;
  check_hip(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
  if (device_id < 0 || device_id >= ndev) throw std::invalid_argument("bad device id (no HIP device visible?)");
  check_hip(hipSetDevice(device_id), "hipSetDevice");
  gtg_context* c = new gtg_context;
  c->device = device_id;
  check_hip(hipStreamCreate(&c->stream), "hipStreamCreate");
  ensure_events(*c);
  *out = c;
  return GTG_OK;
  GTG_CATCH
}

int gtg_destroy(gtg_handle c) {
  if (!c) return GTG_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  auto& f = c->f;
  DevBuf<double>* dbl[] = {&c->values, &c->trial, &c->delta, &c->noise_data, &f.sfm_z, &f.sfm_J, &f.proj_z, &f.proj_J,
                           &f.calib, &f.sensor, &f.between_z, &f.between_J, &f.prior_data, &f.prior_J, &c->Hd, &c->gred0,
                           &c->hdiag_red, &c->V, &c->gp, &c->Hoff, &c->Linv, &c->ylm, &c->E, &c->vobs, &c->pcg_vec, &c->pcg_bj, &c->pcg_y, &c->delta_lm, &c->S,
                           &c->Dinv, &c->xred, &c->partials, &c->scalars, &c->noise_rk};
  for (auto* b : dbl) b->free();
  DevBuf<int32_t>* i32[] = {&c->var_type, &c->lm_var, &c->red_var, &c->red_dim, &c->lm_index, &c->red_index, &c->lm_owned,
                            &c->noise_kind, &c->noise_rkind, &f.sfm_cam, &f.sfm_point, &f.sfm_noise, &f.proj_pose, &f.proj_point,
                            &f.proj_noise, &f.proj_calib, &f.proj_sensor, &f.between_v1, &f.between_v2, &f.between_noise,
                            &f.prior_var, &f.prior_noise, &c->obs_red, &c->obs_lm, &c->lm_obs, &c->lm_pri,
                            &c->red_inc_kind, &c->red_inc_idx, &c->hoff_row, &c->hoff_col, &c->hoff_fac, &c->pair_row,
                            &c->pair_col, &c->pair_oa, &c->pair_ob};
  for (auto* b : i32) b->free();
  c->plan.rows.free(); c->plan.pairs.free(); c->plan.bcols.free(); c->plan.stored.free(); c->plan.exch.free(); c->xbuf.free();
  DevBuf<int64_t>* i64[] = {&c->val_off, &c->dim_off, &c->red_off, &c->noise_off, &f.prior_off, &c->lm_obs_ptr,
                            &c->lm_pri_ptr, &c->red_inc_ptr, &c->hoff_ptr, &c->pair_ptr, &c->pad_index};
  for (auto* b : i64) b->free();
  c->chol_epoch_dev.free(); c->layout_probe.free(); c->xb_row_off.free(); c->xb_col_off.free(); c->xb_dim.free();
  destroy_chol_streams(*c);
  for (hipEvent_t e : c->phase_events) if (e) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(c->stream);
  drop_index(c);
  delete c;
  return GTG_OK;
}

int gtg_upload_problem(gtg_handle c, const gtg_problem* p, int shard, int n_shards) {
  GTG_TRY
  if (!c || !p) throw std::invalid_argument("null argument");
  if (n_shards < 1 || shard < 0 || shard >= n_shards) throw std::invalid_argument("bad shard / n_shards");
  check_hip(hipSetDevice(c->device), "hipSetDevice");
  StageClock clk;
  hipStream_t s = c->stream;
  c->shard = shard; c->n_shards = n_shards;
  c->n_vars = p->n_vars;
  c->h_var_type.assign(p->var_type, p->var_type + p->n_vars);
  c->h_val_off.assign(p->n_vars + 1, 0); c->h_dim_off.assign(p->n_vars + 1, 0);
  for (int v = 0; v < p->n_vars; v++) {
    const int t = p->var_type[v];
    if (t < 0 || t > GTG_VAR_POSE2) throw std::invalid_argument("unknown variable type");
    c->h_val_off[v + 1] = c->h_val_off[v] + storage_size(t);
    c->h_dim_off[v + 1] = c->h_dim_off[v] + tangent_dim(t);
  }
  c->val_size = c->h_val_off[p->n_vars]; c->dim_size = c->h_dim_off[p->n_vars];
  up(c->var_type, c->h_var_type, s); up(c->val_off, c->h_val_off, s); up(c->dim_off, c->h_dim_off, s);
  c->values.alloc(std::max<int64_t>(c->val_size, 1)); c->trial.alloc(std::max<int64_t>(c->val_size, 1));
  c->delta.alloc(std::max<int64_t>(c->dim_size, 1));
  check_hip(hipMemsetAsync(c->delta.p, 0, sizeof(double) * c->delta.n, s), "memset");

  // noise table: derive the inverse sigmas like the reference constructors (NoiseModel.cpp:275-281, Isotropic ctor)
  {
    std::vector<int32_t> kind(p->noise_kind, p->noise_kind + p->n_noise);
    std::vector<int64_t> noff(p->n_noise);
    std::vector<double> data;
    for (int i = 0; i < p->n_noise; i++) {
      const int dim = p->noise_dim[i];
      const double* d = p->noise_data + p->noise_off[i];
      noff[i] = (int64_t)data.size();
      switch (kind[i]) {
        case GTG_NOISE_UNIT: data.push_back(0.0); break;
        case GTG_NOISE_ISOTROPIC: data.push_back(1.0 / d[0]); break;
        case GTG_NOISE_DIAGONAL: for (int k = 0; k < dim; k++) data.push_back(1.0 / d[k]); break;
        case GTG_NOISE_GAUSSIAN: for (int k = 0; k < dim * dim; k++) data.push_back(d[k]); break;
        default: throw std::invalid_argument("unsupported noise model kind (Robust/Constrained are out of scope)");
      }
    }
    std::vector<int32_t> rkind(p->n_noise, GTG_ROBUST_NONE);
    std::vector<double> rk(p->n_noise, 0.0);
    for (int i = 0; i < p->n_noise; i++) {
      if (p->noise_robust) rkind[i] = p->noise_robust[i];
      if (rkind[i] < GTG_ROBUST_NONE || rkind[i] > GTG_ROBUST_GEMANMCCLURE) throw std::invalid_argument("unsupported m-estimator");
      if (rkind[i] != GTG_ROBUST_NONE) {
        rk[i] = p->noise_robust_param ? p->noise_robust_param[i] : 0.0;
        if (!(rk[i] > 0.0)) throw std::invalid_argument("m-estimator parameter must be > 0");   // LossFunctions.cpp ctor checks
      }
    }
    up(c->noise_kind, kind, s); up(c->noise_off, noff, s); up(c->noise_data, data, s);
    up(c->noise_rkind, rkind, s); up(c->noise_rk, rk, s);
  }
  auto check_noise = [&](int idx, int dim, const char* what) {
    if (idx < 0 || idx >= p->n_noise || p->noise_dim[idx] != dim)
      throw std::invalid_argument(std::string(what) + ": NoiseModel has wrong dimension");  // NonlinearFactor.cpp:97-104
  };
  auto check_var = [&](int v) { if (v < 0 || v >= p->n_vars) throw std::invalid_argument("factor refers to a key that is not in Values"); };

  HostIndex& hi = host_index(c);
  auto& f = c->f;
  // shard filter: landmark factors follow their landmark (rank among POINT3 variables), others round-robin
  std::vector<int32_t> lm_rank(p->n_vars, -1);
  { int k = 0; for (int v = 0; v < p->n_vars; v++) if (p->var_type[v] == GTG_VAR_POINT3) lm_rank[v] = k++; }
  auto own_lm = [&](int v) { return lm_rank[v] >= 0 && (lm_rank[v] % n_shards) == shard; };

  hi.all_obs_red_var.clear(); hi.all_obs_point.clear(); hi.all_between_v1.clear(); hi.all_between_v2.clear();
  if (n_shards > 1) {
    for (int64_t i = 0; i < p->n_sfm; i++) { check_var(p->sfm_cam[i]); check_var(p->sfm_point[i]); }
    for (int64_t i = 0; i < p->n_proj; i++) { check_var(p->proj_pose[i]); check_var(p->proj_point[i]); }
    for (int64_t i = 0; i < p->n_between; i++) { check_var(p->between_v1[i]); check_var(p->between_v2[i]); }
    hi.all_obs_red_var.assign(p->sfm_cam, p->sfm_cam + p->n_sfm); hi.all_obs_red_var.insert(hi.all_obs_red_var.end(), p->proj_pose, p->proj_pose + p->n_proj);
    hi.all_obs_point.assign(p->sfm_point, p->sfm_point + p->n_sfm); hi.all_obs_point.insert(hi.all_obs_point.end(), p->proj_point, p->proj_point + p->n_proj);
    hi.all_between_v1.assign(p->between_v1, p->between_v1 + p->n_between); hi.all_between_v2.assign(p->between_v2, p->between_v2 + p->n_between);
  }
  { // SFM
    std::vector<int32_t> cam, pt, nz; std::vector<double> z;
    for (int64_t i = 0; i < p->n_sfm; i++) { check_var(p->sfm_cam[i]); check_var(p->sfm_point[i]); check_noise(p->sfm_noise[i], 2, "GeneralSFMFactor"); }
    if (n_shards == 1) {   // the whole table: block copies
      cam.assign(p->sfm_cam, p->sfm_cam + p->n_sfm); pt.assign(p->sfm_point, p->sfm_point + p->n_sfm);
      nz.assign(p->sfm_noise, p->sfm_noise + p->n_sfm); z.assign(p->sfm_z, p->sfm_z + 2 * p->n_sfm);
    } else {
      for (int64_t i = 0; i < p->n_sfm; i++) {
        if (!own_lm(p->sfm_point[i])) continue;
        cam.push_back(p->sfm_cam[i]); pt.push_back(p->sfm_point[i]); nz.push_back(p->sfm_noise[i]);
        z.push_back(p->sfm_z[2 * i]); z.push_back(p->sfm_z[2 * i + 1]);
      }
    }
    f.n_sfm = (int64_t)cam.size();
    up(f.sfm_cam, cam, s); up(f.sfm_point, pt, s); up(f.sfm_noise, nz, s); up(f.sfm_z, z, s);
    f.sfm_J.alloc(std::max<size_t>((size_t)kSfmRec * f.n_sfm, 1));
    hi.sfm_cam = std::move(cam); hi.sfm_point = std::move(pt);
  }
  { // projection
    std::vector<int32_t> pose, pt, nz, cal, sen; std::vector<double> z;
    for (int64_t i = 0; i < p->n_proj; i++) {
      check_var(p->proj_pose[i]); check_var(p->proj_point[i]); check_noise(p->proj_noise[i], 2, "GenericProjectionFactor");
      if (p->proj_calib[i] < 0 || p->proj_calib[i] >= p->n_calib) throw std::invalid_argument("bad calibration index");
      const int si = p->proj_sensor ? p->proj_sensor[i] : -1;
      if (si >= p->n_sensor) throw std::invalid_argument("bad body_P_sensor index");
      if (n_shards > 1 && !own_lm(p->proj_point[i])) continue;
      pose.push_back(p->proj_pose[i]); pt.push_back(p->proj_point[i]); nz.push_back(p->proj_noise[i]);
      cal.push_back(p->proj_calib[i]); sen.push_back(si);
      z.push_back(p->proj_z[2 * i]); z.push_back(p->proj_z[2 * i + 1]);
    }
    f.n_proj = (int64_t)pose.size();
    up(f.proj_pose, pose, s); up(f.proj_point, pt, s); up(f.proj_noise, nz, s); up(f.proj_calib, cal, s);
    up(f.proj_sensor, sen, s); up(f.proj_z, z, s);
    std::vector<double> calib(p->calib, p->calib + 5 * (size_t)p->n_calib), sensor(p->sensor, p->sensor + 12 * (size_t)p->n_sensor);
    up(f.calib, calib, s); up(f.sensor, sensor, s);
    f.proj_J.alloc(std::max<size_t>((size_t)kProjRec * f.n_proj, 1));
    hi.proj_pose = pose; hi.proj_point = pt;
  }
  { // between
    std::vector<int32_t> v1, v2, nz; std::vector<double> z;
    for (int64_t i = 0; i < p->n_between; i++) {
      check_var(p->between_v1[i]); check_var(p->between_v2[i]);
      check_noise(p->between_noise[i], tangent_dim(p->var_type[p->between_v1[i]]), "BetweenFactor");
      if (n_shards > 1 && (i % n_shards) != shard) continue;
      v1.push_back(p->between_v1[i]); v2.push_back(p->between_v2[i]); nz.push_back(p->between_noise[i]);
      for (int k = 0; k < 12; k++) z.push_back(p->between_z[12 * i + k]);
    }
    f.n_between = (int64_t)v1.size();
    up(f.between_v1, v1, s); up(f.between_v2, v2, s); up(f.between_noise, nz, s); up(f.between_z, z, s);
    f.between_J.alloc(std::max<size_t>((size_t)kBetweenRec * f.n_between, 1));
    hi.between_v1 = v1; hi.between_v2 = v2;
  }
  { // priors
    std::vector<int32_t> var, nz; std::vector<int64_t> poff; std::vector<double> data;
    for (int64_t i = 0; i < p->n_prior; i++) {
      const int v = p->prior_var[i];
      check_var(v); check_noise(p->prior_noise[i], tangent_dim(p->var_type[v]), "PriorFactor");
      const bool mine = lm_rank[v] >= 0 ? own_lm(v) : ((i % n_shards) == shard);
      if (n_shards > 1 && !mine) continue;
      var.push_back(v); nz.push_back(p->prior_noise[i]); poff.push_back((int64_t)data.size());
      const double* d = p->prior_data + p->prior_off[i];
      for (int k = 0; k < storage_size(p->var_type[v]); k++) data.push_back(d[k]);
    }
    f.n_prior = (int64_t)var.size();
    up(f.prior_var, var, s); up(f.prior_noise, nz, s); up(f.prior_off, poff, s); up(f.prior_data, data, s);
    f.prior_J.alloc(std::max<size_t>((size_t)kPriorRec * f.n_prior, 1));
    hi.prior_var = var;
  }
  clk.lap("factor tables (shard filter + upload)");
  analyze(*c);
  c->uploaded = true; c->linearized = false; c->have_trial = false;
  return GTG_OK;
  GTG_CATCH
}

int gtg_set_reduced_ordering(gtg_handle c, const int32_t* order, int32_t n) {
  GTG_TRY
  if (!c) throw std::invalid_argument("null handle");
  HostIndex& hi = host_index(c);
  hi.user_order.assign(order, order + n);
  if (c->uploaded) { check_hip(hipSetDevice(c->device), "hipSetDevice"); analyze(*c); c->linearized = false; c->have_trial = false; }
  return GTG_OK;
  GTG_CATCH
}

int64_t gtg_values_size(gtg_handle c) { return c ? c->val_size : -1; }
int64_t gtg_tangent_size(gtg_handle c) { return c ? c->dim_size : -1; }
int64_t gtg_reduced_dim(gtg_handle c) { return c ? c->n_red : -1; }

int gtg_set_values(gtg_handle c, const double* packed, int64_t n) {
  GTG_TRY
  if (!c || !c->uploaded || n != c->val_size) throw std::invalid_argument("gtg_set_values: wrong size or no problem uploaded");
  check_hip(hipSetDevice(c->device), "hipSetDevice");
  check_hip(hipMemcpyAsync(c->values.p, packed, sizeof(double) * n, hipMemcpyHostToDevice, c->stream), "H2D");
  check_hip(hipStreamSynchronize(c->stream), "sync");
  c->linearized = false; c->have_trial = false;
  return GTG_OK;
  GTG_CATCH
}

static int get_buf(gtg_handle c, const double* dev, int64_t have, double* out, int64_t n) {
  GTG_TRY
  if (!c || !c->uploaded || n != have) throw std::invalid_argument("getter: wrong size or no problem uploaded");
  check_hip(hipSetDevice(c->device), "hipSetDevice");
  check_hip(hipMemcpyAsync(out, dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream), "D2H");
  check_hip(hipStreamSynchronize(c->stream), "sync");
  return GTG_OK;
  GTG_CATCH
}
int gtg_get_values(gtg_handle c, double* packed, int64_t n) { return get_buf(c, c ? c->values.p : nullptr, c ? c->val_size : -1, packed, n); }
int gtg_get_trial_values(gtg_handle c, double* packed, int64_t n) { return get_buf(c, c ? c->trial.p : nullptr, c ? c->val_size : -1, packed, n); }
int gtg_get_delta(gtg_handle c, double* d, int64_t n) { return get_buf(c, c ? c->delta.p : nullptr, c ? c->dim_size : -1, d, n); }

int gtg_error(gtg_handle c, double* error) {
  GTG_TRY
  if (!c || !c->uploaded || !error) throw std::invalid_argument("gtg_error: no problem uploaded");
  check_hip(hipSetDevice(c->device), "hipSetDevice");
  { PhaseTimer t(*c, GTG_PH_ERROR, c->phase_events.data()); launch_error(*c, c->values.p, SC_ERROR); }
  read_scalars(*c);
  collect(*c, {GTG_PH_ERROR});
  *error = c->h_scalars[SC_ERROR];
  return GTG_OK;
  GTG_CATCH
}

int gtg_linearize(gtg_handle c) {
  GTG_TRY
  if (!c || !c->uploaded) throw std::invalid_argument("gtg_linearize: no problem uploaded");
  check_hip(hipSetDevice(c->device), "hipSetDevice");
  { PhaseTimer t(*c, GTG_PH_LINEARIZE, c->phase_events.data()); launch_linearize(*c); }
  { PhaseTimer t(*c, GTG_PH_ASSEMBLE, c->phase_events.data()); launch_assemble(*c); }
  exchange(*c, c->hdiag_red.p, c->NP);   // damping needs the full diagonal on every shard
  check_hip(hipStreamSynchronize(c->stream), "sync");
  collect(*c, {GTG_PH_LINEARIZE, GTG_PH_ASSEMBLE});
  c->linearized = true;
  return GTG_OK;
  GTG_CATCH
}

int gtg_try_lambda(gtg_handle c, double lambda, int diag, double dmin, double dmax, double out[4]) {
  GTG_TRY
  if (!c || !c->uploaded || !c->linearized) throw std::invalid_argument("gtg_try_lambda: call gtg_linearize first");
  if (!(lambda > 0.0)) throw std::invalid_argument("gtg_try_lambda: lambda must be > 0");
  check_hip(hipSetDevice(c->device), "hipSetDevice");
  check_hip(hipMemsetAsync(c->scalars.p + SC_FAIL, 0, sizeof(double), c->stream), "memset");
  { PhaseTimer t(*c, GTG_PH_POINT_ELIM, c->phase_events.data()); launch_point_eliminate(*c, lambda, diag, dmin, dmax); }
  { PhaseTimer t(*c, GTG_PH_SCHUR, c->phase_events.data()); launch_build_reduced(*c, lambda, diag, dmin, dmax); }
  if (c->n_shards > 1) {   // the one big exchange: reduced Hessian + rhs
    static const bool by_tiles = std::getenv("GTG_EXCHANGE_TILES") != nullptr;   // whole 128x128 tiles (the first version) instead of blocks
    if (by_tiles || c->n_xb == 0) {
      const int64_t nb = c->plan.n_exch * kTile * kTile;
      if ((int64_t)c->xbuf.n != nb) c->xbuf.alloc(nb);
      launch_pack_tiles(*c, c->S.p, c->NP, c->plan, c->xbuf.p, false);
      exchange(*c, c->xbuf.p, nb);
      launch_pack_tiles(*c, c->S.p, c->NP, c->plan, c->xbuf.p, true);
    } else {                // only the structurally non-zero d x d blocks (the same list on every shard) + rhs row + padding
      const int64_t nb = exchange_block_doubles(*c);
      if ((int64_t)c->xbuf.n != nb) c->xbuf.alloc(nb);
      launch_pack_blocks(*c, c->S.p, c->NP, c->xbuf.p, false);
      exchange(*c, c->xbuf.p, nb);
      launch_pack_blocks(*c, c->S.p, c->NP, c->xbuf.p, true);
    }
  }
  { PhaseTimer t(*c, GTG_PH_CHOLESKY, c->phase_events.data()); launch_cholesky(*c, c->S.p, c->NP, c->plan, c->Dinv.p, c->scalars.p + SC_FAIL); }
  { PhaseTimer t(*c, GTG_PH_SOLVE, c->phase_events.data());
    launch_backward_solve(*c, c->S.p, c->NP, c->plan, c->Dinv.p, c->xred.p);
    launch_back_substitute(*c);
    if (c->n_lm) exchange(*c, c->delta_lm.p, 3 * (int64_t)c->n_lm);
    launch_scatter_delta(*c); }
  { PhaseTimer t(*c, GTG_PH_LINEAR_ERROR, c->phase_events.data()); launch_linear_error(*c); }
  { PhaseTimer t(*c, GTG_PH_RETRACT, c->phase_events.data()); launch_retract(*c); }
  { PhaseTimer t(*c, GTG_PH_ERROR, c->phase_events.data()); launch_error(*c, c->trial.p, SC_TRIAL_ERROR); }
  read_scalars(*c);
  collect(*c, {GTG_PH_POINT_ELIM, GTG_PH_SCHUR, GTG_PH_CHOLESKY, GTG_PH_SOLVE, GTG_PH_LINEAR_ERROR, GTG_PH_RETRACT, GTG_PH_ERROR});
  c->have_trial = true;
  const double dsq = c->h_scalars[SC_DELTA_SQ];
  if (c->h_scalars[SC_FAIL] != 0.0 || !std::isfinite(dsq)) return GTG_INDETERMINATE;
  out[0] = c->h_scalars[SC_LIN0];
  out[1] = c->h_scalars[SC_LIN1];
  out[2] = (out[0] - out[1] >= 0) ? c->h_scalars[SC_TRIAL_ERROR] : std::numeric_limits<double>::infinity();
  out[3] = std::sqrt(dsq);
  return GTG_OK;
  GTG_CATCH
}

// Same contract as gtg_try_lambda, the damped system solved by block-Jacobi PCG on the implicit Schur complement
// (NonlinearOptimizerParams::Iterative + PCGSolverParameters in the reference, NonlinearOptimizer.cpp:154-172).
int gtg_try_lambda_pcg(gtg_handle c, double lambda, int diag, double dmin, double dmax, const double cg[4], double out[4],
                       int32_t* iterations) {
  GTG_TRY
  if (!c || !c->uploaded || !c->linearized) throw std::invalid_argument("gtg_try_lambda_pcg: call gtg_linearize first");
  if (!(lambda > 0.0) || !cg) throw std::invalid_argument("gtg_try_lambda_pcg: lambda must be > 0, cg = {max, min, eps_rel, eps_abs}");
  check_hip(hipSetDevice(c->device), "hipSetDevice");
  check_hip(hipMemsetAsync(c->scalars.p + SC_FAIL, 0, sizeof(double), c->stream), "memset");
  { PhaseTimer t(*c, GTG_PH_POINT_ELIM, c->phase_events.data()); launch_point_eliminate(*c, lambda, diag, dmin, dmax); }
  double g0 = 0.0, g1 = 0.0;
  int its = 0;
  { PhaseTimer t(*c, GTG_PH_CHOLESKY, c->phase_events.data());
    its = launch_pcg(*c, lambda, diag, dmin, dmax, (int)cg[0], (int)cg[1], cg[2], cg[3], &g0, &g1); }
  if (iterations) *iterations = its;
  { PhaseTimer t(*c, GTG_PH_SOLVE, c->phase_events.data());
    launch_back_substitute(*c);
    launch_scatter_delta(*c); }
  { PhaseTimer t(*c, GTG_PH_LINEAR_ERROR, c->phase_events.data()); launch_linear_error(*c); }
  { PhaseTimer t(*c, GTG_PH_RETRACT, c->phase_events.data()); launch_retract(*c); }
  { PhaseTimer t(*c, GTG_PH_ERROR, c->phase_events.data()); launch_error(*c, c->trial.p, SC_TRIAL_ERROR); }
  read_scalars(*c);
  collect(*c, {GTG_PH_POINT_ELIM, GTG_PH_CHOLESKY, GTG_PH_SOLVE, GTG_PH_LINEAR_ERROR, GTG_PH_RETRACT, GTG_PH_ERROR});
  c->have_trial = true;
  const double dsq = c->h_scalars[SC_DELTA_SQ];
  if (c->h_scalars[SC_FAIL] != 0.0 || !std::isfinite(dsq) || !std::isfinite(g1)) return GTG_INDETERMINATE;
  out[0] = c->h_scalars[SC_LIN0];
  out[1] = c->h_scalars[SC_LIN1];
  out[2] = (out[0] - out[1] >= 0) ? c->h_scalars[SC_TRIAL_ERROR] : std::numeric_limits<double>::infinity();
  out[3] = std::sqrt(dsq);
  return GTG_OK;
  GTG_CATCH
}

int gtg_accept(gtg_handle c) {
  GTG_TRY
  if (!c || !c->have_trial) throw std::invalid_argument("gtg_accept: no trial values (call gtg_try_lambda)");
  std::swap(c->values.p, c->trial.p);
  c->linearized = false; c->have_trial = false;
  return GTG_OK;
  GTG_CATCH
}

int gtg_get_gradient(gtg_handle c, double* g, int64_t n) {
  GTG_TRY
  if (!c || !c->linearized || n != c->dim_size) throw std::invalid_argument("gtg_get_gradient: linearize first / wrong size");
  std::vector<double> gr(9 * (size_t)std::max(c->n_red_vars, 1)), gp(3 * (size_t)std::max(c->n_lm, 1));
  check_hip(hipMemcpy(gr.data(), c->gred0.p, sizeof(double) * gr.size(), hipMemcpyDeviceToHost), "D2H");
  check_hip(hipMemcpy(gp.data(), c->gp.p, sizeof(double) * gp.size(), hipMemcpyDeviceToHost), "D2H");
  for (int v = 0; v < c->n_vars; v++) {
    double* d = g + c->h_dim_off[v];
    if (c->h_lm_index[v] >= 0) for (int k = 0; k < 3; k++) d[k] = gp[3 * c->h_lm_index[v] + k];
    else for (int k = 0; k < c->h_red_dim[c->h_red_index[v]]; k++) d[k] = gr[9 * c->h_red_index[v] + k];
  }
  return GTG_OK;
  GTG_CATCH
}

int gtg_get_hessian_diagonal(gtg_handle c, double* out, int64_t n) {
  GTG_TRY
  if (!c || !c->linearized || n != c->dim_size) throw std::invalid_argument("gtg_get_hessian_diagonal: linearize first / wrong size");
  std::vector<double> hd(c->NP), V(9 * (size_t)std::max(c->n_lm, 1));
  check_hip(hipMemcpy(hd.data(), c->hdiag_red.p, sizeof(double) * hd.size(), hipMemcpyDeviceToHost), "D2H");
  check_hip(hipMemcpy(V.data(), c->V.p, sizeof(double) * V.size(), hipMemcpyDeviceToHost), "D2H");
  for (int v = 0; v < c->n_vars; v++) {
    double* d = out + c->h_dim_off[v];
    if (c->h_lm_index[v] >= 0) for (int k = 0; k < 3; k++) d[k] = V[9 * c->h_lm_index[v] + 4 * k];
    else { const int r = c->h_red_index[v]; for (int k = 0; k < c->h_red_dim[r]; k++) d[k] = hd[c->h_red_off[r] + k]; }
  }
  return GTG_OK;
  GTG_CATCH
}

int gtg_get_jacobians(gtg_handle c, int type, double* out, int64_t n) {
  GTG_TRY
  if (!c || !c->linearized) throw std::invalid_argument("gtg_get_jacobians: linearize first");
  const double* src; int64_t cnt;
  switch (type) {
    case GTG_FAC_GENERAL_SFM: src = c->f.sfm_J.p; cnt = c->f.n_sfm * kSfmRec; break;
    case GTG_FAC_PROJECTION: src = c->f.proj_J.p; cnt = c->f.n_proj * kProjRec; break;
    case GTG_FAC_BETWEEN_POSE3: src = c->f.between_J.p; cnt = c->f.n_between * kBetweenRec; break;
    case GTG_FAC_PRIOR: src = c->f.prior_J.p; cnt = c->f.n_prior * kPriorRec; break;
    default: throw std::invalid_argument("unknown factor type");
  }
  if (n != cnt) throw std::invalid_argument("gtg_get_jacobians: wrong output size");
  if (cnt) check_hip(hipMemcpy(out, src, sizeof(double) * cnt, hipMemcpyDeviceToHost), "D2H");
  return GTG_OK;
  GTG_CATCH
}

int gtg_get_reduced_matrix(gtg_handle c, double* S, int64_t n_elems) {
  GTG_TRY
  if (!c || !c->uploaded || n_elems != c->n_red * c->n_red) throw std::invalid_argument("gtg_get_reduced_matrix: wrong size");
  // S carries alignment gaps (identity rows) between the nested-dissection parts: copy the square part and compact it
  const int64_t n = c->n_red, NP = c->NP;
  std::vector<double> full((size_t)NP * NP);
  check_hip(hipMemcpy(full.data(), c->S.p, sizeof(double) * full.size(), hipMemcpyDeviceToHost), "D2H");
  std::vector<char> is_pad((size_t)NP, 0);
  for (int64_t i : c->h_pad_index) is_pad[(size_t)i] = 1;
  std::vector<int64_t> keep; keep.reserve((size_t)n);
  for (int64_t i = 0; i < NP; i++) if (!is_pad[(size_t)i]) keep.push_back(i);
  if ((int64_t)keep.size() != n) throw std::runtime_error("gtg_get_reduced_matrix: padding bookkeeping is inconsistent");
  for (int64_t i = 0; i < n; i++)
    for (int64_t j = 0; j < n; j++) S[i * n + j] = full[(size_t)keep[(size_t)i] * NP + keep[(size_t)j]];
  return GTG_OK;
  GTG_CATCH
}

int gtg_set_allreduce(gtg_handle c, gtg_allreduce_fn fn, void* user) {
  if (!c) return GTG_ERR_USAGE;
  c->allreduce = fn; c->allreduce_user = user;
  return GTG_OK;
}

int gtg_enable_timing(gtg_handle c, int on) { if (!c) return GTG_ERR_USAGE; c->timing = on != 0; return GTG_OK; }
int gtg_reset_timing(gtg_handle c) {
  if (!c) return GTG_ERR_USAGE;
  for (int i = 0; i < GTG_PH_COUNT; i++) { c->phase_ms[i] = 0; c->phase_calls[i] = 0; }
  return GTG_OK;
}
int gtg_get_phase_ms(gtg_handle c, double* ms, int64_t* calls, int n) {
  if (!c || n < GTG_PH_COUNT) return GTG_ERR_USAGE;
  for (int i = 0; i < GTG_PH_COUNT; i++) { ms[i] = c->phase_ms[i]; if (calls) calls[i] = c->phase_calls[i]; }
  return GTG_OK;
}
double gtg_cholesky_flops(gtg_handle c) { return c ? c->chol_flops : 0.0; }
int64_t gtg_structure_hash(gtg_handle c) { return c ? (int64_t)(c->structure_hash & 0x7FFFFFFFFFFFFFFFull) : -1; }
double gtg_linearize_bytes(gtg_handle c) { return c ? c->lin_bytes : 0.0; }

// debug only (not in the public header): ms per K=256 trailing update over an m x m tile grid, with ablations
double gtg_debug_syrk_ms(gtg_handle c, int m, int abl, int reps) {
  try {
    check_hip(hipSetDevice(c->device), "hipSetDevice");
    const int NP = (m + 2) * kTile;
    DevBuf<double> S; S.alloc((size_t)NP * NP);
    check_hip(hipMemset(S.p, 0, sizeof(double) * S.n), "memset");
    const double ms = debug_time_syrk(*c, S.p, NP, m, abl, reps);
    S.free();
    return ms;
  } catch (const std::exception& e) { g_last_error = e.what(); return -1.0; }
}

// debug only (not in the public header): cycle stamps of k_potrf128 stages on a 128x128 SPD matrix
int gtg_debug_potrf_stamps(gtg_handle c, double* A128, long long* out15) {
  GTG_TRY
  DevBuf<long long> dbg; dbg.alloc(16);
  check_hip(hipMemset(dbg.p, 0, 16 * sizeof(long long)), "memset");
  g_potrf_dbg_set(dbg.p);
  const int rc = gtg_dense_cholesky_host(c, A128, 128, nullptr);
  g_potrf_dbg_set(nullptr);
  check_hip(hipMemcpy(out15, dbg.p, 15 * sizeof(long long), hipMemcpyDeviceToHost), "D2H");
  dbg.free();
  return rc;
  GTG_CATCH
}

// tests: the tile schedule of the reduced-system Cholesky as the kernels read it (index lists only)
int gtg_debug_plan_sizes(gtg_handle c, int64_t sizes[8]) {
  GTG_TRY
  if (!c || !c->uploaded || !sizes) throw std::invalid_argument("gtg_debug_plan_sizes: no problem uploaded");
  const CholPlan& pl = c->plan;
  sizes[0] = pl.nt; sizes[1] = (int64_t)pl.rows.n; sizes[2] = (int64_t)pl.pairs.n; sizes[3] = (int64_t)pl.bcols.n;
  sizes[4] = pl.n_stored; sizes[5] = pl.n_exch; sizes[6] = (int64_t)pl.s1_off.size(); sizes[7] = (int64_t)pl.part_parent.size();
  return GTG_OK;
  GTG_CATCH
}
int gtg_debug_plan_lists(gtg_handle c, int32_t* rows, int32_t* pairs, int32_t* bcols, int32_t* stored, int32_t* exch,
                         int64_t* per_tile, int64_t* per_pair, int32_t* pair_part, int32_t* part_parent) {
  GTG_TRY
  if (!c || !c->uploaded) throw std::invalid_argument("gtg_debug_plan_lists: no problem uploaded");
  check_hip(hipSetDevice(c->device), "hipSetDevice");
  const CholPlan& pl = c->plan;
  auto down = [&](int32_t* dst, const DevBuf<int32_t>& b, size_t n) { if (dst && n) check_hip(hipMemcpy(dst, b.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost), "D2H"); };
  down(rows, pl.rows, pl.rows.n); down(pairs, pl.pairs, pl.pairs.n); down(bcols, pl.bcols, pl.bcols.n);
  down(stored, pl.stored, 2 * (size_t)pl.n_stored); down(exch, pl.exch, 2 * (size_t)pl.n_exch);
  if (per_tile) for (int k = 0; k < pl.nt; k++) { per_tile[4 * k] = pl.trsm_off[k]; per_tile[4 * k + 1] = pl.trsm_cnt[k]; per_tile[4 * k + 2] = pl.bwd_off[k]; per_tile[4 * k + 3] = pl.bwd_cnt[k]; }
  if (per_pair) for (size_t p = 0; p < pl.s1_off.size(); p++) {
    int64_t* q = per_pair + 8 * p;
    q[0] = pl.s1_off[p]; q[1] = pl.s1_cnt[p]; q[2] = pl.nar_off[p]; q[3] = pl.nar_cnt[p]; q[4] = pl.rest_off[p]; q[5] = pl.rest_cnt[p];
    q[6] = pl.anc_off[p]; q[7] = pl.anc_cnt[p];
  }
  if (pair_part) for (size_t p = 0; p < pl.pair_part.size(); p++) pair_part[p] = pl.pair_part[p];
  if (part_parent) for (size_t x = 0; x < pl.part_parent.size(); x++) part_parent[x] = pl.part_parent[x];
  return GTG_OK;
  GTG_CATCH
}

int gtg_dense_cholesky_host(gtg_handle c, double* A, int32_t n, double* rhs) {
  GTG_TRY
  if (!c || !A || n < 1) throw std::invalid_argument("gtg_dense_cholesky_host: bad arguments");
  check_hip(hipSetDevice(c->device), "hipSetDevice");
  const int NP = (n + kTile - 1) / kTile * kTile;
  DevBuf<double> S, Dinv, x, fail;
  S.alloc((size_t)(NP + kTile) * NP); Dinv.alloc((size_t)(NP / kTile) * kTile * kTile); x.alloc(NP); fail.alloc(1);
  check_hip(hipMemset(Dinv.p, 0, sizeof(double) * Dinv.n), "memset");
  if (!c->chol_epoch_dev.p) { c->chol_epoch_dev.alloc(1); check_hip(hipMemset(c->chol_epoch_dev.p, 0, sizeof(long long)), "memset"); }
  check_hip(hipMemsetAsync(S.p, 0, sizeof(double) * S.n, c->stream), "memset");
  check_hip(hipMemsetAsync(fail.p, 0, sizeof(double), c->stream), "memset");
  check_hip(hipMemcpy2DAsync(S.p, sizeof(double) * NP, A, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyHostToDevice, c->stream), "H2D 2D");
  std::vector<double> ones(NP - n, 1.0);
  if (NP > n) check_hip(hipMemcpy2DAsync(S.p + (size_t)n * NP + n, sizeof(double) * (NP + 1), ones.data(), sizeof(double), sizeof(double), NP - n, hipMemcpyHostToDevice, c->stream), "pad");
  if (rhs) check_hip(hipMemcpyAsync(S.p + (size_t)NP * NP, rhs, sizeof(double) * n, hipMemcpyHostToDevice, c->stream), "rhs");
  CholPlan plan;
  build_chol_plan(plan, NP / kTile, nullptr, c->stream);   // dense
  launch_cholesky(*c, S.p, NP, plan, Dinv.p, fail.p);
  if (rhs) launch_backward_solve(*c, S.p, NP, plan, Dinv.p, x.p);
  double hf = 0;
  check_hip(hipMemcpyAsync(&hf, fail.p, sizeof(double), hipMemcpyDeviceToHost, c->stream), "D2H");
  check_hip(hipMemcpy2DAsync(A, sizeof(double) * n, S.p, sizeof(double) * NP, sizeof(double) * n, n, hipMemcpyDeviceToHost, c->stream), "D2H 2D");
  if (rhs) check_hip(hipMemcpyAsync(rhs, x.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream), "D2H");
  check_hip(hipStreamSynchronize(c->stream), "sync");
  S.free(); Dinv.free(); x.free(); fail.free(); plan.rows.free(); plan.pairs.free(); plan.bcols.free(); plan.stored.free();
  return hf != 0.0 ? GTG_INDETERMINATE : GTG_OK;
  GTG_CATCH
}

}  // extern "C"
