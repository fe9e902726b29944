// api.hip -- the C ABI (include/gtsam_amd.h): handle life cycle, upload of the factor tables (shard filter, noise table),
// the per-iteration entry points (linearize / try_lambda / accept) that issue the HIP kernels, getters and test hooks.
// The one-time symbolic analysis of a graph is analysis.hip.  All arithmetic of the hot path runs in the HIP kernels.
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
#include <map>
#include <set>
#include <memory>
#include <new>
#include <sys/mman.h>
#include <mutex>
#include <stdexcept>

#include "analysis.h"
#include "factors.h"
#include "kernels.h"

// GTG_FUSED_SFM=0 at compile time builds the stored-record form of rounds 1-4 for every graph (the A/B of the fused linearisation,
// `make records`); the product library is built with 1
#ifndef GTG_FUSED_SFM
#define GTG_FUSED_SFM 1
#endif
namespace gt {

// gtg_prewarm's registry (kernels.h): filled by the static PrewarmUnit objects of the translation units
static std::vector<void (*)(int)>& prewarm_units() { static std::vector<void (*)(int)> v; return v; }
PrewarmUnit::PrewarmUnit(void (*fn)(int device)) { prewarm_units().push_back(fn); }

static thread_local std::string g_last_error;

void check_hip(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// The big buffers of a handle (>= 16 MB: the E slots, the term lists, the stored tiles of the reduced system) are kept for the next
// handle of the process when one is released instead of going back to the driver, at most GTG_ALLOC_CACHE_MB per DEVICE (default
// 2048 = about one handle of the headline size, 0.7 % of the device's memory; 0 switches it off); a kept block serves a request of
// 80 - 100 % of its size on the same device.  On boxes where the driver clears device memory as it hands it out a fresh hipMalloc costs
// ~30 ms per GB -- 10.9 of the 33 ms of a warm set-up of the L1723 shape in round 4, when the default was 0 -- and programs construct
// optimizers one after the other (GncOptimizer: one per outer iteration; the reference's own timing programs).  History of the
// default: 8192 in round 3 (the reduced system was a dense 1.9 GB - 31 GB array then), 0 in round 4, 2048 since round 5.  When an
// allocation fails, every kept block of that device is released and the allocation is tried once more;
// gtg_release_cached_memory() releases them at any time.  Every such buffer is fully written by the kernels before it is read, so
// recycled contents are never observed (GTG_ALLOC_POISON=1 fills a recycled block with NaNs first: a debug mode the parity suite
// can be run under).
namespace {
struct KeptBlock { void* p; size_t bytes; int device; };
std::mutex g_kept_mu;
std::vector<KeptBlock> g_kept;
constexpr size_t kKeepMin = (size_t)16 << 20;
size_t keep_limit() {
  static const size_t lim = [] { const char* e = std::getenv("GTG_ALLOC_CACHE_MB"); return (size_t)(e ? std::max(0L, std::atol(e)) : 2048L) << 20; }();
  return lim;
}
size_t kept_bytes_on(int dev) { size_t b = 0; for (const auto& k : g_kept) if (k.device == dev) b += k.bytes; return b; }   // (g_kept_mu held)
void* take_kept(size_t bytes, size_t* got) {
  if (bytes < kKeepMin || keep_limit() == 0) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_kept_mu);
  int best = -1;
  for (int i = 0; i < (int)g_kept.size(); i++)
    if (g_kept[i].device == dev && g_kept[i].bytes >= bytes && g_kept[i].bytes - bytes <= g_kept[i].bytes / 5 &&
        (best < 0 || g_kept[i].bytes < g_kept[best].bytes)) best = i;
  if (best < 0) return nullptr;
  void* q = g_kept[best].p;
  *got = g_kept[best].bytes;
  g_kept.erase(g_kept.begin() + best);
  static const bool poison = std::getenv("GTG_ALLOC_POISON") != nullptr;
  if (poison) (void)hipMemset(q, 0xFF, *got);   // all-ones bytes = a NaN in every double, -1 in every index
  return q;
}
bool keep_block(void* q, size_t bytes) {        // (hipFree synchronises the device; a kept block must be idle as well)
  if (bytes < kKeepMin || keep_limit() == 0) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lk(g_kept_mu);
  if (kept_bytes_on(dev) + bytes > keep_limit()) return false;
  if (hipDeviceSynchronize() != hipSuccess) return false;
  g_kept.push_back(KeptBlock{q, bytes, dev});
  return true;
}
size_t release_kept(int dev) {                  // dev < 0: every device
  std::lock_guard<std::mutex> lk(g_kept_mu);
  size_t freed = 0;
  for (size_t i = 0; i < g_kept.size();) {
    if (dev < 0 || g_kept[i].device == dev) {
      int cur = 0;
      const bool sw = hipGetDevice(&cur) == hipSuccess && cur != g_kept[i].device && hipSetDevice(g_kept[i].device) == hipSuccess;
      (void)hipFree(g_kept[i].p);
      if (sw) (void)hipSetDevice(cur);
      freed += g_kept[i].bytes;
      g_kept.erase(g_kept.begin() + (long)i);
    } else i++;
  }
  return freed;
}
}  // namespace

template <class T> void DevBuf<T>::alloc(size_t count) {
  free();
  n = count;
  if (!count) return;
  if (void* q = take_kept(sizeof(T) * count, &cap)) { p = static_cast<T*>(q); return; }
  hipError_t e = hipMalloc(&p, sizeof(T) * count);
  if (e != hipSuccess) {   // out of memory with blocks kept aside: give them back to the driver and try once more
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && release_kept(dev) > 0) e = hipMalloc(&p, sizeof(T) * count);
  }
  if (e != hipSuccess) { p = nullptr; n = 0; }
  check_hip(e, "hipMalloc");
  cap = sizeof(T) * count;
}
template <class T> void DevBuf<T>::upload(const T* host, size_t count, hipStream_t s) {
  if (count != n || (count && !p)) alloc(count);
  if (count) check_hip(hipMemcpyAsync(p, host, sizeof(T) * count, hipMemcpyHostToDevice, s), "H2D");
}
template <class T> void DevBuf<T>::free() {
  if (p && !keep_block(p, cap ? cap : sizeof(T) * n)) (void)hipFree(p);
  p = nullptr; n = 0; cap = 0;
}
template struct DevBuf<double>;
template struct DevBuf<int32_t>;
template struct DevBuf<int64_t>;
template struct DevBuf<long long>;
template struct DevBuf<unsigned char>;

static void exchange(gtg_context& c, double* ptr, int64_t n);
void exchange_sum(gtg_context& c, double* ptr, int64_t n) { exchange(c, ptr, n); }
static void exchange(gtg_context& c, double* ptr, int64_t n) {
  if (c.n_shards > 1) {
    if (!c.allreduce) throw std::runtime_error("n_shards > 1 but no allreduce callback was set (gtg_set_allreduce)");
    if (!c.layout_verified) verify_layout(c);   // the callback was registered after the upload
    const int rc = c.allreduce(ptr, n, (void*)c.stream, c.allreduce_user);
    if (rc != 0) throw std::runtime_error("allreduce callback failed");
  }
}

// The dataflow factorisation is a pair of persistent kernels that wait for each other.  Two of them in flight on one device (two
// handles driven from two host threads) can starve each other: the runtime multiplexes streams onto a few hardware queues, and
// handle A's chain kernel may sit behind handle B's bulk kernel in one queue while B's chain kernel sits behind A's bulk kernel
// in another -- neither pair completes until the wait bound breaks the cycle (seen with three handles: time-outs, never wrong
// numbers).  So a process runs one dataflow factorisation per device at a time: the lock is taken before the launch and released
// once the call has synchronised with its stream.
static std::mutex& df_device_lock(int device) {
  static std::mutex guard;
  static std::map<int, std::unique_ptr<std::mutex>> locks;
  std::lock_guard<std::mutex> g(guard);
  auto& p = locks[device];
  if (!p) p.reset(new std::mutex);
  return *p;
}

// Every entry point runs on the handle's device and leaves the caller's current device as it found it (a torch or multi-GPU host
// keeps its own notion of "current device").
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) check_hip(hipSetDevice(dev), "hipSetDevice"); else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Sharded: the scalars are partial sums and get all-reduced -- a COPY of them (second half of the buffer), so that slots a call
// does not rewrite are not multiplied by the number of shards on every call (they would overflow to inf after a few hundred).
static void read_scalars(gtg_context& c) {
  const double* src = c.scalars.p;
  if (c.n_shards > 1) {
    check_hip(hipMemcpyAsync(c.scalars.p + SC_COUNT, c.scalars.p, sizeof(double) * SC_COUNT, hipMemcpyDeviceToDevice, c.stream), "D2D");
    exchange(c, c.scalars.p + SC_COUNT, SC_COUNT);
    src = c.scalars.p + SC_COUNT;
  }
  check_hip(hipMemcpyAsync(c.h_scalars, src, sizeof(double) * SC_COUNT, hipMemcpyDeviceToHost, c.stream), "D2H");
  check_hip(hipStreamSynchronize(c.stream), "sync");
  c.h_scalars[SC_DELTA_SQ] /= c.n_shards;  // identical on every shard, summed by the exchange
}

// The kernels that follow LM's "is the linear cost change >= 0" (LevenbergMarquardtOptimizer.cpp:180-191: no error evaluation, hence no
// re-triangulation, of a trial step the model does not like) read the two linear errors on the device.  On a sharded graph every shard
// holds only its share of them: the sums are exchanged first, into a copy (read_scalars sums the originals once more, later).
static bool smart_gate(gtg_context& c) {
  if (!c.n_smart || c.n_shards == 1) return false;
  check_hip(hipMemcpyAsync(c.scalars.p + 2 * SC_COUNT, c.scalars.p, sizeof(double) * SC_COUNT, hipMemcpyDeviceToDevice, c.stream), "D2D");
  exchange(c, c.scalars.p + 2 * SC_COUNT, SC_COUNT);
  return true;
}

static void check_smart_supported(gtg_context& c, const char* where) {
  if (!c.n_smart || c.h_scalars[SC_UNSUPPORTED] == 0.0) return;
  // the two places where the reference's smart factor throws out of linearize() / error() instead of returning a number
  if (c.h_scalars[SC_UNSUPPORTED] >= kUnsupportedCheirality)      // (summed over the shards: calibrate counts 1 per shard)
    throw std::runtime_error(std::string(where) + ": CheiralityException -- a smart factor's point at infinity (IGNORE_DEGENERACY / HANDLE_INFINITY "
                             "with a landmark that did not triangulate) lies behind one of the factor's cameras, or the refinement of a "
                             "triangulation (enableEPI) linearised at a point behind a camera; the reference throws here");
  throw std::runtime_error(std::string(where) + ": Cal3Bundler::calibrate did not converge for a measurement of a smart factor; the reference throws here");
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
// A handle's stream and events outlive it: gtg_destroy parks them (idle) per device and the next gtg_create of the process takes them from
// there -- creating a stream and 18 events cost 1.7 ms of every construction of an optimizer (tools/cpp/cold_start_probe.cpp), a third of
// what the whole symbolic analysis of the L1723 shape takes now.  At most kParkedMax sets per device are kept; gtg_release_cached_memory()
// destroys them with the cached device memory.
struct ParkedQueue { int device; hipStream_t stream; hipStream_t copy_stream; std::vector<hipEvent_t> events; };
static std::mutex g_parked_mu;
static std::vector<ParkedQueue> g_parked;
constexpr size_t kParkedMax = 4;
static bool take_parked(gtg_context& c) {
  if (std::getenv("GTG_NO_PARKED_STREAMS")) return false;   // (A/B: every handle creates its own stream and events)
  std::lock_guard<std::mutex> lk(g_parked_mu);
  for (size_t i = 0; i < g_parked.size(); i++)
    if (g_parked[i].device == c.device) {
      c.stream = g_parked[i].stream; c.copy_stream = g_parked[i].copy_stream; c.phase_events = std::move(g_parked[i].events);
      g_parked.erase(g_parked.begin() + (long)i);
      return true;
    }
  return false;
}
static bool park_queue(gtg_context& c) {   // (the caller has synchronised the streams)
  if (std::getenv("GTG_NO_PARKED_STREAMS")) return false;
  std::lock_guard<std::mutex> lk(g_parked_mu);
  size_t n = 0;
  for (const auto& q : g_parked) n += q.device == c.device;
  if (n >= kParkedMax) return false;
  g_parked.push_back(ParkedQueue{c.device, c.stream, c.copy_stream, std::move(c.phase_events)});
  c.stream = nullptr; c.copy_stream = nullptr; c.phase_events.clear();
  return true;
}
static void destroy_parked() {
  std::lock_guard<std::mutex> lk(g_parked_mu);
  int cur = 0; (void)hipGetDevice(&cur);
  for (auto& q : g_parked) {
    (void)hipSetDevice(q.device);
    for (hipEvent_t e : q.events) if (e) (void)hipEventDestroy(e);
    if (q.stream) (void)hipStreamDestroy(q.stream);
    if (q.copy_stream) (void)hipStreamDestroy(q.copy_stream);
  }
  g_parked.clear();
  (void)hipSetDevice(cur);
}
static void collect(gtg_context& c, std::initializer_list<int> phases) {
  if (!c.timing) return;
  for (int ph : phases) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c.phase_events[2 * ph], c.phase_events[2 * ph + 1]) == hipSuccess) { c.phase_ms[ph] += ms; c.phase_calls[ph]++; }
  }
}

}  // namespace gt

namespace gt { long long* g_potrf_dbg_set(long long*); float debug_time_syrk(gtg_context&, SMat, int, int, int); }
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
  if (!take_parked(*c)) {
    check_hip(hipStreamCreate(&c->stream), "hipStreamCreate");
    check_hip(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking), "hipStreamCreate");   // uploads beside the analysis (gtg_upload_problem)
    ensure_events(*c);
  }
  *out = c;
  return GTG_OK;
  GTG_CATCH
}

// The one-time work of a process on a device, done ahead of the first handle: runtime start and device context, the code objects of
// the library's translation units and the function objects of their kernels (every unit registers its list, kernels.h), the default
// pair of CU-masked streams of the dataflow factorisation, a first stream / event.  Idempotent per device; safe to call from a helper
// thread while the caller prepares its problem (the C++ shim's constructor does: GpuLevenbergMarquardtOptimizer.cpp).
int gtg_prewarm(int device_id) {
  GTG_TRY
  static std::mutex mu;
  static std::set<int> done;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count(device_id)) return GTG_OK;
  int ndev = 0;
  check_hip(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
  if (device_id < 0 || device_id >= ndev) throw std::invalid_argument("bad device id (no HIP device visible?)");
  check_hip(hipSetDevice(device_id), "hipSetDevice");
  (void)hipFree(nullptr);
  hipStream_t st = nullptr; hipEvent_t ev = nullptr;
  check_hip(hipStreamCreate(&st), "hipStreamCreate");
  check_hip(hipEventCreate(&ev), "hipEventCreate");
  for (auto fn : prewarm_units()) fn(device_id);
  (void)hipEventRecord(ev, st);
  (void)hipStreamSynchronize(st);
  (void)hipEventDestroy(ev);
  (void)hipStreamDestroy(st);
  done.insert(device_id);
  return GTG_OK;
  GTG_CATCH
}

int gtg_destroy(gtg_handle c) {
  if (!c) return GTG_OK;
  try { join_block_level(*c); } catch (...) {}
  try { df_join_prepared(); } catch (...) {}
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  // The releases below run while no dataflow factorisation of ANOTHER handle is in flight on this device: a handle that was destroyed
  // beside a running factorisation stalled its persistent kernels past the 20 ms bound of their dependency waits (three of three time-outs of
  // the round's last multi-handle stress runs sat in the second-to-last factorisation of a handle whose neighbour had just finished:
  // profiles/r06_stress_240s.txt).  GTG_DESTROY_UNLOCKED=1: the A/B.
  std::unique_lock<std::mutex> no_factorisation_in_flight;
  if (!std::getenv("GTG_DESTROY_UNLOCKED")) no_factorisation_in_flight = std::unique_lock<std::mutex>(df_device_lock(c->device));
  auto& f = c->f;
  DevBuf<double>* dbl[] = {&c->values, &c->trial, &c->delta, &c->noise_data, &f.sfm_z, &f.sfm_J, &f.proj_z, &f.proj_J,
                           &f.calib, &f.sensor, &f.between_z, &f.between_J, &f.prior_data, &f.prior_J, &c->Hd, &c->gred0,
                           &c->hdiag_red, &c->V, &c->gp, &c->Hoff, &c->Linv, &c->ylm, &c->E, &c->vobs, &c->wobs, &c->cam_part, &c->cam_pack, &c->pcg_vec, &c->pcg_bj, &c->pcg_y, &c->delta_lm, &c->S,
                           &c->Dinv, &c->xred, &c->partials, &c->scalars, &c->noise_rk};
  for (auto* b : dbl) b->free();
  DevBuf<int32_t>* i32[] = {&c->var_type, &c->lm_var, &c->red_var, &c->red_dim, &c->lm_index, &c->red_index, &c->lm_owned,
                            &c->noise_kind, &c->noise_rkind, &f.sfm_cam, &f.sfm_point, &f.sfm_noise, &f.sfm_cam_at, &f.sfm_point_at, &c->obs_wpos, &f.proj_pose, &f.proj_point,
                            &f.proj_noise, &f.proj_calib, &f.proj_sensor, &f.between_v1, &f.between_v2, &f.between_noise,
                            &f.prior_var, &f.prior_noise, &c->obs_red, &c->obs_lm, &c->lm_obs, &c->lm_pri,
                            &c->red_inc_kind, &c->red_inc_idx, &c->hoff_row, &c->hoff_col, &c->hoff_fac, &c->pair_row,
                            &c->pair_col, &c->pair_oa, &c->pair_ob, &c->smart_status, &c->smart_lin_status, &c->smart_cache_state, &c->sfm_smart, &c->lm_smart};
  for (auto* b : i32) b->free();
  c->plan.rows.free(); c->plan.pairs.free(); c->plan.bcols.free(); c->plan.bwd_col_off.free(); c->plan.bwd_col_rows.free(); c->plan.stored.free(); c->plan.slot.free(); c->yred.free(); c->plan.exch.free(); c->xbuf.free();
  DevBuf<int64_t>* i64[] = {&c->val_off, &c->dim_off, &c->red_off, &c->noise_off, &f.prior_off, &c->lm_obs_ptr,
                            &c->lm_pri_ptr, &c->red_inc_ptr, &c->hoff_ptr, &c->pair_ptr, &c->smart_ptr, &c->pad_index};
  for (auto* b : i64) b->free();
  c->smart_params.free(); c->smart_cache_pose.free(); c->smart_cache_point.free();
  c->chol_epoch_dev.free(); c->layout_probe.free(); c->xb_row_off.free(); c->xb_col_off.free(); c->xb_dim.free();
  free_df_plan(c->df);
  c->pivot_kind.free(); c->tile_exp.free();
  destroy_chol_streams(*c);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  if (!park_queue(*c)) {
    for (hipEvent_t e : c->phase_events) if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  }
  drop_index(c);
  delete c;
  return GTG_OK;
}

int gtg_upload_problem(gtg_handle c, const gtg_problem* p_user, int shard, int n_shards) {
  GTG_TRY
  if (!c || !p_user) throw std::invalid_argument("null argument");
  if (n_shards < 1 || shard < 0 || shard >= n_shards) throw std::invalid_argument("bad shard / n_shards");
  DeviceGuard on_device(c->device);
  StageClock clk;
  hipStream_t s = c->stream;
  c->shard = shard; c->n_shards = n_shards;
  // Smart factors become what the rest of the library already knows: a hidden POINT3 variable per factor behind the caller's
  // variables and one GeneralSFM observation per measurement behind the caller's; the tables below are built from this view.
  gtg_problem q = *p_user;
  const gtg_problem* p = &q;
  std::vector<int32_t> x_var_type, x_sfm_cam, x_sfm_point, x_sfm_noise, x_of_obs;
  std::vector<double> x_sfm_z;
  const int64_t n_smart = p_user->n_smart > 0 ? p_user->n_smart : 0;
  c->n_smart = n_smart; c->n_user_vars = p_user->n_vars; c->smart_obs0 = p_user->n_sfm;
  // GeneralSFM records recomputed where they are needed instead of stored (fused.h) -- not with smart factors, whose measurements'
  // records depend on the factor's triangulation status
  c->fused_sfm = GTG_FUSED_SFM != 0 && n_smart == 0;
  if (n_smart) {
    if (!p_user->smart_ptr || !p_user->smart_cam || !p_user->smart_z || !p_user->smart_noise || !p_user->smart_params)
      throw std::invalid_argument("smart factor tables missing");
    const int64_t n_meas = p_user->smart_ptr[n_smart];
    x_var_type.assign(p_user->var_type, p_user->var_type + p_user->n_vars); x_var_type.resize((size_t)p_user->n_vars + n_smart, GTG_VAR_POINT3);
    x_sfm_cam.assign(p_user->sfm_cam, p_user->sfm_cam + p_user->n_sfm); x_sfm_point.assign(p_user->sfm_point, p_user->sfm_point + p_user->n_sfm);
    x_sfm_noise.assign(p_user->sfm_noise, p_user->sfm_noise + p_user->n_sfm); x_sfm_z.assign(p_user->sfm_z, p_user->sfm_z + 2 * p_user->n_sfm);
    std::vector<int32_t> of_obs((size_t)p_user->n_sfm, -1);
    // the device addresses a factor's measurements as smart_obs0 + smart_ptr[i] while the expanded observations are appended one
    // after the other: the offsets must start at 0 and be strictly increasing
    if (p_user->smart_ptr[0] != 0) throw std::invalid_argument("smart factors: smart_ptr[0] must be 0");
    for (int64_t i = 0; i < n_smart; i++) {
      const int64_t k0 = p_user->smart_ptr[i], k1 = p_user->smart_ptr[i + 1];
      if (k1 <= k0 || k1 > n_meas) throw std::invalid_argument("smart factor without measurements / bad smart_ptr");
      const double* sp = p_user->smart_params + 8 * i;
      if (!(sp[4] == 0.0 || sp[4] == 1.0 || sp[4] == 2.0)) throw std::invalid_argument("smart factor: unknown degeneracy mode");
      if (!(sp[6] == 0.0 || sp[6] == 1.0)) throw std::invalid_argument("smart factor: enableEPI must be 0 or 1");
      // LinearizationMode (SmartFactorParams.h:31-33): HESSIAN, JACOBIAN_Q, JACOBIAN_SVD give the same normal equations (they differ in
      // what a failed track contributes and in the constant of the linear error); IMPLICIT_SCHUR factors cannot be eliminated by
      // the reference's direct solvers at all (RegularImplicitSchurFactor has no augmentedJacobian / augmentedInformation)
      if (!(sp[5] == 0.0 || sp[5] == 2.0 || sp[5] == 3.0)) throw std::invalid_argument("smart factor: linearization mode must be 0 HESSIAN, 2 JACOBIAN_Q or 3 JACOBIAN_SVD");
      // rankTolerance, landmarkDistanceThreshold, dynamicOutlierRejectionThreshold (negative = off, as in the reference),
      // retriangulationThreshold: numbers, not NaN / inf (a NaN threshold silently disables the test it guards)
      for (int e = 0; e < 4; e++) if (!std::isfinite(sp[e])) throw std::invalid_argument("smart factor: a threshold is not finite");
      // the reference requires an isotropic model (SmartFactorBase.h:107-114: "SmartFactorBase: needs isotropic")
      const int32_t nz = p_user->smart_noise[i];
      if (nz < 0 || nz >= p_user->n_noise || p_user->noise_dim[nz] != 2 ||
          !(p_user->noise_kind[nz] == GTG_NOISE_UNIT || p_user->noise_kind[nz] == GTG_NOISE_ISOTROPIC))
        throw std::invalid_argument("smart factor: smart_noise must index a dim-2 Unit or Isotropic noise model");
      for (int64_t k = k0; k < k1; k++) {
        const int cam = p_user->smart_cam[k];
        if (cam < 0 || cam >= p_user->n_vars || p_user->var_type[cam] != GTG_VAR_SFM_CAMERA)
          throw std::invalid_argument("smart factor: its keys must be SFM_CAMERA variables");
        x_sfm_cam.push_back(cam); x_sfm_point.push_back((int32_t)(p_user->n_vars + i)); x_sfm_noise.push_back(p_user->smart_noise[i]);
        x_sfm_z.push_back(p_user->smart_z[2 * k]); x_sfm_z.push_back(p_user->smart_z[2 * k + 1]);
        of_obs.push_back((int32_t)i);
      }
    }
    q.n_vars = (int32_t)x_var_type.size(); q.var_type = x_var_type.data();
    q.n_sfm = (int64_t)x_sfm_cam.size(); q.sfm_cam = x_sfm_cam.data(); q.sfm_point = x_sfm_point.data();
    q.sfm_noise = x_sfm_noise.data(); q.sfm_z = x_sfm_z.data();
    std::vector<double> prm(p_user->smart_params, p_user->smart_params + 8 * n_smart);
    up(c->smart_params, prm, s);
    x_of_obs = std::move(of_obs);           // (smart_ptr and sfm_smart follow the shard filter of the observation table below)
    std::vector<int32_t> none((size_t)n_smart, -1);
    up(c->smart_cache_state, none, s);
    c->smart_status.alloc((size_t)n_smart); c->smart_lin_status.alloc((size_t)n_smart); c->smart_cache_point.alloc(3 * (size_t)n_smart); c->smart_cache_pose.alloc(12 * (size_t)n_meas);
    check_hip(hipMemsetAsync(c->smart_status.p, 0, sizeof(int32_t) * n_smart, s), "memset");
    check_hip(hipMemsetAsync(c->smart_lin_status.p, 0, sizeof(int32_t) * n_smart, s), "memset");
  } else {
    c->smart_ptr.free(); c->smart_params.free(); c->sfm_smart.free(); c->lm_smart.free(); c->smart_status.free(); c->smart_lin_status.free();
    c->smart_cache_state.free(); c->smart_cache_pose.free(); c->smart_cache_point.free();
  }
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
  c->user_val_size = c->h_val_off[c->n_user_vars]; c->user_dim_size = c->h_dim_off[c->n_user_vars];
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
      if (rkind[i] < GTG_ROBUST_NONE || rkind[i] > GTG_ROBUST_L2WITHDEADZONE) throw std::invalid_argument("unsupported m-estimator");
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
  std::thread side_upload; std::exception_ptr side_err;
  struct JoinSide { std::thread& t; ~JoinSide() { if (t.joinable()) t.join(); } } join_side{side_upload};
  { // SFM
    std::vector<int32_t> cam, pt, nz; std::vector<double> z;
    for (int64_t i = 0; i < p->n_sfm; i++) { check_var(p->sfm_cam[i]); check_var(p->sfm_point[i]); check_noise(p->sfm_noise[i], 2, "GeneralSFMFactor"); }
    const bool whole = n_shards == 1 && p->n_sfm > 0;   // the whole table: noise rows and measurements go up straight from the caller's arrays
    if (n_shards == 1) {
      cam.assign(p->sfm_cam, p->sfm_cam + p->n_sfm); pt.assign(p->sfm_point, p->sfm_point + p->n_sfm);   // (kept by the host index: gtg_set_reduced_ordering analyses again)
      if (!whole) { nz.assign(p->sfm_noise, p->sfm_noise + p->n_sfm); z.assign(p->sfm_z, p->sfm_z + 2 * p->n_sfm); }
    } else {
      for (int64_t i = 0; i < p->n_sfm; i++) {
        if (!own_lm(p->sfm_point[i])) continue;
        cam.push_back(p->sfm_cam[i]); pt.push_back(p->sfm_point[i]); nz.push_back(p->sfm_noise[i]);
        z.push_back(p->sfm_z[2 * i]); z.push_back(p->sfm_z[2 * i + 1]);
      }
    }
    if (n_smart) {
      // Sharded, a smart factor follows its hidden landmark like any landmark factor: this shard holds the measurements of the
      // tracks it owns, in the order of the whole table.  The per-factor arrays keep the GLOBAL factor index (parameters, status,
      // cache); a factor of another shard has no measurements here (smart_ptr[i + 1] == smart_ptr[i]) and is skipped.
      std::vector<int64_t> rel((size_t)n_smart + 1, 0);
      std::vector<int32_t> of_local;
      int64_t obs0 = 0;
      for (int64_t i = 0; i < p->n_sfm; i++) {
        if (n_shards > 1 && !own_lm(p->sfm_point[i])) continue;
        const int32_t sf = x_of_obs[(size_t)i];
        of_local.push_back(sf);
        if (sf < 0) obs0++; else rel[(size_t)sf + 1]++;
      }
      for (int64_t i = 0; i < n_smart; i++) rel[(size_t)i + 1] += rel[(size_t)i];
      c->smart_obs0 = obs0;
      up(c->smart_ptr, rel, s);
      up(c->sfm_smart, of_local, s);
    }
    f.n_sfm = (int64_t)cam.size();
    up(f.sfm_cam, cam, s); up(f.sfm_point, pt, s);
    if (whole) {
      // the noise rows and the measurements (20 bytes per factor: 13.5 MB on the L1723 shape, 1.2 ms from pageable memory) are not read by
      // the symbolic analysis: they go up on a helper thread and the handle's copy stream beside it (joined below, before this call returns)
      f.sfm_noise.alloc((size_t)p->n_sfm); f.sfm_z.alloc(2 * (size_t)p->n_sfm);
      const int dev = c->device; hipStream_t cs = c->copy_stream;
      int32_t* d_nz = f.sfm_noise.p; double* d_z = f.sfm_z.p;
      const int32_t* h_nz = p->sfm_noise; const double* h_z = p->sfm_z; const size_t n = (size_t)p->n_sfm;
      side_upload = std::thread([dev, cs, d_nz, d_z, h_nz, h_z, n, &side_err] {
        try {
          check_hip(hipSetDevice(dev), "hipSetDevice");
          check_hip(hipMemcpyAsync(d_nz, h_nz, sizeof(int32_t) * n, hipMemcpyHostToDevice, cs), "H2D");
          check_hip(hipMemcpyAsync(d_z, h_z, sizeof(double) * 2 * n, hipMemcpyHostToDevice, cs), "H2D");
          check_hip(hipStreamSynchronize(cs), "sync");
        } catch (...) { side_err = std::current_exception(); }
      });
    }
    else { up(f.sfm_noise, nz, s); up(f.sfm_z, z, s); }
    if (c->val_size >= (int64_t)1 << 31) throw std::invalid_argument("gtg_upload_problem: more than 2^31 packed value entries");
    f.sfm_cam_at.alloc(std::max<size_t>(cam.size(), 1)); f.sfm_point_at.alloc(std::max<size_t>(pt.size(), 1));
    launch_sfm_value_offsets(*c);     // where each factor's camera / point start in the packed values (a gather through val_off, on the device)
    f.sfm_J.alloc(c->fused_sfm ? 1 : std::max<size_t>((size_t)kSfmRec * f.n_sfm, 1));
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
    // device calibration table: 9 per entry, fx fy s u0 v0 k1 k2 p1 p2 (the distortion part zero for a Cal3_S2)
    std::vector<double> calib(kCalibStride * (size_t)p->n_calib, 0.0), sensor(p->sensor, p->sensor + 12 * (size_t)p->n_sensor);
    for (int32_t k = 0; k < p->n_calib; k++) {
      for (int j = 0; j < 5; j++) calib[kCalibStride * (size_t)k + j] = p->calib[5 * (size_t)k + j];
      if (p->calib_distortion) for (int j = 0; j < 4; j++) calib[kCalibStride * (size_t)k + 5 + j] = p->calib_distortion[4 * (size_t)k + j];
    }
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
  if (side_upload.joinable()) side_upload.join();
  if (side_err) std::rethrow_exception(side_err);
  if (c->n_smart) {   // landmark index of every smart factor's hidden variable
    std::vector<int32_t> lm_smart((size_t)std::max(c->n_lm, 1), -1);
    for (int64_t i = 0; i < c->n_smart; i++) {
      const int l = c->h_lm_index[c->n_user_vars + i];
      if (l < 0) throw std::runtime_error("smart factor: its hidden landmark was not classified as a landmark");
      lm_smart[(size_t)l] = (int32_t)i;
    }
    up(c->lm_smart, lm_smart, s);
    check_hip(hipStreamSynchronize(s), "sync");
  }
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

int64_t gtg_values_size(gtg_handle c) { return c ? c->user_val_size : -1; }
int64_t gtg_tangent_size(gtg_handle c) { return c ? c->user_dim_size : -1; }
int64_t gtg_reduced_dim(gtg_handle c) { return c ? c->n_red : -1; }

int gtg_set_values(gtg_handle c, const double* packed, int64_t n) {
  GTG_TRY
  if (!c || !c->uploaded || n != c->user_val_size) throw std::invalid_argument("gtg_set_values: wrong size or no problem uploaded");
  DeviceGuard on_device(c->device);
  check_hip(hipMemcpyAsync(c->values.p, packed, sizeof(double) * n, hipMemcpyHostToDevice, c->stream), "H2D");
  check_hip(hipStreamSynchronize(c->stream), "sync");
  c->linearized = false; c->have_trial = false;
  return GTG_OK;
  GTG_CATCH
}

static int get_buf(gtg_handle c, const double* dev, int64_t have, double* out, int64_t n) {
  GTG_TRY
  if (!c || !c->uploaded || n != have) throw std::invalid_argument("getter: wrong size or no problem uploaded");
  DeviceGuard on_device(c->device);
  check_hip(hipMemcpyAsync(out, dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream), "D2H");
  check_hip(hipStreamSynchronize(c->stream), "sync");
  return GTG_OK;
  GTG_CATCH
}
int gtg_get_values(gtg_handle c, double* packed, int64_t n) { return get_buf(c, c ? c->values.p : nullptr, c ? c->user_val_size : -1, packed, n); }
int gtg_get_trial_values(gtg_handle c, double* packed, int64_t n) { return get_buf(c, c ? c->trial.p : nullptr, c ? c->user_val_size : -1, packed, n); }
int gtg_get_delta(gtg_handle c, double* d, int64_t n) { return get_buf(c, c ? c->delta.p : nullptr, c ? c->user_dim_size : -1, d, n); }

int gtg_error(gtg_handle c, double* error) {
  GTG_TRY
  if (!c || !c->uploaded || !error) throw std::invalid_argument("gtg_error: no problem uploaded");
  DeviceGuard on_device(c->device);
  if (c->n_smart) check_hip(hipMemsetAsync(c->scalars.p + SC_UNSUPPORTED, 0, sizeof(double), c->stream), "memset");
  { PhaseTimer t(*c, GTG_PH_ERROR, c->phase_events.data()); launch_smart_triangulate(*c, c->values.p, nullptr, false); launch_error(*c, c->values.p, SC_ERROR); }
  read_scalars(*c);
  collect(*c, {GTG_PH_ERROR});
  check_smart_supported(*c, "gtg_error");
  *error = c->h_scalars[SC_ERROR];
  return GTG_OK;
  GTG_CATCH
}

// GTG_DEBUG_TIMING=1: the host time of every phase of the FIRST linearisation and the FIRST lambda try of the process, each behind a stream
// synchronisation -- what a cold process pays there once (40 - 50 ms on the L1723 shape against 6 ms for every later iteration).
struct FirstCallClock {
  bool on; hipStream_t s; std::chrono::high_resolution_clock::time_point t;
  FirstCallClock(std::atomic<int>& calls, hipStream_t stream) : on(std::getenv("GTG_DEBUG_TIMING") != nullptr && calls.fetch_add(1) == 0), s(stream), t(std::chrono::high_resolution_clock::now()) {}
  void lap(const char* what) {
    if (!on) return;
    (void)hipStreamSynchronize(s);
    const auto n = std::chrono::high_resolution_clock::now();
    std::fprintf(stderr, "[gtsam_amd first ] %-40s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};
static std::atomic<int> g_first_linearize{0}, g_first_try{0};

int gtg_linearize(gtg_handle c) {
  GTG_TRY
  if (!c || !c->uploaded) throw std::invalid_argument("gtg_linearize: no problem uploaded");
  DeviceGuard on_device(c->device);
  if (c->n_smart) check_hip(hipMemsetAsync(c->scalars.p + SC_UNSUPPORTED, 0, sizeof(double), c->stream), "memset");
  FirstCallClock first(g_first_linearize, c->stream);
  { PhaseTimer t(*c, GTG_PH_LINEARIZE, c->phase_events.data()); launch_smart_triangulate(*c, c->values.p, nullptr, true); launch_linearize(*c); }
  first.lap("linearize");
  { PhaseTimer t(*c, GTG_PH_ASSEMBLE, c->phase_events.data()); launch_assemble(*c);
    if (c->n_smart) {   // the cameras' Hessian diagonal is that of the Schur-complemented smart factors: needs their E blocks (undamped)
      launch_point_eliminate(*c, 1.0, 0, 1e-6, 1e32);
      launch_smart_hdiag(*c);
    } }
  first.lap("assemble");
  exchange(*c, c->hdiag_red.p, c->NP);   // damping needs the full diagonal on every shard
  if (c->n_smart) {   // (sharded: every shard must see what any shard met)
    if (c->n_shards > 1) exchange(*c, c->scalars.p + SC_UNSUPPORTED, 1);
    check_hip(hipMemcpyAsync(c->h_scalars + SC_UNSUPPORTED, c->scalars.p + SC_UNSUPPORTED, sizeof(double), hipMemcpyDeviceToHost, c->stream), "D2H");
  }
  check_hip(hipStreamSynchronize(c->stream), "sync");
  collect(*c, {GTG_PH_LINEARIZE, GTG_PH_ASSEMBLE});
  check_smart_supported(*c, "gtg_linearize");
  c->linearized = true;
  return GTG_OK;
  GTG_CATCH
}

int gtg_try_lambda(gtg_handle c, double lambda, int diag, double dmin, double dmax, double out[4]) {
  GTG_TRY
  if (!c || !c->uploaded || !c->linearized) throw std::invalid_argument("gtg_try_lambda: call gtg_linearize first");
  if (!(lambda > 0.0)) throw std::invalid_argument("gtg_try_lambda: lambda must be > 0");
  DeviceGuard on_device(c->device);
  // The factorisation's kernels wait for each other inside a launch (dataflow schedule: two persistent kernels; stream schedule: the
  // TRSM workgroups of a panel launch).  A dependency wait that runs into its bound raises SC_TIMEOUT and the kernels drain: a chain
  // kernel that was not placed, a device shared with another process, or -- seen with several handles on one device -- a flag that an
  // XCD's L2 kept serving with its old value although the flags are published twice (chol_dataflow.hip::st_flag).  The try is then
  // repeated: first with the SAME schedule (a stuck wait is a property of one pass, not of the problem, and the repeat returns the
  // same bits as an undisturbed try: the trajectory does not depend on whether a wait timed out), then, should that time out as
  // well, with the other schedule (stream / event launches of cholesky.hip: kernels that never wait for a kernel launched after
  // them; its plan is always built; with one chain it runs the same sums in the same order, with several chains the cross-part
  // updates are summed in another order: the same numbers to rounding).  The reduced system is assembled again each time, because
  // the factorisation works in place.
  // Sharded: the scalars are summed over the shards by read_scalars, so every shard sees the time-out of any shard and all repeat.
  // (GTG_CHOL=streams: there is no other schedule to fall back to -- one repeat, then the error)
  const int max_attempts = c->use_df ? 3 : 2;
  FirstCallClock first(g_first_try, c->stream);
  for (int attempt = 0; attempt < max_attempts; attempt++) {
    const bool df = c->use_df && attempt != 2;
    check_hip(hipMemsetAsync(c->scalars.p + SC_FAIL, 0, 3 * sizeof(double), c->stream), "memset");
    { PhaseTimer t(*c, GTG_PH_POINT_ELIM, c->phase_events.data()); launch_point_eliminate(*c, lambda, diag, dmin, dmax); }
    first.lap("point elimination");
    { PhaseTimer t(*c, GTG_PH_SCHUR, c->phase_events.data()); launch_build_reduced(*c, lambda, diag, dmin, dmax); }
    first.lap("reduced system");
    if (c->n_shards > 1) {   // the one big exchange: reduced Hessian + rhs
      if (c->n_xb == 0) {     // (no block list: whole stored 128x128 tiles, the first version of the exchange)
        const int64_t nb = c->plan.n_exch * kTile * kTile;
        if ((int64_t)c->xbuf.n != nb) c->xbuf.alloc(nb);
        launch_pack_tiles(*c, smat(*c), c->plan, c->xbuf.p, false);
        exchange(*c, c->xbuf.p, nb);
        launch_pack_tiles(*c, smat(*c), c->plan, c->xbuf.p, true);
      } else {                // only the structurally non-zero d x d blocks (the same list on every shard) + rhs row + padding
        const int64_t nb = exchange_block_doubles(*c);
        if ((int64_t)c->xbuf.n != nb) c->xbuf.alloc(nb);
        launch_pack_blocks(*c, smat(*c), c->NP, c->xbuf.p, false);
        exchange(*c, c->xbuf.p, nb);
        launch_pack_blocks(*c, smat(*c), c->NP, c->xbuf.p, true);
      }
    }
    std::unique_lock<std::mutex> one_at_a_time;
    if (df) one_at_a_time = std::unique_lock<std::mutex>(df_device_lock(c->device));
    { PhaseTimer t(*c, GTG_PH_CHOLESKY, c->phase_events.data());
      if (df) launch_cholesky_df(*c, smat(*c), c->NP, c->df, c->Dinv.p, c->scalars.p + SC_FAIL, c->pivot_kind.p, c->tile_exp.p);
      else launch_cholesky(*c, smat(*c), c->NP, c->plan, c->Dinv.p, c->scalars.p + SC_FAIL, c->pivot_kind.p, c->tile_exp.p); }
    first.lap("factorisation");
    { PhaseTimer t(*c, GTG_PH_SOLVE, c->phase_events.data());
      launch_backward_solve(*c, smat(*c), c->NP, c->plan, c->Dinv.p, c->xred.p, c->scalars.p + SC_FAIL);
      launch_back_substitute(*c);
      first.lap("solves");
      if (c->n_shards > 1 && one_at_a_time.owns_lock()) {   // sharded: the exchange below may wait for another handle of this
        check_hip(hipStreamSynchronize(c->stream), "sync");   // process (two shards on one device in the tests): the factorisation is
        one_at_a_time.unlock();                                // done, let the other one start before waiting for it
      }
      if (c->n_lm) exchange(*c, c->delta_lm.p, 3 * (int64_t)c->n_lm);
      launch_scatter_delta(*c); }
    { PhaseTimer t(*c, GTG_PH_LINEAR_ERROR, c->phase_events.data()); launch_linear_error(*c); launch_smart_lin1(*c); }
    { PhaseTimer t(*c, GTG_PH_RETRACT, c->phase_events.data()); launch_retract(*c); }
    { PhaseTimer t(*c, GTG_PH_ERROR, c->phase_events.data()); const double* gate = c->scalars.p + (smart_gate(*c) ? 2 * SC_COUNT : 0);
      launch_smart_triangulate(*c, c->trial.p, gate, false); launch_error(*c, c->trial.p, SC_TRIAL_ERROR, gate); }
    first.lap("linear error, retract, error");
    read_scalars(*c);
    if (one_at_a_time.owns_lock()) one_at_a_time.unlock();
    if (attempt + 1 < max_attempts && c->h_scalars[SC_TIMEOUT] != 0.0) {
      static const bool quiet = std::getenv("GTG_QUIET") != nullptr;
      if (!quiet) {
        // post-mortem: which wait gave up (chol_dataflow.hip::wait_flags records the first one of a dataflow pass), what it saw, and
        // what the same words hold in memory NOW, read from the host after the kernels have drained
        int32_t ctl[16] = {0};
        long long now1 = -1, now2 = -1;
        const int nt = c->NP / kTile;
        if (df && c->df.ctrl.p) {
          (void)hipMemcpy(ctl, c->df.ctrl.p, sizeof(ctl), hipMemcpyDeviceToHost);
          const int kind = ctl[8], I = ctl[9], J = ctl[10], k = ctl[11];
          const long long *w1 = nullptr, *w2 = nullptr;
          // (flag words are indexed by tile slot; kinds 1 / 2 record the SLOT of the first operand tile in k)
          auto slot_of = [&](int a, int b) { return (a >= 0 && a <= nt && b >= 0 && b < nt) ? (int64_t)c->plan.h_slot[(size_t)a * nt + b] : (int64_t)-1; };
          if (kind == 1 || kind == 2) { if (k >= 0 && k < c->plan.n_stored) w1 = w2 = c->df.tile_flag.p + k; }
          else if (kind == 3 && slot_of(J, J) >= 0) w1 = w2 = c->df.tile_flag.p + slot_of(J, J);
          else if (kind == 4) w1 = w2 = c->df.pd_flag.p + I;
          else if (kind == 5 && slot_of(I, J) >= 0) w1 = w2 = c->df.tile_flag.p + slot_of(I, J);
          else if (kind == 7 && slot_of(I, J) >= 0) w1 = w2 = c->df.part_flag.p + slot_of(I, J);
          if (w1) { (void)hipMemcpy(&now1, w1, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&now2, w2, 8, hipMemcpyDeviceToHost); }
        }
        std::fprintf(stderr, "[gtsam_amd] %s factorisation (epoch %lld): a dependency wait ran into its bound; repeating the lambda try with the %s "
                     "schedule.  wait kind %d at (%d, %d, %d): saw %d / %d, wanted %d / %d; memory now holds %lld / %lld; waiter on XCD %d (hw id 0x%x); "
                     "tickets taken %d, diagonal tiles started %d of %d; over this handle's life: waits that ended on the shadow words %d, on the read-modify-write poll %d\n",
                     df ? "dataflow" : "stream-schedule", c->chol_epoch, (c->use_df && attempt + 1 != 2) ? "dataflow" : "stream", ctl[8], ctl[9], ctl[10], ctl[11], ctl[12], ctl[13], ctl[14],
                     ctl[15], now1, now2, ctl[2], (unsigned)ctl[3], ctl[0], ctl[1], nt, ctl[6], ctl[7]);
      }
      c->df_fallbacks++;
      continue;
    }
    break;
  }
  collect(*c, {GTG_PH_POINT_ELIM, GTG_PH_SCHUR, GTG_PH_CHOLESKY, GTG_PH_SOLVE, GTG_PH_LINEAR_ERROR, GTG_PH_RETRACT, GTG_PH_ERROR});
  c->have_trial = true;
  const double dsq = c->h_scalars[SC_DELTA_SQ];
  check_smart_supported(*c, "gtg_try_lambda");
  if (c->h_scalars[SC_TIMEOUT] != 0.0) throw std::runtime_error("gtg_try_lambda: a dependency wait of the factorisation ran into its bound (GPU shared or preempted?); the step was not computed");
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
  DeviceGuard on_device(c->device);
  check_hip(hipMemsetAsync(c->scalars.p + SC_FAIL, 0, 3 * sizeof(double), c->stream), "memset");
  { PhaseTimer t(*c, GTG_PH_POINT_ELIM, c->phase_events.data()); launch_point_eliminate(*c, lambda, diag, dmin, dmax); }
  double g0 = 0.0, g1 = 0.0;
  int its = 0;
  { PhaseTimer t(*c, GTG_PH_CHOLESKY, c->phase_events.data());
    its = launch_pcg(*c, lambda, diag, dmin, dmax, (int)cg[0], (int)cg[1], cg[2], cg[3], &g0, &g1); }
  if (iterations) *iterations = its;
  { PhaseTimer t(*c, GTG_PH_SOLVE, c->phase_events.data());
    launch_back_substitute(*c);
    if (c->n_lm) exchange(*c, c->delta_lm.p, 3 * (int64_t)c->n_lm);   // sharded: every landmark's step from the shard that owns it
    launch_scatter_delta(*c); }
  { PhaseTimer t(*c, GTG_PH_LINEAR_ERROR, c->phase_events.data()); launch_linear_error(*c); launch_smart_lin1(*c); }
  { PhaseTimer t(*c, GTG_PH_RETRACT, c->phase_events.data()); launch_retract(*c); }
  { PhaseTimer t(*c, GTG_PH_ERROR, c->phase_events.data()); const double* gate = c->scalars.p + (smart_gate(*c) ? 2 * SC_COUNT : 0);
      launch_smart_triangulate(*c, c->trial.p, gate, false); launch_error(*c, c->trial.p, SC_TRIAL_ERROR, gate); }
  read_scalars(*c);
  collect(*c, {GTG_PH_POINT_ELIM, GTG_PH_CHOLESKY, GTG_PH_SOLVE, GTG_PH_LINEAR_ERROR, GTG_PH_RETRACT, GTG_PH_ERROR});
  c->have_trial = true;
  const double dsq = c->h_scalars[SC_DELTA_SQ];
  check_smart_supported(*c, "gtg_try_lambda_pcg");
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
  DeviceGuard on_device(c->device);
  std::swap(c->values.p, c->trial.p);
  c->linearized = false; c->have_trial = false;
  return GTG_OK;
  GTG_CATCH
}

// Replicated handles (one per GPU, each holding the whole graph -- the speculative lambda search of gtsam_amd/speculative.py): the
// device addresses of the packed values for a device-to-device exchange run by the caller on the handle's stream.
int gtg_values_device_ptr(gtg_handle c, int which, void** ptr, int64_t* n_doubles, void** stream) {
  GTG_TRY
  if (!c || !c->uploaded || !ptr || (which != 0 && which != 1)) throw std::invalid_argument("gtg_values_device_ptr: no problem uploaded / bad arguments");
  if (which == 1 && !c->have_trial) throw std::invalid_argument("gtg_values_device_ptr: no trial values (call gtg_try_lambda)");
  *ptr = which == 0 ? (void*)c->values.p : (void*)c->trial.p;
  if (n_doubles) *n_doubles = c->user_val_size;
  if (stream) *stream = (void*)c->stream;
  return GTG_OK;
  GTG_CATCH
}
int gtg_values_changed(gtg_handle c) {
  GTG_TRY
  if (!c || !c->uploaded) throw std::invalid_argument("gtg_values_changed: no problem uploaded");
  DeviceGuard on_device(c->device);
  check_hip(hipStreamSynchronize(c->stream), "sync");
  c->linearized = false; c->have_trial = false;
  return GTG_OK;
  GTG_CATCH
}

int gtg_get_gradient(gtg_handle c, double* g, int64_t n) {
  GTG_TRY
  if (!c || !c->linearized || n != c->user_dim_size) throw std::invalid_argument("gtg_get_gradient: linearize first / wrong size");
  DeviceGuard on_device(c->device);
  std::vector<double> gr(9 * (size_t)std::max(c->n_red_vars, 1)), gp(3 * (size_t)std::max(c->n_lm, 1));
  check_hip(hipMemcpy(gr.data(), c->gred0.p, sizeof(double) * gr.size(), hipMemcpyDeviceToHost), "D2H");
  check_hip(hipMemcpy(gp.data(), c->gp.p, sizeof(double) * gp.size(), hipMemcpyDeviceToHost), "D2H");
  for (int v = 0; v < c->n_user_vars; v++) {   // (the hidden landmarks of smart factors are not the caller's variables)
    double* d = g + c->h_dim_off[v];
    if (c->h_lm_index[v] >= 0) for (int k = 0; k < 3; k++) d[k] = gp[3 * c->h_lm_index[v] + k];
    else for (int k = 0; k < c->h_red_dim[c->h_red_index[v]]; k++) d[k] = gr[9 * c->h_red_index[v] + k];
  }
  return GTG_OK;
  GTG_CATCH
}

int gtg_get_hessian_diagonal(gtg_handle c, double* out, int64_t n) {
  GTG_TRY
  if (!c || !c->linearized || n != c->user_dim_size) throw std::invalid_argument("gtg_get_hessian_diagonal: linearize first / wrong size");
  DeviceGuard on_device(c->device);
  std::vector<double> hd(c->NP), V(9 * (size_t)std::max(c->n_lm, 1));
  check_hip(hipMemcpy(hd.data(), c->hdiag_red.p, sizeof(double) * hd.size(), hipMemcpyDeviceToHost), "D2H");
  check_hip(hipMemcpy(V.data(), c->V.p, sizeof(double) * V.size(), hipMemcpyDeviceToHost), "D2H");
  for (int v = 0; v < c->n_user_vars; v++) {
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
  DeviceGuard on_device(c->device);
  const double* src; int64_t cnt;
  switch (type) {
    case GTG_FAC_GENERAL_SFM: src = c->f.sfm_J.p; cnt = (c->n_smart ? c->smart_obs0 : c->f.n_sfm) * kSfmRec; break;   // (not the observations of smart factors)
    case GTG_FAC_PROJECTION: src = c->f.proj_J.p; cnt = c->f.n_proj * kProjRec; break;
    case GTG_FAC_BETWEEN_POSE3: src = c->f.between_J.p; cnt = c->f.n_between * kBetweenRec; break;
    case GTG_FAC_PRIOR: src = c->f.prior_J.p; cnt = c->f.n_prior * kPriorRec; break;
    default: throw std::invalid_argument("unknown factor type");
  }
  if (n != cnt) throw std::invalid_argument("gtg_get_jacobians: wrong output size");
  DevBuf<double> recomputed;
  struct Release { DevBuf<double>& b; ~Release() { b.free(); } } release{recomputed};   // (declared before the allocation: a throwing launch / sync frees it too)
  if (type == GTG_FAC_GENERAL_SFM && c->fused_sfm && cnt) {   // debug path: these records are not stored (fused.h) -- recomputed at the current values
    recomputed.alloc((size_t)cnt);
    launch_sfm_records(*c, recomputed.p);
    check_hip(hipStreamSynchronize(c->stream), "sync");
    src = recomputed.p;
  }
  if (cnt) check_hip(hipMemcpy(out, src, sizeof(double) * cnt, hipMemcpyDeviceToHost), "D2H");
  return GTG_OK;
  GTG_CATCH
}

int gtg_get_reduced_matrix(gtg_handle c, double* S, int64_t n_elems) {
  GTG_TRY
  if (!c || !c->uploaded || n_elems != c->n_red * c->n_red) throw std::invalid_argument("gtg_get_reduced_matrix: wrong size");
  DeviceGuard on_device(c->device);
  // S carries alignment gaps (identity rows) between the nested-dissection parts: copy the square part and compact it
  const int64_t n = c->n_red, NP = c->NP;
  // (the stored tiles are brought over and scattered into a dense array on the host; tiles without a slot are zero)
  std::vector<double> full((size_t)NP * NP, 0.0);
  {
    std::vector<double> tiles(c->S.n);
    check_hip(hipMemcpy(tiles.data(), c->S.p, sizeof(double) * tiles.size(), hipMemcpyDeviceToHost), "D2H");
    const int nt = (int)(NP / kTile);
    for (int I = 0; I < nt; I++)
      for (int J = 0; J < nt; J++) {
        const int32_t q = c->plan.h_slot[(size_t)I * nt + J];
        if (q < 0) continue;
        for (int r = 0; r < kTile; r++)
          std::memcpy(&full[((size_t)I * kTile + r) * NP + (size_t)J * kTile], &tiles[(size_t)q * kTileDoubles + (size_t)r * kTile], sizeof(double) * kTile);
      }
  }
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
double gtg_cholesky_flops_executed(gtg_handle c) { return !c ? 0.0 : (c->use_df && c->df.flops_executed > 0.0) ? c->df.flops_executed : c->chol_flops; }
double gtg_cholesky_flops_block_level(gtg_handle c) {
  if (!c) return 0.0;
  try { join_block_level(*c); } catch (...) { return 0.0; }
  return c->chol_flops_block;
}
int64_t gtg_structure_hash(gtg_handle c) { return c ? (int64_t)(c->structure_hash & 0x7FFFFFFFFFFFFFFFull) : -1; }
double gtg_linearize_bytes(gtg_handle c) { return c ? c->lin_bytes : 0.0; }

// debug only (not in the public header): ms per K=256 trailing update over an m x m tile grid, with ablations
double gtg_debug_syrk_ms(gtg_handle c, int m, int abl, int reps) {
  try {
    DeviceGuard on_device(c->device);
    const int nt = m + 2;      // every tile of an nt x nt grid gets a slot
    DevBuf<double> S; S.alloc((size_t)nt * nt * kTileDoubles);
    std::vector<int32_t> hs((size_t)(nt + 1) * nt, -1);
    for (int q = 0; q < nt * nt; q++) hs[(size_t)q] = q;
    DevBuf<int32_t> slot; slot.upload(hs.data(), hs.size(), c->stream);
    check_hip(hipMemset(S.p, 0, sizeof(double) * S.n), "memset");
    const double ms = debug_time_syrk(*c, SMat{S.p, slot.p, nt}, m, abl, reps);
    S.free(); slot.free();
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
  (void)hipSetDevice(c->device);
  ensure_stream_lists(c->plan, c->stream);
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
  DeviceGuard on_device(c->device);
  (void)hipSetDevice(c->device);
  ensure_stream_lists(c->plan, c->stream);
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

int gtg_debug_df_plan(gtg_handle c, int64_t sizes[4], int32_t* tasks, int32_t* klist) {
  GTG_TRY
  if (!c || !c->uploaded || !sizes) throw std::invalid_argument("gtg_debug_df_plan: no problem uploaded");
  const DfPlan& df = c->df;
  sizes[0] = df.nt; sizes[1] = df.n_tasks; sizes[2] = (int64_t)df.h_klist.size(); sizes[3] = c->use_df ? 1 : 0;
  if (tasks) std::copy(df.h_tasks.begin(), df.h_tasks.end(), tasks);
  if (klist) std::copy(df.h_klist.begin(), df.h_klist.end(), klist);
  return GTG_OK;
  GTG_CATCH
}

int gtg_debug_df_device_tables(gtg_handle c, int64_t sizes[3], int32_t* tasks, int32_t* steps, int32_t* chain) {
  GTG_TRY
  if (!c || !c->uploaded || !sizes) throw std::invalid_argument("gtg_debug_df_device_tables: no problem uploaded");
  const DfPlan& df = c->df;
  sizes[0] = (int64_t)df.tasks.n; sizes[1] = (int64_t)df.klist.n; sizes[2] = (int64_t)df.has_sub.n;
  (void)hipSetDevice(c->device);
  if (tasks && df.tasks.n) check_hip(hipMemcpyAsync(tasks, df.tasks.p, sizeof(int32_t) * df.tasks.n, hipMemcpyDeviceToHost, c->stream), "D2H");
  if (steps && df.klist.n) check_hip(hipMemcpyAsync(steps, df.klist.p, sizeof(int32_t) * df.klist.n, hipMemcpyDeviceToHost, c->stream), "D2H");
  if (chain && df.has_sub.n) check_hip(hipMemcpyAsync(chain, df.has_sub.p, sizeof(int32_t) * df.has_sub.n, hipMemcpyDeviceToHost, c->stream), "D2H");
  check_hip(hipStreamSynchronize(c->stream), "sync");
  return GTG_OK;
  GTG_CATCH
}

int gtg_debug_df_chains(gtg_handle c, int64_t sizes[3], int32_t* chain_off, int32_t* chain_tiles, int32_t* seq) {
  GTG_TRY
  if (!c || !c->uploaded || !sizes) throw std::invalid_argument("gtg_debug_df_chains: no problem uploaded");
  const DfPlan& df = c->df;
  sizes[0] = df.n_chain; sizes[1] = (int64_t)df.h_chain_tiles.size(); sizes[2] = (int64_t)df.h_seq.size();
  if (chain_off) std::copy(df.h_chain_off.begin(), df.h_chain_off.end(), chain_off);
  if (chain_tiles) std::copy(df.h_chain_tiles.begin(), df.h_chain_tiles.end(), chain_tiles);
  if (seq) std::copy(df.h_seq.begin(), df.h_seq.end(), seq);
  return GTG_OK;
  GTG_CATCH
}

int gtg_debug_reduced_order(gtg_handle c, int32_t* var_of_position, int32_t n) {
  GTG_TRY
  if (!c || !c->uploaded || !var_of_position || n != c->n_red_vars) throw std::invalid_argument("gtg_debug_reduced_order: no problem uploaded or wrong size");
  for (int r = 0; r < c->n_red_vars; r++) var_of_position[c->h_red_pos[r]] = c->h_red_var[r];
  return GTG_OK;
  GTG_CATCH
}

int gtg_debug_df_trace(gtg_handle c, int64_t* out, int64_t n) {
  GTG_TRY
  if (!c || !c->uploaded || !out || !c->df.trace.p || n != (int64_t)c->df.trace.n) throw std::invalid_argument("gtg_debug_df_trace: no trace (GTG_DF_TRACE=1 at upload) or wrong size");
  DeviceGuard on_device(c->device);
  check_hip(hipMemcpy(out, c->df.trace.p, sizeof(long long) * n, hipMemcpyDeviceToHost), "D2H");
  return GTG_OK;
  GTG_CATCH
}

int gtg_debug_df_ctrl(gtg_handle c, int32_t out[16]) {
  GTG_TRY
  if (!c || !c->uploaded || !out || !c->df.ctrl.p) throw std::invalid_argument("gtg_debug_df_ctrl: no dataflow schedule");
  DeviceGuard on_device(c->device);
  check_hip(hipMemcpy(out, c->df.ctrl.p, sizeof(int32_t) * 16, hipMemcpyDeviceToHost), "D2H");
  out[15] = (int32_t)c->df_fallbacks;   // (host counter) lambda tries repeated with the stream schedule after a time-out
  return GTG_OK;
  GTG_CATCH
}

int gtg_debug_df_poll_stats(gtg_handle c, int64_t out[5]) {
  GTG_TRY
  if (!c || !c->uploaded || !out || !c->df.ctrl.p) throw std::invalid_argument("gtg_debug_df_poll_stats: no dataflow schedule");
  DeviceGuard on_device(c->device);
  int32_t w[32];
  check_hip(hipMemcpy(w, c->df.ctrl.p, sizeof w, hipMemcpyDeviceToHost), "D2H");
  out[0] = w[18]; out[1] = w[7]; out[2] = w[16]; out[3] = w[6]; out[4] = w[17];
  return GTG_OK;
  GTG_CATCH
}

int gtg_dense_cholesky_host(gtg_handle c, double* A, int32_t n, double* rhs) {
  GTG_TRY
  if (!c || !A || n < 1) throw std::invalid_argument("gtg_dense_cholesky_host: bad arguments");
  DeviceGuard on_device(c->device);
  const int NP = (n + kTile - 1) / kTile * kTile;
  const int nt = NP / kTile;
  CholPlan plan;
  build_chol_plan(plan, nt, nullptr, c->stream);   // dense: every lower tile + the rhs row has a slot
  DevBuf<double> S, Dinv, x, fail;
  DfPlan df;
  const bool use_df = dataflow_schedule_selected();
  if (use_df) build_df_plan(df, nt, nullptr, c->stream, plan.h_slot, plan.n_stored);   // (before S: the plan may want scratch slots behind the tiles)
  S.alloc((size_t)(plan.n_stored + df.n_scratch) * kTileDoubles); Dinv.alloc((size_t)nt * kTile * kTile); x.alloc(2 * (size_t)NP); fail.alloc(2);
  check_hip(hipMemset(Dinv.p, 0, sizeof(double) * Dinv.n), "memset");
  if (!c->chol_epoch_dev.p) { c->chol_epoch_dev.alloc(1); check_hip(hipMemset(c->chol_epoch_dev.p, 0, sizeof(long long)), "memset"); }
  check_hip(hipMemsetAsync(fail.p, 0, 2 * sizeof(double), c->stream), "memset");
  // the matrix by tiles (host side): lower triangle of A, identity on the padding, rhs in row 0 of the rhs tiles
  std::vector<double> tiles((size_t)plan.n_stored * kTileDoubles, 0.0);
  const SMat hs{tiles.data(), plan.h_slot.data(), nt};
  for (int64_t i = 0; i < n; i++) for (int64_t j = 0; j <= i; j++) *hs.at(i, j) = A[i * n + j];
  for (int64_t i = 0; i < n; i++) for (int64_t j = i + 1; j < std::min<int64_t>(n, (i / kTile + 1) * kTile); j++) *hs.at(i, j) = A[i * n + j];   // (upper part of the diagonal tiles, as the dense copy had it)
  for (int64_t i = n; i < NP; i++) *hs.at(i, i) = 1.0;
  if (rhs) for (int64_t j = 0; j < n; j++) *hs.at(NP, j) = rhs[j];
  check_hip(hipMemcpyAsync(S.p, tiles.data(), sizeof(double) * tiles.size(), hipMemcpyHostToDevice, c->stream), "H2D");
  const SMat Sm{S.p, plan.slot.p, nt};
  // rank test: the matrix is ONE frontal block, as in choleskyPartial(ABC, nFrontal = n) (base/cholesky.cpp:144-157)
  std::vector<unsigned char> pk(NP, 0);
  pk[n - 1] = n >= 2 ? 1 : 2;
  DevBuf<unsigned char> dpk; dpk.upload(pk.data(), pk.size(), c->stream);
  DevBuf<double> dexp; dexp.alloc(NP / kTile + 1);
  std::unique_lock<std::mutex> one_at_a_time;
  if (use_df) one_at_a_time = std::unique_lock<std::mutex>(df_device_lock(c->device));
  if (use_df) launch_cholesky_df(*c, Sm, NP, df, Dinv.p, fail.p, dpk.p, dexp.p);
  else launch_cholesky(*c, Sm, NP, plan, Dinv.p, fail.p, dpk.p, dexp.p);
  if (rhs) launch_backward_solve(*c, Sm, NP, plan, Dinv.p, x.p, fail.p);
  double hf2[2] = {0, 0};
  check_hip(hipMemcpyAsync(hf2, fail.p, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream), "D2H");
  check_hip(hipMemcpyAsync(tiles.data(), S.p, sizeof(double) * tiles.size(), hipMemcpyDeviceToHost, c->stream), "D2H");
  if (rhs) check_hip(hipMemcpyAsync(rhs, x.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream), "D2H");
  check_hip(hipStreamSynchronize(c->stream), "sync");
  for (int64_t i = 0; i < n; i++)       // the factor: lower triangle (and what the diagonal tiles hold above it, as before)
    for (int64_t j = 0; j < std::min<int64_t>(n, (i / kTile + 1) * kTile); j++) A[i * n + j] = *hs.at(i, j);
  dpk.free(); dexp.free();
  S.free(); Dinv.free(); x.free(); fail.free(); plan.rows.free(); plan.pairs.free(); plan.bcols.free(); plan.stored.free(); plan.slot.free(); plan.bwd_col_off.free(); plan.bwd_col_rows.free();
  free_df_plan(df);
  if (hf2[1] != 0.0) throw std::runtime_error("gtg_dense_cholesky_host: a dependency wait of the factorisation ran into its bound");
  return hf2[0] != 0.0 ? GTG_INDETERMINATE : GTG_OK;
  GTG_CATCH
}

int64_t gtg_release_cached_memory(void) { destroy_parked(); return (int64_t)release_kept(-1); }
int64_t gtg_cached_memory_bytes(void) { std::lock_guard<std::mutex> lk(g_kept_mu); size_t b = 0; for (const auto& k : g_kept) b += k.bytes; return (int64_t)b; }

}  // extern "C"
