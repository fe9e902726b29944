// primitives.hip -- exclusive scan, stable LSD radix sort and run detection for the set-up passes, hand-written for gfx950 (64-wide
// wavefronts: ballots are 64-bit masks, wave scans are six shuffle steps).  See primitives.h for the contracts.
//
//   scan    three launches: tile sums (256 lanes x 8 items), one workgroup scans the tile sums in place, every tile scans itself from
//           its offset.  The arrays here are 0.15 - 6 M elements: two passes over the data at HBM rate, no look-back chain.
//   sort    per 8-bit digit: k_rs_hist (a 256-bin histogram per tile of 4096 keys, LDS atomics, written bin-major so that ONE scan of
//           the whole table yields the global offset of every (bin, tile)), the scan, k_rs_scatter.  Stability inside a tile: the tile is
//           walked in 16 rounds of 256 keys in index order; in a round every wavefront finds, with 8 ballots, the lanes that hold the same
//           digit (rank of a lane among them = popcount of the lower lanes), the round's per-wavefront digit counts go through LDS, and a
//           running per-digit count carries from round to round: the destination of a key is its digit's global offset + the keys of that
//           digit before it in the tile.
//   runs    head flags (key differs from its predecessor), scan, compaction: distinct keys, first positions.
#include "primitives.h"

#include <algorithm>
#include <stdexcept>

#include "kernels.h"

namespace gt {
namespace prim {

namespace {

constexpr int kScanItems = 8, kScanTile = 256 * kScanItems;
constexpr int kSortRounds = 16, kSortTile = 256 * kSortRounds;

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

__device__ __forceinline__ long long wave_inclusive(long long x, int lane) {
  for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
  return x;
}

template <class In>
__global__ __launch_bounds__(256) void k_scan_sums(const In* __restrict__ in, size_t n, long long* __restrict__ sums) {
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  long long t = 0;
  for (int j = 0; j < kScanItems; j++) if (base + j < n) t += (long long)in[base + j];
  for (int o = 32; o; o >>= 1) t += __shfl_down(t, o, 64);
  __shared__ long long w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// one workgroup: sums[0 .. nb) -> their exclusive prefix sums, in place
__global__ __launch_bounds__(1024) void k_scan_blocks(long long* __restrict__ sums, int nb) {
  __shared__ long long wsum[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long carry = 0;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const long long v = i < nb ? sums[i] : 0;
    const long long x = wave_inclusive(v, lane);
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    long long woff = 0, total = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) woff += wsum[w]; total += wsum[w]; }
    if (i < nb) sums[i] = carry + woff + x - v;
    carry += total;
    __syncthreads();
  }
}

template <class In>
__global__ __launch_bounds__(256) void k_scan_apply(const In* __restrict__ in, long long* __restrict__ out, size_t n, const long long* __restrict__ tile_off) {
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long v[kScanItems], t = 0;
  for (int j = 0; j < kScanItems; j++) { v[j] = base + j < n ? (long long)in[base + j] : 0; t += v[j]; }
  const long long x = wave_inclusive(t, lane);
  __shared__ long long wsum[4];
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  long long off = tile_off[blockIdx.x] + (x - t);
  for (int w = 0; w < wave; w++) off += wsum[w];
  for (int j = 0; j < kScanItems; j++) if (base + j < n) { out[base + j] = off; off += v[j]; }
}

template <class In>
void scan_impl(const In* in, int64_t* out, size_t n, void* scratch, hipStream_t s) {
  if (n == 0) return;
  const size_t nb = (n + kScanTile - 1) / kScanTile;
  if (nb >= ((size_t)1 << 31)) throw std::runtime_error("exclusive_scan: array too long");
  long long* sums = static_cast<long long*>(scratch);
  hipLaunchKernelGGL(k_scan_sums<In>, dim3((unsigned)nb), dim3(256), 0, s, in, n, sums);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, sums, (int)nb);
  hipLaunchKernelGGL(k_scan_apply<In>, dim3((unsigned)nb), dim3(256), 0, s, in, reinterpret_cast<long long*>(out), n, sums);
}

// ---- radix sort ------------------------------------------------------------------------------------------------------------------
template <class Key>
__global__ __launch_bounds__(256) void k_rs_hist(const Key* __restrict__ key, size_t n, int shift, int mask, int ntiles, int32_t* __restrict__ hist) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kSortTile;
  for (int r = 0; r < kSortRounds; r++) {
    const size_t i = base + (size_t)r * 256 + threadIdx.x;
    if (i < n) atomicAdd(&h[(int)((key[i] >> shift) & (Key)mask)], 1);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];     // bin-major: the scan of the table is the global offset of (bin, tile)
}

template <class Key, bool kValues>
__global__ __launch_bounds__(256) void k_rs_scatter(const Key* __restrict__ key, const uint32_t* __restrict__ val, Key* __restrict__ key_out,
                                                    uint32_t* __restrict__ val_out, size_t n, int shift, int mask, int ntiles,
                                                    const long long* __restrict__ offset) {
  __shared__ long long goff[256];
  __shared__ int running[256];
  __shared__ int wcount[4][256];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  goff[t] = offset[(size_t)t * ntiles + blockIdx.x];
  running[t] = 0;
  for (int w = 0; w < 4; w++) wcount[w][t] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kSortTile;
  for (int r = 0; r < kSortRounds; r++) {
    const size_t i = base + (size_t)r * 256 + t;
    const bool valid = i < n;
    const Key k = valid ? key[i] : (Key)0;
    const int d = (int)((k >> shift) & (Key)mask);
    // the lanes of this wavefront that hold the same digit (valid lanes only)
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1;
      const unsigned long long m = __ballot(valid && bit);
      peers &= bit ? m : ~m;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank == 0) wcount[wave][d] = __popcll(peers);
    __syncthreads();
    if (valid) {
      int before = running[d] + rank;
      for (int w = 0; w < wave; w++) before += wcount[w][d];
      const long long dst = goff[d] + before;
      key_out[dst] = k;
      if (kValues) val_out[dst] = val[i];
    }
    __syncthreads();
    running[t] += wcount[0][t] + wcount[1][t] + wcount[2][t] + wcount[3][t];
    for (int w = 0; w < 4; w++) wcount[w][t] = 0;
    __syncthreads();
  }
}

template <class Key, bool kValues>
void sort_impl(Key* key_in, Key* key_out, uint32_t* val_in, uint32_t* val_out, size_t n, int bits, void* scratch, hipStream_t s) {
  if (n == 0) return;
  bits = std::max(1, std::min(bits, (int)(8 * sizeof(Key))));
  const size_t ntiles = (n + kSortTile - 1) / kSortTile;
  if (ntiles * 256 >= ((size_t)1 << 31)) throw std::runtime_error("radix sort: array too long");
  char* p = static_cast<char*>(scratch);
  int32_t* hist = reinterpret_cast<int32_t*>(p); p += al256(4 * 256 * ntiles);
  long long* offset = reinterpret_cast<long long*>(p); p += al256(8 * 256 * ntiles);
  void* scan_scratch = p;
  const int passes = (bits + 7) / 8;
  Key* src = key_in; Key* dst = key_out; uint32_t* vsrc = val_in; uint32_t* vdst = val_out;
  int shift = 0;
  for (int pass = 0; pass < passes; pass++) {
    const int width = (bits - shift + (passes - pass) - 1) / (passes - pass);     // the remaining bits, evenly over the remaining passes
    const int mask = (1 << width) - 1;
    hipLaunchKernelGGL((k_rs_hist<Key>), dim3((unsigned)ntiles), dim3(256), 0, s, src, n, shift, mask, (int)ntiles, hist);
    scan_impl<int32_t>(hist, reinterpret_cast<int64_t*>(offset), 256 * ntiles, scan_scratch, s);
    hipLaunchKernelGGL((k_rs_scatter<Key, kValues>), dim3((unsigned)ntiles), dim3(256), 0, s, src, vsrc, dst, vdst, n, shift, mask, (int)ntiles, offset);
    std::swap(src, dst); std::swap(vsrc, vdst);
    shift += width;
  }
  if (src != key_out) {     // an even number of passes ends in the input buffers
    check_hip(hipMemcpyAsync(key_out, src, sizeof(Key) * n, hipMemcpyDeviceToDevice, s), "D2D");
    if (kValues) check_hip(hipMemcpyAsync(val_out, vsrc, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, s), "D2D");
  }
}

// ---- runs ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_run_heads(const uint64_t* __restrict__ k, size_t n, int32_t* __restrict__ head) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) head[i] = (i == 0 || k[i] != k[i - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_run_compact(const uint64_t* __restrict__ k, size_t n, const int32_t* __restrict__ head, const long long* __restrict__ pos,
                                                     uint64_t* __restrict__ uniq, long long* __restrict__ start, int32_t* __restrict__ n_runs) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (head[i]) { uniq[pos[i]] = k[i]; if (start) start[pos[i]] = (long long)i; }
  if (i == n - 1) { const long long nr = pos[i] + head[i]; *n_runs = (int32_t)nr; if (start) start[nr] = (long long)n; }
}
__global__ void k_run_none(long long* start, int32_t* n_runs) { *n_runs = 0; if (start) start[0] = 0; }

}  // namespace

size_t scan_scratch_bytes(size_t n) { return al256(8 * ((n + kScanTile - 1) / kScanTile + 1)); }
void exclusive_scan(const int64_t* in, int64_t* out, size_t n, void* scratch, hipStream_t s) { scan_impl<int64_t>(in, out, n, scratch, s); }
void exclusive_scan(const int32_t* in, int64_t* out, size_t n, void* scratch, hipStream_t s) { scan_impl<int32_t>(in, out, n, scratch, s); }

size_t sort_scratch_bytes(size_t n) {
  const size_t ntiles = (n + kSortTile - 1) / kSortTile + 1;
  return al256(4 * 256 * ntiles) + al256(8 * 256 * ntiles) + scan_scratch_bytes(256 * ntiles);
}
void sort_pairs(uint64_t* key_in, uint64_t* key_out, uint32_t* val_in, uint32_t* val_out, size_t n, int bits, void* scratch, hipStream_t s) {
  sort_impl<uint64_t, true>(key_in, key_out, val_in, val_out, n, bits, scratch, s);
}
void sort_pairs(uint32_t* key_in, uint32_t* key_out, uint32_t* val_in, uint32_t* val_out, size_t n, int bits, void* scratch, hipStream_t s) {
  sort_impl<uint32_t, true>(key_in, key_out, val_in, val_out, n, bits, scratch, s);
}
void sort_keys(uint64_t* key_in, uint64_t* key_out, size_t n, int bits, void* scratch, hipStream_t s) {
  sort_impl<uint64_t, false>(key_in, key_out, nullptr, nullptr, n, bits, scratch, s);
}

size_t runs_scratch_bytes(size_t n) { return al256(4 * (n + 1)) + al256(8 * (n + 1)) + scan_scratch_bytes(n + 1); }
void runs(const uint64_t* sorted, size_t n, uint64_t* uniq, int64_t* start, int32_t* n_runs, void* scratch, hipStream_t s) {
  if (n == 0) { hipLaunchKernelGGL(k_run_none, dim3(1), dim3(1), 0, s, reinterpret_cast<long long*>(start), n_runs); return; }
  char* p = static_cast<char*>(scratch);
  int32_t* head = reinterpret_cast<int32_t*>(p); p += al256(4 * (n + 1));
  long long* pos = reinterpret_cast<long long*>(p); p += al256(8 * (n + 1));
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_run_heads, dim3(grid), dim3(256), 0, s, sorted, n, head);
  scan_impl<int32_t>(head, reinterpret_cast<int64_t*>(pos), n, p, s);
  hipLaunchKernelGGL(k_run_compact, dim3(grid), dim3(256), 0, s, sorted, n, head, pos, uniq, reinterpret_cast<long long*>(start), n_runs);
}

// gtg_prewarm: this unit's kernels (kernels.h)
static void prewarm_primitives(int) {
  prewarm_kernels({(const void*)k_scan_sums<int32_t>, (const void*)k_scan_sums<int64_t>, (const void*)k_scan_blocks, (const void*)k_scan_apply<int32_t>,
                   (const void*)k_scan_apply<int64_t>, (const void*)k_rs_hist<uint32_t>, (const void*)k_rs_hist<uint64_t>, (const void*)k_rs_scatter<uint32_t, true>,
                   (const void*)k_rs_scatter<uint64_t, true>, (const void*)k_rs_scatter<uint64_t, false>, (const void*)k_run_heads, (const void*)k_run_compact});
}
static PrewarmUnit prewarm_primitives_registered(prewarm_primitives);

}  // namespace prim
}  // namespace gt

// ---- tests: the primitives on host arrays (tests/test_gpu_primitives.py compares them with numpy) ------------------------------------------
namespace {
template <class T> struct Dev {
  T* p = nullptr;
  explicit Dev(size_t n) { gt::check_hip(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc"); }
  ~Dev() { (void)hipFree(p); }
};
template <class F> int guarded(int device, F f) {
  try { gt::check_hip(hipSetDevice(device), "hipSetDevice"); f(); gt::check_hip(hipDeviceSynchronize(), "sync"); return 0; } catch (...) { return -2; }
}
}  // namespace

extern "C" {
int gtg_debug_scan(int device, const int64_t* in, int64_t n, int64_t* out) {
  return guarded(device, [&] {
    Dev<int64_t> a((size_t)n), b((size_t)n); Dev<unsigned char> tmp(gt::prim::scan_scratch_bytes((size_t)n));
    gt::check_hip(hipMemcpy(a.p, in, 8 * (size_t)n, hipMemcpyHostToDevice), "H2D");
    gt::prim::exclusive_scan(a.p, b.p, (size_t)n, tmp.p, nullptr);
    gt::check_hip(hipMemcpy(out, b.p, 8 * (size_t)n, hipMemcpyDeviceToHost), "D2H");
  });
}
int gtg_debug_sort_pairs(int device, int key_bytes, const void* key, const uint32_t* val, int64_t n, int bits, void* key_out, uint32_t* val_out) {
  return guarded(device, [&] {
    const size_t N = (size_t)n, kb = (size_t)key_bytes;
    Dev<unsigned char> k1(kb * N), k2(kb * N), tmp(gt::prim::sort_scratch_bytes(N)); Dev<uint32_t> v1(N), v2(N);
    gt::check_hip(hipMemcpy(k1.p, key, kb * N, hipMemcpyHostToDevice), "H2D");
    if (val) gt::check_hip(hipMemcpy(v1.p, val, 4 * N, hipMemcpyHostToDevice), "H2D");
    if (key_bytes == 8 && val) gt::prim::sort_pairs(reinterpret_cast<uint64_t*>(k1.p), reinterpret_cast<uint64_t*>(k2.p), v1.p, v2.p, N, bits, tmp.p, nullptr);
    else if (key_bytes == 8) gt::prim::sort_keys(reinterpret_cast<uint64_t*>(k1.p), reinterpret_cast<uint64_t*>(k2.p), N, bits, tmp.p, nullptr);
    else gt::prim::sort_pairs(reinterpret_cast<uint32_t*>(k1.p), reinterpret_cast<uint32_t*>(k2.p), v1.p, v2.p, N, bits, tmp.p, nullptr);
    gt::check_hip(hipMemcpy(key_out, k2.p, kb * N, hipMemcpyDeviceToHost), "D2H");
    if (val) gt::check_hip(hipMemcpy(val_out, v2.p, 4 * N, hipMemcpyDeviceToHost), "D2H");
  });
}
int gtg_debug_runs(int device, const uint64_t* sorted, int64_t n, uint64_t* uniq, int64_t* start, int32_t* n_runs) {
  return guarded(device, [&] {
    const size_t N = (size_t)n;
    Dev<uint64_t> k(N), u(N); Dev<int64_t> st(N + 1); Dev<int32_t> nr(1); Dev<unsigned char> tmp(gt::prim::runs_scratch_bytes(N));
    gt::check_hip(hipMemcpy(k.p, sorted, 8 * N, hipMemcpyHostToDevice), "H2D");
    gt::prim::runs(k.p, N, u.p, start ? st.p : nullptr, nr.p, tmp.p, nullptr);
    gt::check_hip(hipDeviceSynchronize(), "sync");
    gt::check_hip(hipMemcpy(n_runs, nr.p, 4, hipMemcpyDeviceToHost), "D2H");
    gt::check_hip(hipMemcpy(uniq, u.p, 8 * (size_t)*n_runs, hipMemcpyDeviceToHost), "D2H");
    if (start) gt::check_hip(hipMemcpy(start, st.p, 8 * ((size_t)*n_runs + 1), hipMemcpyDeviceToHost), "D2H");
  });
}
}  // extern "C"
