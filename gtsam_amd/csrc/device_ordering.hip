// device_ordering.hip -- the fill-reducing ordering of the reduced variables on the device.
//
// The reference gets its elimination order from COLAMD (inference/Ordering.cpp:42-124); here the order of the cameras / poses only
// decides where each block sits in S, and the default is reverse Cuthill-McKee on the block graph (analysis.hip: a banded pattern
// leaves most 128 x 128 tiles of the factor empty).  The host runs it as a serial queue; this file runs the SAME ordering -- position
// for position, pinned by tests/test_device_analysis_spec.py (the level-synchronous formulation against the host, CPU) and by
// tests/test_gpu_device_analysis.py (this kernel against the host, GPU) -- as data-parallel steps per BFS level:
//   adjacency      both directions of every off-diagonal block as 64-bit keys, radix sort, unique, CSR offsets by binary search
//                  (neighbours ascending, as the host's sorted lists)
//   k_rcm          ONE workgroup of 16 wavefronts walks the graph level by level:
//                    claim    every unvisited neighbour of the level is claimed by the EARLIEST node of the level it is adjacent
//                             to (atomicMin over the level's adjacency, one wavefront per level node);
//                    collect  the claiming edges emit (position of the claiming node, degree, id) keys;
//                    order    rank sort of the keys in LDS: the next level in the order in which the serial queue appends it;
//                  two sweeps of "node of smallest degree in the last level of a plain BFS" find the pseudo-peripheral start of a
//                  component, the third BFS (neighbours by degree) is its Cuthill-McKee order; components in ascending order of
//                  their first node; the concatenation reversed.
// The camera graph of the bench shapes has 1 723 nodes and 0.2 M edges: a single workgroup is the right size (a level is a few
// dozen to a few hundred nodes; the whole ordering is ~150 levels x 3 sweeps of barrier-separated steps), and nothing of the
// O(edges) work is left on the host.  Levels of more than kMaxLevel nodes, graphs of more than 4 M nodes and the nested-dissection
// orderings of the pose graphs stay with the host code (the function returns false and analysis.hip takes the host path).
#include <cstdio>

#include <climits>
#include <mutex>
#include <stdexcept>

#include "kernels.h"
#include "primitives.h"

namespace gt {

namespace {

constexpr int kRcmThreads = 1024;
constexpr int kMaxLevel = 4096;          // keys of one level in LDS (32 KB)
constexpr int kRcmLdsNodes = 8192;       // k_rcm<true>: 14 bytes of LDS per node (+ the 32 KB of level keys): 147 KB of the CU's 160 KB

__global__ __launch_bounds__(256) void k_edge_keys(int64_t m, int n, const int32_t* __restrict__ a, const int32_t* __restrict__ b, uint64_t* __restrict__ key) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= m) return;
  const uint64_t u = (uint32_t)a[i], v = (uint32_t)b[i];
  // (a block of a variable with itself is not an edge: both keys become the key of a node n, which sorts behind every node's -- k_csr's
  // offsets end in front of it)
  key[2 * i] = u == v ? (uint64_t)(uint32_t)n << 32 : (u << 32) | v;
  key[2 * i + 1] = u == v ? (uint64_t)(uint32_t)n << 32 : (v << 32) | u;
}

// ptr[u] = first key whose source is >= u (a self pair's key belongs to node n: behind ptr[n])
__global__ __launch_bounds__(256) void k_csr(int n, const uint64_t* __restrict__ key, const int32_t* __restrict__ nkeys, int32_t* __restrict__ ptr,
                                              int32_t* __restrict__ adj) {
  const int m = *nkeys;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i <= n) {
    const uint64_t want = (uint64_t)(uint32_t)i << 32;
    int lo = 0, hi = m;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] < want) lo = mid + 1; else hi = mid; }
    ptr[i] = lo;
  }
  for (int e = i; e < m; e += gridDim.x * 256) adj[e] = (int32_t)(key[e] & 0xFFFFFFFFull);
}

struct Rcm {
  int n; const int32_t* ptr; const int32_t* adj;
  int32_t* claim; unsigned char* visited; unsigned char* active;
  unsigned long long* keys;   // LDS [kMaxLevel]
  int* sh;                    // LDS scalars: [0] candidate count, [1] scratch, [2] failure
};

__device__ __forceinline__ int degree(const Rcm& g, int v) { return g.ptr[v + 1] - g.ptr[v]; }

// One BFS over the active nodes from `start`; the queue goes to q[0 ..), returns its length; *last_begin = start of the last level.
// by_degree: a level's new nodes ordered by (claiming position, degree, id), else by (claiming position, id).
__device__ int bfs(const Rcm& g, int start, bool by_degree, int32_t* __restrict__ q, int* last_begin) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < g.n; i += kRcmThreads) { g.visited[i] = 0; g.claim[i] = INT_MAX; }
  __syncthreads();
  if (tid == 0) { q[0] = start; g.visited[start] = 1; }
  int lb = 0, le = 1;
  for (;;) {
    if (tid == 0) g.sh[0] = 0;
    __syncthreads();
    // claim: one wavefront per node of the level, lanes over its neighbours
    for (int p = lb + wave; p < le; p += kRcmThreads / 64) {
      const int v = q[p];
      for (int e = g.ptr[v] + lane; e < g.ptr[v + 1]; e += 64) {
        const int w = g.adj[e];
        if (g.active[w] && !g.visited[w]) atomicMin(&g.claim[w], p);
      }
    }
    __syncthreads();
    // collect: the claiming edge of every claimed node emits its key
    for (int p = lb + wave; p < le; p += kRcmThreads / 64) {
      const int v = q[p];
      for (int e = g.ptr[v] + lane; e < g.ptr[v + 1]; e += 64) {
        const int w = g.adj[e];
        // (the claims were made by atomics, which execute in L2: read them past this CU's L1)
        if (g.active[w] && !g.visited[w] && __hip_atomic_load(&g.claim[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p) {
          const int at = atomicAdd(&g.sh[0], 1);
          if (at < kMaxLevel)
            g.keys[at] = ((unsigned long long)(unsigned)(p - lb) << 44) | ((unsigned long long)(by_degree ? (unsigned)degree(g, w) : 0u) << 22) | (unsigned)w;
        }
      }
    }
    __syncthreads();
    const int m = g.sh[0];
    if (m == 0) break;
    if (m > kMaxLevel) { if (tid == 0) g.sh[2] = 1; __syncthreads(); break; }
    // order: rank of every key among the level's keys (all distinct: the ids are)
    for (int i = tid; i < m; i += kRcmThreads) {
      const unsigned long long k = g.keys[i];
      int r = 0;
      for (int j = 0; j < m; j++) r += g.keys[j] < k;
      const int w = (int)(k & 0x3FFFFFull);
      q[le + r] = w;
      g.visited[w] = 1;
    }
    __syncthreads();
    lb = le; le += m;
  }
  *last_begin = lb;
  return le;
}

// the node the next sweep starts from: the last node of the last level unless a node of that level has a strictly smaller degree,
// then the first such node of minimum degree (analysis.hip::far_node)
__device__ int far_node(const Rcm& g, int start, int32_t* __restrict__ q) {
  int lb;
  const int le = bfs(g, start, false, q, &lb);
  const int tid = threadIdx.x;
  if (tid == 0) { g.sh[1] = INT_MAX; }
  __syncthreads();
  for (int p = lb + tid; p < le; p += kRcmThreads) atomicMin(&g.sh[1], degree(g, q[p]));
  __syncthreads();
  const int mdeg = g.sh[1];
  __syncthreads();
  if (tid == 0) g.sh[1] = INT_MAX;
  __syncthreads();
  if (degree(g, q[le - 1]) != mdeg)
    for (int p = lb + tid; p < le; p += kRcmThreads) if (degree(g, q[p]) == mdeg) atomicMin(&g.sh[1], p);
  __syncthreads();
  const int pos = g.sh[1];
  __syncthreads();
  return pos == INT_MAX ? q[le - 1] : q[pos];
}

// kLds (graphs of up to kRcmLdsNodes nodes -- every camera system of the bench shapes): the per-node state (CSR offsets, claims, the
// visited / active marks, the queue) lives in LDS instead of global memory.  A BFS level is a chain of dependent accesses -- queue ->
// offsets -> neighbour -> marks -> claim, twice -- and with all of them in global memory a level cost 4.9 us (2.2 ms for the ~450 levels
// of the three sweeps on the L1723 shape); only the neighbour lists are fetched from memory now.
template <bool kLds>
__global__ __launch_bounds__(kRcmThreads) void k_rcm(int n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ adj, int32_t* __restrict__ order,
                                                     int32_t* __restrict__ queue, int32_t* __restrict__ claim, unsigned char* __restrict__ visited,
                                                     unsigned char* __restrict__ active, int32_t* __restrict__ status) {
  __shared__ unsigned long long keys[kMaxLevel];
  __shared__ int sh[4];
  extern __shared__ __attribute__((aligned(16))) char rcm_dyn[];
  const int tid = threadIdx.x;
  if (kLds) {
    int32_t* ptr_s = reinterpret_cast<int32_t*>(rcm_dyn);
    for (int i = tid; i <= n; i += kRcmThreads) ptr_s[i] = ptr[i];
    claim = ptr_s + (n + 1); queue = claim + n;
    visited = reinterpret_cast<unsigned char*>(queue + n); active = visited + n;
    ptr = ptr_s;
  }
  Rcm g{n, ptr, adj, claim, visited, active, keys, sh};
  for (int i = tid; i < n; i += kRcmThreads) active[i] = 1;
  if (tid == 0) sh[2] = 0;
  __syncthreads();
  int done = 0, seed = 0;
  while (done < n) {
    // the component's first node: the smallest active id (>= the previous seed)
    if (tid == 0) sh[1] = INT_MAX;
    __syncthreads();
    for (int i = seed + tid; i < n; i += kRcmThreads) if (active[i]) { atomicMin(&sh[1], i); break; }
    __syncthreads();
    seed = sh[1];
    __syncthreads();
    if (seed == INT_MAX) break;
    const int start = far_node(g, far_node(g, seed, queue), queue);
    int lb;
    const int len = bfs(g, start, true, kLds ? queue : order + done, &lb);
    if (kLds) for (int p = tid; p < len; p += kRcmThreads) order[done + p] = queue[p];
    for (int p = tid; p < len; p += kRcmThreads) active[kLds ? queue[p] : order[done + p]] = 0;
    __syncthreads();
    done += len;
    if (sh[2]) break;
  }
  __syncthreads();
  // reversed
  for (int i = tid; i < n / 2; i += kRcmThreads) { const int a = order[i], b = order[n - 1 - i]; order[i] = b; order[n - 1 - i] = a; }
  if (tid == 0) status[0] = (sh[2] || done != n) ? 1 : 0;
}

void hc(hipError_t e, const char* what) { check_hip(e, what); }

}  // namespace

// edges: the off-diagonal blocks of the reduced system as pairs of reduced indices (either orientation, repeats allowed).
// Returns false when the graph is outside what the kernel handles (the caller then runs the host ordering).
bool device_rcm(gtg_context& c, int n, const std::vector<int32_t>& ea, const std::vector<int32_t>& eb, std::vector<int32_t>& order,
                const int32_t* d_ea, const int32_t* d_eb, int64_t d_edges) {
  const int64_t m = d_ea ? d_edges : (int64_t)ea.size();
  if (n < 2 || n >= (1 << 22) || m == 0 || 2 * m >= INT_MAX) return false;
  hipStream_t s = c.stream;
  // one workspace, carved (a dozen separate allocations and their releases cost more than the ordering itself)
  int bits = 33;
  while (bits < 64 && ((uint64_t)n >> (bits - 32)) != 0) bits++;
  const size_t need_sort = prim::sort_scratch_bytes((size_t)(2 * m)), need_uniq = prim::runs_scratch_bytes((size_t)(2 * m));
  size_t off = 0;
  auto carve = [&](size_t bytes) { const size_t at = off; off = (off + bytes + 255) / 256 * 256; return at; };
  const size_t o_k1 = carve(16 * (size_t)m), o_k2 = carve(16 * (size_t)m), o_tmp = carve(std::max(need_sort, need_uniq) + 16);
  const size_t o_a = carve(4 * (size_t)m), o_b = carve(4 * (size_t)m), o_ptr = carve(4 * ((size_t)n + 1)), o_adj = carve(8 * (size_t)m);
  const size_t o_order = carve(4 * (size_t)n), o_queue = carve(4 * (size_t)n), o_claim = carve(4 * (size_t)n), o_vis = carve((size_t)n), o_act = carve((size_t)n);
  const size_t o_n = carve(16), o_status = carve(16);
  // (the workspace is released on every path; a HIP error inside -- an illegal key, a failed allocation, a driver error -- makes this
  // function return false, as its contract says: the caller then runs the host ordering instead of failing the upload)
  DevBuf<unsigned char> ws;
  struct Release { DevBuf<unsigned char>& b; ~Release() { b.free(); } } release{ws};
  int32_t status = 1;
  try {
  ws.alloc(off);
  auto at = [&](size_t o) { return ws.p + o; };
  uint64_t* k1 = reinterpret_cast<uint64_t*>(at(o_k1)); uint64_t* k2 = reinterpret_cast<uint64_t*>(at(o_k2));
  int32_t* da = reinterpret_cast<int32_t*>(at(o_a)); int32_t* db = reinterpret_cast<int32_t*>(at(o_b));
  int32_t* dptr = reinterpret_cast<int32_t*>(at(o_ptr)); int32_t* dadj = reinterpret_cast<int32_t*>(at(o_adj));
  int32_t* dorder = reinterpret_cast<int32_t*>(at(o_order)); int32_t* dqueue = reinterpret_cast<int32_t*>(at(o_queue));
  int32_t* dclaim = reinterpret_cast<int32_t*>(at(o_claim)); int32_t* dn = reinterpret_cast<int32_t*>(at(o_n)); int32_t* dstatus = reinterpret_cast<int32_t*>(at(o_status));
  if (!d_ea) {
    hc(hipMemcpyAsync(da, ea.data(), 4 * (size_t)m, hipMemcpyHostToDevice, s), "H2D");
    hc(hipMemcpyAsync(db, eb.data(), 4 * (size_t)m, hipMemcpyHostToDevice, s), "H2D");
  }
  hipLaunchKernelGGL(k_edge_keys, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, m, n, d_ea ? d_ea : da, d_ea ? d_eb : db, k1);
  prim::sort_keys(k1, k2, (size_t)(2 * m), bits, at(o_tmp), s);
  prim::runs(k2, (size_t)(2 * m), k1, nullptr, dn, at(o_tmp), s);     // the distinct keys, ascending
  hipLaunchKernelGGL(k_csr, dim3((unsigned)std::max<int64_t>((n + 256) / 256, 64)), dim3(256), 0, s, n, k1, dn, dptr, dadj);
  if (n <= kRcmLdsNodes) {
    const size_t lds = 4 * ((size_t)n + 1) + 10 * (size_t)n + 16;
    static std::once_flag attr_once;     // (the attribute is per function, the largest request once)
    std::call_once(attr_once, [] { hc(hipFuncSetAttribute((const void*)k_rcm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 14 * kRcmLdsNodes + 32), "smem attr"); });
    hipLaunchKernelGGL(k_rcm<true>, dim3(1), dim3(kRcmThreads), lds, s, n, dptr, dadj, dorder, dqueue, dclaim, at(o_vis), at(o_act), dstatus);
  } else
    hipLaunchKernelGGL(k_rcm<false>, dim3(1), dim3(kRcmThreads), 0, s, n, dptr, dadj, dorder, dqueue, dclaim, at(o_vis), at(o_act), dstatus);
  hc(hipGetLastError(), "device ordering");
  order.resize((size_t)n);
  hc(hipMemcpyAsync(order.data(), dorder, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s), "D2H");
  hc(hipMemcpyAsync(&status, dstatus, sizeof(int32_t), hipMemcpyDeviceToHost, s), "D2H");
  hc(hipStreamSynchronize(s), "device ordering");
  } catch (const std::exception& e) {
    (void)hipGetLastError();
    std::fprintf(stderr, "[gtsam_amd] device ordering failed (%s): falling back to the host queue\n", e.what());
    return false;
  }
  return status == 0;
}

// gtg_prewarm: this unit's kernels (kernels.h)
static void prewarm_device_ordering(int) { prewarm_kernels({(const void*)k_edge_keys, (const void*)k_csr, (const void*)k_rcm<true>, (const void*)k_rcm<false>}); }
static PrewarmUnit prewarm_device_ordering_registered(prewarm_device_ordering);

}  // namespace gt
