// tools/graph_probe.hip -- stand-alone probe for DESIGN.md section 8, item 1: does an EXPLICITLY constructed hipGraph
// (hipGraphAddKernelNode + dependencies; stream capture segfaults in this ROCm build) work, and what does a dependent
// kernel cost inside it compared with the stream / event issue the library uses today?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/graph_probe.hip -o /tmp/graph_probe && /tmp/graph_probe [pairs=61]
//
// The DAG is the shape of the tile-Cholesky schedule of one factorisation, with spin kernels of the measured durations
// standing in for the real ones (one workgroup each: only dispatch and dependency latency is measured):
//   chain, per column pair p:  A_p (34 us) -> T_p (9 us) -> B_p (34 us) -> N_p (13 us) -> A_{p+1} ...      (high-priority stream)
//   bulk update:               R_p (85 us) after N_p and R_{p-1};  N_p also waits for R_{p-2}               (main stream)
// Reported: the sum of the chain's kernel durations (the floor), the stream / event schedule, the explicit graph.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)

__global__ void k_spin(long long ticks, int* sink) {   // wall_clock64 ticks at 100 MHz on gfx9: 100 ticks = 1 us
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && ticks < 0) *sink = 1;
}

struct Node { long long ticks; std::vector<int> deps; int stream; };   // stream 0 = chain, 1 = bulk

int main(int argc, char** argv) {
  const int P = argc > 1 ? std::atoi(argv[1]) : 61;
  const long long us = 100;   // ticks per microsecond
  std::vector<Node> dag;
  std::vector<int> N(P, -1), R(P, -1);
  int prev = -1;
  double chain_us = 0;
  for (int p = 0; p < P; p++) {
    auto add = [&](long long t, std::vector<int> deps, int stream) { dag.push_back(Node{t * us, deps, stream}); return (int)dag.size() - 1; };
    const int A = add(34, prev >= 0 ? std::vector<int>{prev} : std::vector<int>{}, 0);
    const int T = add(9, {A}, 0);
    const int B = add(34, {T}, 0);
    std::vector<int> nd{B}; if (p >= 2) nd.push_back(R[p - 2]);
    N[p] = add(13, nd, 0);
    std::vector<int> rd{N[p]}; if (p >= 1) rd.push_back(R[p - 1]);
    R[p] = add(85, rd, 1);
    prev = N[p]; chain_us += 34 + 9 + 34 + 13;
  }
  int lo = 0, hi = 0; CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s[2]; CHECK(hipStreamCreateWithPriority(&s[0], hipStreamNonBlocking, hi)); CHECK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  hipEvent_t t0, t1; CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
  std::vector<hipEvent_t> ev(dag.size());
  for (auto& e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  // ---- today's way: streams + events -----------------------------------------------------------------------------------
  float best_stream = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(t0, s[1])); CHECK(hipStreamWaitEvent(s[0], t0, 0));
    for (size_t i = 0; i < dag.size(); i++) {
      for (int d : dag[i].deps) if (dag[d].stream != dag[i].stream) CHECK(hipStreamWaitEvent(s[dag[i].stream], ev[d], 0));
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[dag[i].stream], dag[i].ticks, (int*)nullptr);
      CHECK(hipEventRecord(ev[i], s[dag[i].stream]));
    }
    CHECK(hipStreamWaitEvent(s[1], ev[N[P - 1]], 0));
    CHECK(hipEventRecord(t1, s[1])); CHECK(hipEventSynchronize(t1));
    float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best_stream) best_stream = ms;
  }
  // ---- explicit graph --------------------------------------------------------------------------------------------------
  hipGraph_t graph; CHECK(hipGraphCreate(&graph, 0));
  std::vector<hipGraphNode_t> gn(dag.size());
  std::vector<long long> ticks(dag.size()); int* sink = nullptr;
  std::vector<void*> argp(2 * dag.size());
  for (size_t i = 0; i < dag.size(); i++) {
    ticks[i] = dag[i].ticks; argp[2 * i] = &ticks[i]; argp[2 * i + 1] = &sink;
    hipKernelNodeParams kp = {};
    kp.func = (void*)k_spin; kp.gridDim = dim3(1); kp.blockDim = dim3(64); kp.sharedMemBytes = 0; kp.kernelParams = &argp[2 * i]; kp.extra = nullptr;
    std::vector<hipGraphNode_t> deps; for (int d : dag[i].deps) deps.push_back(gn[d]);
    CHECK(hipGraphAddKernelNode(&gn[i], graph, deps.data(), deps.size(), &kp));
  }
  hipGraphExec_t exec; CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  float best_graph = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(t0, s[1])); CHECK(hipGraphLaunch(exec, s[1])); CHECK(hipEventRecord(t1, s[1])); CHECK(hipEventSynchronize(t1));
    float ms; CHECK(hipEventElapsedTime(&ms, t0, t1)); if (ms < best_graph) best_graph = ms;
  }
  std::printf("{\"pairs\": %d, \"nodes\": %zu, \"chain_kernel_sum_ms\": %.3f, \"stream_event_schedule_ms\": %.3f, \"explicit_graph_ms\": %.3f}\n",
              P, dag.size(), chain_us * 1e-3, best_stream, best_graph);
  return 0;
}
