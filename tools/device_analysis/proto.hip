// tools/device_analysis/proto.hip -- stand-alone prototype of the device-side symbolic analysis planned for the next round
// (DESIGN.md section 8, item 4).  NOT part of the product library: it exists so that ONE GPU call can validate the device
// formulation against the host's lists before it is wired into gtg_upload_problem.
//
//   python tools/device_analysis/make_input.py ladybug1723 /tmp/l1723.bin        (CPU: inputs + the lists the host builds)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/device_analysis/proto.hip -o /tmp/proto && /tmp/proto /tmp/l1723.bin
//
// Pipeline (tests/test_device_analysis_spec.py is the numpy statement of the same thing, pinned against the library):
//   k_count   one landmark per lane: number of terms = k (k + 1) / 2 + pairs of observations by the same camera
//   ExclusiveSum                      -> term offsets (the emission order = landmark order)
//   k_emit    one landmark per lane: key = row position * n + column position, oriented (oa, ob), mirrored duplicates
//   DeviceRadixSort::SortPairs        stable: the terms of a block keep the landmark order (= the summation order)
//   gather (oa, ob), DeviceRunLengthEncode::Encode -> unique blocks, ExclusiveSum -> pair_ptr
// Prints the time of every stage and PASS / FAIL of the bit-for-bit comparison with the expected lists.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)

__global__ void k_count(int n_lm, const int64_t* __restrict__ ptr, const int32_t* __restrict__ lm_obs, const int32_t* __restrict__ obs_pos,
                        int64_t* __restrict__ cnt) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_lm) return;
  const int64_t b0 = ptr[l], k = ptr[l + 1] - b0;
  int64_t c = k * (k + 1) / 2;
  for (int64_t a = 1; a < k; a++) {
    const int pa = obs_pos[lm_obs[b0 + a]];
    for (int64_t b = 0; b < a; b++) c += obs_pos[lm_obs[b0 + b]] == pa;
  }
  cnt[l] = c;
}

__global__ void k_emit(int n_lm, int nrv, const int64_t* __restrict__ ptr, const int32_t* __restrict__ lm_obs, const int32_t* __restrict__ obs_pos,
                       const int64_t* __restrict__ off, uint64_t* __restrict__ key, int32_t* __restrict__ oa_out, int32_t* __restrict__ ob_out) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_lm) return;
  const int64_t b0 = ptr[l], k = ptr[l + 1] - b0;
  int64_t w = off[l];
  for (int64_t a = 0; a < k; a++) {
    const int32_t xa0 = lm_obs[b0 + a]; const int pa0 = obs_pos[xa0];
    for (int64_t b = 0; b <= a; b++) {
      int32_t oa = xa0, ob = lm_obs[b0 + b];
      int pa = pa0, pb = obs_pos[ob];
      if (pa < pb) { const int32_t t = oa; oa = ob; ob = t; const int u = pa; pa = pb; pb = u; }
      const uint64_t kk = (uint64_t)pa * (uint64_t)nrv + (uint64_t)pb;
      key[w] = kk; oa_out[w] = oa; ob_out[w] = ob; w++;
      if (pa == pb && oa != ob) { key[w] = kk; oa_out[w] = ob; ob_out[w] = oa; w++; }   // same camera twice
    }
  }
}

__global__ void k_iota(int64_t n, uint32_t* v) { const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = (uint32_t)i; }
__global__ void k_gather(int64_t n, const uint32_t* __restrict__ idx, const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                         int32_t* __restrict__ ao, int32_t* __restrict__ bo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { ao[i] = a[idx[i]]; bo[i] = b[idx[i]]; }
}

template <class T> static T* dev(const std::vector<T>& h) { T* p; CHECK(hipMalloc(&p, sizeof(T) * (h.size() + 1))); CHECK(hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice)); return p; }
template <class T> static std::vector<T> rd(FILE* f, size_t n) { std::vector<T> v(n); if (n && std::fread(v.data(), sizeof(T), n, f) != n) { std::printf("short read\n"); std::exit(2); } return v; }

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: proto <input.bin>\n"); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::printf("cannot open %s\n", argv[1]); return 2; }
  const std::vector<int64_t> head = rd<int64_t>(f, 5);   // n_lm, n_obs, nrv, n_terms, n_blocks
  const int n_lm = (int)head[0]; const int64_t n_obs = head[1]; const int nrv = (int)head[2]; const int64_t n_terms = head[3], n_blocks = head[4];
  const auto ptr = rd<int64_t>(f, n_lm + 1); const auto lm_obs = rd<int32_t>(f, n_obs); const auto obs_pos = rd<int32_t>(f, n_obs);
  const auto exp_oa = rd<int32_t>(f, n_terms); const auto exp_ob = rd<int32_t>(f, n_terms); const auto exp_ptr = rd<int64_t>(f, n_blocks + 1);
  std::fclose(f);
  std::printf("landmarks %d, observations %lld, reduced variables %d, expected terms %lld in %lld blocks\n", n_lm, (long long)n_obs, nrv, (long long)n_terms, (long long)n_blocks);
  int64_t* d_ptr = dev(ptr); int32_t* d_lm_obs = dev(lm_obs); int32_t* d_pos = dev(obs_pos);
  int64_t *d_cnt, *d_off; CHECK(hipMalloc(&d_cnt, 8 * (n_lm + 1))); CHECK(hipMalloc(&d_off, 8 * (n_lm + 1)));
  hipEvent_t ev[8]; for (auto& e : ev) CHECK(hipEventCreate(&e));
  size_t tmp_bytes = 0, need = 0; void* tmp = nullptr;
  auto ensure = [&](size_t n) { if (n > tmp_bytes) { if (tmp) CHECK(hipFree(tmp)); CHECK(hipMalloc(&tmp, n)); tmp_bytes = n; } };
  for (int rep = 0; rep < 2; rep++) {   // second repetition is the timed one (allocations, lazy module load out of the way)
    CHECK(hipEventRecord(ev[0], 0));
    hipLaunchKernelGGL(k_count, dim3((n_lm + 255) / 256), dim3(256), 0, 0, n_lm, d_ptr, d_lm_obs, d_pos, d_cnt);
    CHECK(hipMemsetAsync(d_cnt + n_lm, 0, 8, 0));
    need = 0; CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_cnt, d_off, n_lm + 1)); ensure(need);
    CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, need, d_cnt, d_off, n_lm + 1));
    int64_t total = 0; CHECK(hipMemcpy(&total, d_off + n_lm, 8, hipMemcpyDeviceToHost));
    CHECK(hipEventRecord(ev[1], 0));
    if (total != n_terms) { std::printf("FAIL: %lld terms counted, %lld expected\n", (long long)total, (long long)n_terms); return 1; }
    static uint64_t *d_key = nullptr, *d_key2; static int32_t *d_oa, *d_ob, *d_oa2, *d_ob2; static uint32_t *d_idx, *d_idx2; static uint64_t* d_uniq; static int32_t* d_runs; static int64_t* d_pp; static int* d_nruns;
    if (!d_key) {
      CHECK(hipMalloc(&d_key, 8 * total)); CHECK(hipMalloc(&d_key2, 8 * total)); CHECK(hipMalloc(&d_oa, 4 * total)); CHECK(hipMalloc(&d_ob, 4 * total));
      CHECK(hipMalloc(&d_oa2, 4 * total)); CHECK(hipMalloc(&d_ob2, 4 * total)); CHECK(hipMalloc(&d_idx, 4 * total)); CHECK(hipMalloc(&d_idx2, 4 * total));
      CHECK(hipMalloc(&d_uniq, 8 * total)); CHECK(hipMalloc(&d_runs, 4 * (total + 1))); CHECK(hipMalloc(&d_pp, 8 * (total + 1))); CHECK(hipMalloc(&d_nruns, 4));
    }
    hipLaunchKernelGGL(k_emit, dim3((n_lm + 255) / 256), dim3(256), 0, 0, n_lm, nrv, d_ptr, d_lm_obs, d_pos, d_off, d_key, d_oa, d_ob);
    hipLaunchKernelGGL(k_iota, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, total, d_idx);
    CHECK(hipEventRecord(ev[2], 0));
    int bits = 1; while (((uint64_t)1 << bits) < (uint64_t)nrv * (uint64_t)nrv) bits++;
    need = 0; CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, d_key, d_key2, d_idx, d_idx2, (int)total, 0, bits)); ensure(need);
    CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, need, d_key, d_key2, d_idx, d_idx2, (int)total, 0, bits));
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, total, d_idx2, d_oa, d_ob, d_oa2, d_ob2);
    CHECK(hipEventRecord(ev[3], 0));
    need = 0; CHECK(hipcub::DeviceRunLengthEncode::Encode(nullptr, need, d_key2, d_uniq, d_runs, d_nruns, (int)total)); ensure(need);
    CHECK(hipcub::DeviceRunLengthEncode::Encode(tmp, need, d_key2, d_uniq, d_runs, d_nruns, (int)total));
    int nruns = 0; CHECK(hipMemcpy(&nruns, d_nruns, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemsetAsync(d_runs + nruns, 0, 4, 0));
    need = 0; CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_runs, d_pp, nruns + 1)); ensure(need);   // int32 counts -> int64 offsets
    CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, need, d_runs, d_pp, nruns + 1));
    CHECK(hipEventRecord(ev[4], 0)); CHECK(hipEventSynchronize(ev[4]));
    if (rep == 0) continue;
    float t01, t12, t23, t34; CHECK(hipEventElapsedTime(&t01, ev[0], ev[1])); CHECK(hipEventElapsedTime(&t12, ev[1], ev[2])); CHECK(hipEventElapsedTime(&t23, ev[2], ev[3])); CHECK(hipEventElapsedTime(&t34, ev[3], ev[4]));
    std::printf("count + scan %.3f ms | emit %.3f ms | stable sort (%d bits) + gather %.3f ms | run-length encode + scan %.3f ms | total %.3f ms\n", t01, t12, bits, t23, t34, t01 + t12 + t23 + t34);
    std::vector<int32_t> oa(total), ob(total); std::vector<int64_t> pp(nruns + 1);
    CHECK(hipMemcpy(oa.data(), d_oa2, 4 * total, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ob.data(), d_ob2, 4 * total, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(pp.data(), d_pp, 8 * (nruns + 1), hipMemcpyDeviceToHost));
    bool ok = nruns == n_blocks && oa == exp_oa && ob == exp_ob && pp == exp_ptr;
    std::printf("%s: %d blocks (expected %lld), term lists %s, block offsets %s\n", ok ? "PASS" : "FAIL", nruns, (long long)n_blocks,
                (oa == exp_oa && ob == exp_ob) ? "identical" : "DIFFERENT", pp == exp_ptr ? "identical" : "DIFFERENT");
    return ok ? 0 : 1;
  }
  return 0;
}
