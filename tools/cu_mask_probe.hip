// tools/cu_mask_probe.hip -- where do the workgroups of a kernel on a hipExtStreamCreateWithCUMask stream land?
// Prints, per mask, the set of (xcc, se, cu) the workgroups ran on.   hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/cu_mask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include <string>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(2); } } while (0)

__global__ void k_where(unsigned* out, long long ticks) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}

static void run(const char* name, const std::vector<uint32_t>& mask, int nblocks) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { std::printf("{\"mask\": \"%s\", \"error\": \"%s\"}\n", name, hipGetErrorString(e)); return; }
  unsigned* d; CHECK(hipMalloc(&d, sizeof(unsigned) * 2 * nblocks));
  CHECK(hipMemset(d, 0, sizeof(unsigned) * 2 * nblocks));
  hipLaunchKernelGGL(k_where, dim3(nblocks), dim3(256), 0, s, d, 2000LL);   // 20 us each
  CHECK(hipStreamSynchronize(s));
  std::vector<unsigned> h(2 * nblocks);
  CHECK(hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * nblocks, hipMemcpyDeviceToHost));
  std::set<unsigned> places; std::set<unsigned> xccs;
  for (int i = 0; i < nblocks; i++) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    places.insert((xcc << 16) | (se << 8) | (sh << 4) | cu); xccs.insert(xcc);
  }
  std::printf("{\"mask\": \"%s\", \"distinct_cus\": %zu, \"xccs\": %zu, \"first\": [", name, places.size(), xccs.size());
  int n = 0;
  for (unsigned p : places) { if (n++ >= 12) break; std::printf("%s\"x%u.se%u.sh%u.cu%u\"", n > 1 ? ", " : "", p >> 16, (p >> 8) & 0xff, (p >> 4) & 0xf, p & 0xf); }
  std::printf("]}\n");
  CHECK(hipFree(d)); CHECK(hipStreamDestroy(s));
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  std::printf("{\"compute_units\": %d}\n", ncu);
  const int words = (ncu + 31) / 32;
  auto mk = [&](auto pred) { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < ncu; i++) if (pred(i)) m[i >> 5] |= 1u << (i & 31); return m; };
  run("all", mk([](int) { return true; }), 4096);
  run("bit0", mk([](int i) { return i == 0; }), 512);
  run("bit1", mk([](int i) { return i == 1; }), 512);
  run("bit8", mk([](int i) { return i == 8; }), 512);
  run("bit32", mk([](int i) { return i == 32; }), 512);
  run("bit255", mk([&](int i) { return i == ncu - 1; }), 512);
  run("bits0-7", mk([](int i) { return i < 8; }), 1024);
  run("bits0-31", mk([](int i) { return i < 32; }), 2048);
  run("all-but-last2", mk([&](int i) { return i < ncu - 2; }), 4096);
  run("all-but-first2", mk([&](int i) { return i >= 2; }), 4096);
  run("last2", mk([&](int i) { return i >= ncu - 2; }), 512);
  run("first2", mk([&](int i) { return i < 2; }), 512);
  std::vector<uint32_t> one(1, 0x3u);
  run("one-word-0x3", one, 512);
  std::vector<uint32_t> onef(1, 0xfffffffcu);
  run("one-word-0xfffffffc", onef, 4096);
  return 0;
}
