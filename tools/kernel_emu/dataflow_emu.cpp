// tools/kernel_emu/dataflow_emu.cpp -- the two persistent kernels of the dataflow Cholesky (csrc/chol_dataflow.hip: bulk_loop, chain_loop,
// with chol_device.h::potrf_body inside) from their OWN source on host threads: one PROCESS per workgroup (a workgroup's LDS is the
// process's static storage), one thread per work-item, device memory = a shared mapping, so that the chain and the bulk workgroups run
// side by side and talk through flags exactly as on the device.  tests/test_dataflow_emulated.py factors a small dense system with it.
//   dataflow_emu <in.bin> <out.bin>
//   in : int64 header {nt, n_slots, n_tasks, n_kpairs, n_chain_wg, n_bulk_wg, sh}, S [n_slots][128][128] f64, tasks [n_tasks][12] i32,
//        klist [n_kpairs][2] i32, chain_slots [3 nt] i32, chain_off [n_chain_wg + 1] i32, chain_tiles [nt] i32
//   out: S, Xinv [nt][128][128] f64, fail [2] f64, ctrl [16] i32
#include "emu_hip.h"

#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>

constexpr int kTile = 128;
constexpr int kTileDoubles = kTile * kTile;
constexpr int kSub = 16;
#include "../../gtsam_amd/csrc/chol_dataflow.hip"

template <class T> static T* shared_alloc(size_t n) {
  void* p = mmap(nullptr, sizeof(T) * std::max<size_t>(n, 1), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) { std::perror("mmap"); std::exit(2); }
  return static_cast<T*>(p);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int64_t h[7];
  if (std::fread(h, sizeof(int64_t), 7, f) != 7) return 2;
  const int nt = (int)h[0], n_slots = (int)h[1], n_tasks = (int)h[2], n_kp = (int)h[3], n_chain = (int)h[4], n_bulk = (int)h[5];
  const long long sh = h[6];
  double* S = shared_alloc<double>((size_t)n_slots * kTileDoubles);
  double* Xinv = shared_alloc<double>((size_t)nt * kTileDoubles);
  int32_t* tasks = shared_alloc<int32_t>((size_t)12 * n_tasks);
  int32_t* klist = shared_alloc<int32_t>((size_t)6 * n_kp);   // 6 words per contraction step: two slots, two 64-bit sub-tile masks (chol_dataflow.hip::kStepWords)
  int32_t* chain_slots = shared_alloc<int32_t>((size_t)4 * nt);   // chol_dataflow.hip::kChainWords
  int32_t* chain_off = shared_alloc<int32_t>((size_t)n_chain + 1);
  int32_t* chain_tiles = shared_alloc<int32_t>((size_t)nt);
  long long* tile_flag = shared_alloc<long long>((size_t)n_slots + sh + 8);
  long long* part_flag = shared_alloc<long long>((size_t)n_slots + sh + 8);
  long long* pd_flag = shared_alloc<long long>((size_t)nt + sh + 8);
  int32_t* ctrl = shared_alloc<int32_t>(32);
  double* fail = shared_alloc<double>(2);
  auto rd = [&](void* p, size_t bytes) { if (bytes && std::fread(p, 1, bytes, f) != bytes) { std::fprintf(stderr, "short input\n"); std::exit(2); } };
  rd(S, sizeof(double) * (size_t)n_slots * kTileDoubles); rd(tasks, 4 * (size_t)12 * n_tasks); rd(klist, 4 * (size_t)6 * n_kp);
  rd(chain_slots, 4 * (size_t)4 * nt); rd(chain_off, 4 * ((size_t)n_chain + 1)); rd(chain_tiles, 4 * (size_t)nt);
  std::fclose(f);
  ctrl[2] = ctrl[3] = ctrl[4] = ctrl[5] = -1;
  const long long epoch = 1;
  std::vector<pid_t> kids;
  for (int w = 0; w < n_chain + n_bulk; w++) {
    const pid_t pid = fork();
    if (pid < 0) { std::perror("fork"); return 2; }
    if (pid == 0) {
      if (w < n_chain) {
        static char smem[gt::kSmemChain + 64];
        emu::run_workgroup(512, (unsigned)w, [&] {
          gt::chain_loop(smem, S, Xinv, pd_flag, tile_flag, chain_slots, fail, epoch, ctrl, nullptr, chain_tiles + chain_off[w],
                         chain_off[w + 1] - chain_off[w], nullptr, nullptr);
        });
      } else {
        static char smem[gt::kSmemBulk + 64];
        emu::run_workgroup(gt::kBulkThreads, (unsigned)(w - n_chain), [&] {
          gt::bulk_loop(smem, S, tasks, n_tasks, klist, tile_flag, part_flag, pd_flag, Xinv, ctrl, fail, epoch, nullptr);
        });
      }
      _exit(0);
    }
    kids.push_back(pid);
  }
  int bad = 0;
  for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad++; }
  FILE* o = std::fopen(argv[2], "wb");
  if (!o) return 2;
  std::fwrite(S, sizeof(double), (size_t)n_slots * kTileDoubles, o);
  std::fwrite(Xinv, sizeof(double), (size_t)nt * kTileDoubles, o);
  std::fwrite(fail, sizeof(double), 2, o);
  std::fwrite(ctrl, sizeof(int32_t), 16, o);
  std::fclose(o);
  return bad ? 3 : 0;
}
