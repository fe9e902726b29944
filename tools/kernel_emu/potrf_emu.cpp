// tools/kernel_emu/potrf_emu.cpp -- csrc/chol_device.h::potrf_body (the diagonal-tile body both Cholesky schedules share: pivot chain
// wavefront, follower wavefronts polling its progress words in LDS, deferred MFMA updates, operand images) on host threads.
// tests/test_potrf_emulated.py factors random SPD tiles with it, repeatedly: the factor and the block inverses against numpy, and every
// repetition against the first bit for bit -- the body's LDS protocol under timings no GPU would produce.
#include "emu_hip.h"

constexpr int kTile = 128;
constexpr int kTileDoubles = kTile * kTile;
#include "../../gtsam_amd/csrc/chol_device.h"

// tile: 128 x 128 row-major (lower part read, factor written back in place); Xinv: 128 x 128 doubles (the four 32 x 32 inverses, the
// operand images, the progress word at gt::kFlagOff); returns the progress word
extern "C" long long emu_potrf128(double* tile, double* Xinv, double* fail, int write_through, long long epoch) {
  static char smem[sizeof(double) * gt::kPotrfSmemDoubles + 64];
  long long* pflag = reinterpret_cast<long long*>(Xinv + gt::kFlagOff);
  emu::run_workgroup(512, 0, [&] {
    gt::potrf_body(smem, tile, 0, Xinv, fail, nullptr, epoch, pflag, 64, false, write_through != 0, nullptr, nullptr);
  });
  return *pflag;
}

// the same with the reference's rank test switched on (pivot kinds of the tile's 128 columns: chol_device.h::stage_potrf)
extern "C" long long emu_potrf128_ranktest(double* tile, double* Xinv, double* fail, int write_through, long long epoch,
                                           const unsigned char* pivot_kind, double* tile_exp) {
  static char smem[sizeof(double) * gt::kPotrfSmemDoubles + 64];
  long long* pflag = reinterpret_cast<long long*>(Xinv + gt::kFlagOff);
  emu::run_workgroup(512, 0, [&] {
    gt::potrf_body(smem, tile, 0, Xinv, fail, nullptr, epoch, pflag, 64, false, write_through != 0, pivot_kind, tile_exp);
  });
  return *pflag;
}
