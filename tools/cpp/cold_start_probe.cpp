// tools/cpp/cold_start_probe.cpp -- where do the milliseconds of the FIRST optimizer of a process go?  A C-ABI-only client (no GTSAM):
// reads a BAL file with gtg_io_read_bal, then walks through what GpuLevenbergMarquardtOptimizer's constructor and first optimize()
// call make of the library -- HIP runtime start, gtg_create, gtg_upload_problem, gtg_set_values, gtg_error, gtg_linearize, two
// gtg_try_lambda -- with the wall-clock of every call, and then does the same with a SECOND handle of the same process (warm).
// The differences are the one-time costs (runtime initialisation, code-object load, first launch of every kernel, first allocations).
//
//   cold_start_probe <BALfile> [--prewarm]      (--prewarm: gtg_prewarm(0) first, timed on its own)
//
// Prints one line per step: "<step> <cold ms> <warm ms>".
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gtsam_amd.h"

typedef std::chrono::high_resolution_clock Clock;
static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: cold_start_probe <BALfile> [--prewarm]\n"); return 2; }
  const bool prewarm = argc > 2 && !std::strcmp(argv[2], "--prewarm");
  int64_t nc = 0, np = 0, no = 0;
  if (gtg_io_bal_sizes(argv[1], &nc, &np, &no) != GTG_OK) { std::printf("%s\n", gtg_io_last_error()); return 1; }
  std::vector<double> cams(17 * nc), pts(3 * np), z(2 * no);
  std::vector<int32_t> oc(no), op(no);
  if (gtg_io_read_bal(argv[1], nc, np, no, cams.data(), pts.data(), oc.data(), op.data(), z.data()) != GTG_OK) { std::printf("%s\n", gtg_io_last_error()); return 1; }
  // the timeSFMBAL problem: cameras are variables 0..nc-1, points nc..nc+np-1, Unit(2) noise
  std::vector<int32_t> vt(nc + np, GTG_VAR_POINT3), sfm_pt(no), nz(no, 0);
  for (int64_t i = 0; i < nc; i++) vt[i] = GTG_VAR_SFM_CAMERA;
  for (int64_t k = 0; k < no; k++) sfm_pt[k] = (int32_t)(nc + op[k]);
  const int32_t nkind = GTG_NOISE_UNIT, ndim = 2; const int64_t noff = 0; const double ndata = 0.0;
  gtg_problem pb{};
  pb.n_vars = (int32_t)(nc + np); pb.var_type = vt.data();
  pb.n_noise = 1; pb.noise_kind = &nkind; pb.noise_dim = &ndim; pb.noise_off = &noff; pb.noise_data = &ndata;
  pb.n_sfm = no; pb.sfm_cam = oc.data(); pb.sfm_point = sfm_pt.data(); pb.sfm_z = z.data(); pb.sfm_noise = nz.data();
  std::vector<double> values(cams);
  values.insert(values.end(), pts.begin(), pts.end());

  std::vector<std::pair<std::string, double>> rec[2];
  double prewarm_ms = 0.0;
#ifdef GTG_HAVE_PREWARM
  if (prewarm) { const auto a = Clock::now(); gtg_prewarm(0); prewarm_ms = ms(a, Clock::now()); }
#endif
  for (int pass = 0; pass < 2; pass++) {
    auto& r = rec[pass];
    auto t = Clock::now();
    auto lap = [&](const char* what) { const auto n = Clock::now(); r.emplace_back(what, ms(t, n)); t = n; };
    int nd = 0; (void)hipGetDeviceCount(&nd); lap("hipGetDeviceCount (runtime start)");
    (void)hipSetDevice(0); (void)hipFree(nullptr); lap("hipSetDevice + context");
    gtg_handle h = nullptr;
    if (gtg_create(&h, 0) != GTG_OK) { std::printf("%s\n", gtg_last_error()); return 1; }
    lap("gtg_create");
    if (gtg_upload_problem(h, &pb, 0, 1) != GTG_OK) { std::printf("%s\n", gtg_last_error()); return 1; }
    lap("gtg_upload_problem");
    gtg_set_values(h, values.data(), (int64_t)values.size()); lap("gtg_set_values");
    double e = 0; gtg_error(h, &e); lap("gtg_error");
    gtg_linearize(h); lap("gtg_linearize (first)");
    double out[4];
    gtg_try_lambda(h, 1e-4, 1, 1e-6, 1e32, out); lap("gtg_try_lambda (first)");
    gtg_try_lambda(h, 1e-4, 1, 1e-6, 1e32, out); lap("gtg_try_lambda (second)");
    gtg_accept(h); lap("gtg_accept");
    gtg_linearize(h); lap("gtg_linearize (second)");
    gtg_try_lambda(h, 1e-5, 1, 1e-6, 1e32, out); lap("gtg_try_lambda (third)");
    std::vector<double> back(values.size());
    gtg_get_values(h, back.data(), (int64_t)back.size()); lap("gtg_get_values");
    gtg_destroy(h); lap("gtg_destroy");
  }
  if (prewarm) std::printf("%-36s %9.2f\n", "gtg_prewarm", prewarm_ms);
  double tc = 0, tw = 0;
  for (size_t k = 0; k < rec[0].size(); k++) { std::printf("%-36s %9.2f %9.2f\n", rec[0][k].first.c_str(), rec[0][k].second, rec[1][k].second); tc += rec[0][k].second; tw += rec[1][k].second; }
  std::printf("%-36s %9.2f %9.2f\n", "total", tc, tw);
  return 0;
}
