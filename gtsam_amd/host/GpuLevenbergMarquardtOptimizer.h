// GpuLevenbergMarquardtOptimizer.h -- the GTSAM-side drop-in: same constructors and accessors as
// gtsam::LevenbergMarquardtOptimizer (nonlinear/LevenbergMarquardtOptimizer.h:59-71), so user code changes
// one type name.  GTSAM has no plugin ABI; its extension point is subclassing the optimizer
// (NonlinearOptimizer::iterate() pure virtual nonlinear/NonlinearOptimizer.h:136; precedent
// tests/testNonlinearOptimizer.cpp:507-551).  This class overrides iterate() / optimize() and re-states
// tryLambda()'s decisions (LevenbergMarquardtOptimizer.cpp:121-270) around calls through the C ABI
// (include/gtsam_amd.h) into the HIP library; the graph, the values and the lambda state machine stay host
// C++ with GTSAM's own types.
#pragma once

#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>

#include <memory>
#include <vector>

#include "../../include/gtsam_amd.h"

namespace gtsam_amd {

/// Multi-GPU (one process per GPU): this process owns shard `shard` of `n_shards` (landmarks by rank modulo n_shards with all
/// their factors; every shard keeps all cameras / poses).  `allreduce` sums a DEVICE buffer of doubles in place across the
/// shards on the given stream (e.g. ncclAllReduce(ptr, ptr, n, ncclDouble, ncclSum, comm, (hipStream_t)stream), see
/// RcclExchange.h); every process builds the same graph and runs the same optimizer calls in lock step.
struct ShardSpec {
  int shard = 0, n_shards = 1;
  gtg_allreduce_fn allreduce = nullptr;
  void* user = nullptr;
};

class GpuLevenbergMarquardtOptimizer : public gtsam::LevenbergMarquardtOptimizer {
 public:
  GpuLevenbergMarquardtOptimizer(const gtsam::NonlinearFactorGraph& graph, const gtsam::Values& initialValues,
                                 const gtsam::LevenbergMarquardtParams& params = gtsam::LevenbergMarquardtParams(),
                                 int device = 0, const ShardSpec& shards = ShardSpec());
  GpuLevenbergMarquardtOptimizer(const gtsam::NonlinearFactorGraph& graph, const gtsam::Values& initialValues,
                                 const gtsam::Ordering& ordering,
                                 const gtsam::LevenbergMarquardtParams& params = gtsam::LevenbergMarquardtParams(),
                                 int device = 0, const ShardSpec& shards = ShardSpec());
  ~GpuLevenbergMarquardtOptimizer() override;

  /// One LM iteration on the GPU; state_ (values, error, lambda, counters) is updated exactly like the
  /// reference does, and like the reference (LevenbergMarquardtOptimizer.cpp:273-308) it returns the graph it linearised at
  /// the values it started from: the device's whitened records [A1 A2 b], downloaded and wrapped as JacobianFactors in the
  /// order of the nonlinear graph.  (optimize() does not pay for that: its iterations stay on the device.)
  gtsam::GaussianFactorGraph::shared_ptr iterate() override;

  /// A/B testing of the two halves of the path against the CPU (LevenbergMarquardtOptimizer.h:112-113,
  /// NonlinearOptimizer.h:129-130; precedent tests/testNonlinearOptimizer.cpp:507-551):
  /// linearize(): the DEVICE's linearisation of graph_ at values(), as a GaussianFactorGraph -- compare with
  /// gtsam::LevenbergMarquardtOptimizer::linearize() / graph.linearize(values).
  gtsam::GaussianFactorGraph::shared_ptr linearize() const override;
  /// solve(): when `gfg` is the damped system LevenbergMarquardtState::buildDampedSystem makes of this graph's linearisation
  /// at values() (its factors followed by one damping factor per variable), lambda and the damping mode are read off the
  /// damping factors and the system is solved on the DEVICE (Schur complement + tile Cholesky, or PCG when params say
  /// Iterative); any other graph goes to the reference's CPU solve.  Throws IndeterminantLinearSystemException where the
  /// reference would.
  gtsam::VectorValues solve(const gtsam::GaussianFactorGraph& gfg, const gtsam::NonlinearOptimizerParams& params) const override;

  /// defaultOptimize() (nonlinear/NonlinearOptimizer.cpp:62-117) with the Values kept on the device
  /// between iterations and synchronised to the host once at the end (and for the iteration hook).
  const gtsam::Values& optimize() override;

  /// per-phase device timings (ms, accumulated since enablePhaseTiming(true)) -- names via gtg_phase_name()
  void enablePhaseTiming(bool on);
  std::vector<double> phaseMilliseconds() const;
  std::vector<long long> phaseCalls() const;
  /// The C-ABI handle behind this optimizer, for the library's read-only getters (gtg_cholesky_flops_block_level, gtg_reduced_dim,
  /// gtg_get_phase_ms, ...).  Calls that change the handle's state behind the optimizer's back are the caller's responsibility.
  gtg_handle handle() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
  void init(const gtsam::NonlinearFactorGraph& graph, const gtsam::Values& initial, int device, const ShardSpec& shards);
  bool tryLambdaDevice();            // LevenbergMarquardtOptimizer::tryLambda restated
  void iterateDevice();              // LevenbergMarquardtOptimizer::iterate restated (logFile rows, SUMMARY header)
  void writeLogFileDevice(double currentError);
  void syncValuesToHost(bool force);
  void adoptStateIfForeign() const;      // a State installed by the inherited tryLambda() (or anybody else): the device follows it
  gtsam::GaussianFactorGraph::shared_ptr downloadLinearization() const;   // the device's current records as JacobianFactors
};

/// LevenbergMarquardtParams that name the GPU optimizer as their optimizer type (LevenbergMarquardtParams.h:45 does so for the
/// reference's own), for the callers that are templated on it: gtsam::GncOptimizer<gtsam::GncParams<GpuLevenbergMarquardtParams>>
/// (nonlinear/GncOptimizer.h:47, 185-187, 236-238) then runs every inner weighted least-squares problem on the device.
struct GpuLevenbergMarquardtParams : gtsam::LevenbergMarquardtParams {
  using OptimizerType = GpuLevenbergMarquardtOptimizer;
  GpuLevenbergMarquardtParams() = default;
  GpuLevenbergMarquardtParams(const gtsam::LevenbergMarquardtParams& p) : gtsam::LevenbergMarquardtParams(p) {}   // NOLINT: a drop-in for the base
};

}  // namespace gtsam_amd
