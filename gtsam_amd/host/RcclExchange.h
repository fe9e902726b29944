// RcclExchange.h -- the all-reduce callback of a multi-GPU job for C++ hosts: one process per GPU, one RCCL communicator.
//
//   ncclComm_t comm;  ncclCommInitRank(&comm, world, id, rank);           // id from ncclGetUniqueId on rank 0, shared by any means
//   gtsam_amd::RcclExchange ex{comm};
//   gtsam_amd::ShardSpec shards{rank, world, &gtsam_amd::RcclExchange::allreduce, &ex};
//   gtsam_amd::GpuLevenbergMarquardtOptimizer lm(graph, initial, params, /*device=*/local_rank, shards);
//
// The collective is enqueued on the library's own stream, behind the kernels that produced the buffer and in front of the
// ones that consume it: no host synchronisation in the exchange.  (The Python host does the same through
// torch.distributed, gtsam_amd/distributed.py.)  Header-only; link librccl.  Compile-checked in this repository, exercised
// on hardware only through the Python path (the build image has one GPU per box).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>

namespace gtsam_amd {

struct RcclExchange {
  ncclComm_t comm;
  // gtg_allreduce_fn: sum `n` doubles at device pointer `ptr` in place across the communicator, on `stream`
  static int allreduce(void* ptr, int64_t n, void* stream, void* user) {
    RcclExchange* self = static_cast<RcclExchange*>(user);
    return ncclAllReduce(ptr, ptr, (size_t)n, ncclDouble, ncclSum, self->comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
  }
};

}  // namespace gtsam_amd
